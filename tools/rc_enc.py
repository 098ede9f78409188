#!/usr/bin/env python3
"""The encoder's 16-lane AVX-512 loop on one thread: noise planes (per-lane look-ups) and dominant-symbol planes (candidate
compares), 4 / 8 / 16 planes, against the scalar loop of three.  CPU only.  usage: rc_enc.py [blocks per plane]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = 60000 * nb
rs = np.random.RandomState(1)


def plane(kind):
    if kind == "two":
        return rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.8, 0.2])
    if kind == "noise":
        return np.minimum(rs.randint(0, 256, n), rs.randint(64, 320, n)).astype(np.uint8)
    raise ValueError(kind)


for kind in ("noise", "two"):
    base = [plane(kind) for _ in range(4)]
    best = 1e9
    for _ in range(2):
        t = time.time(); api.range_encode_multi(base[:3]); best = min(best, time.time() - t)
    print("scalar loop of three, kind %-5s encode %7.1f Msym/s per thread" % (kind, 3 * n / best / 1e6), flush=True)
    for k in (4, 8, 16):
        ps = [base[i % 4] for i in range(k)]
        best = 1e9
        for _ in range(3):
            t = time.time(); api.range_encode_vec(ps); best = min(best, time.time() - t)
        print("16-lane loop, %2d planes of kind %-5s encode %7.1f Msym/s per thread (%.1f per stream)" % (k, kind, k * n / best / 1e6, n / best / 1e6), flush=True)
