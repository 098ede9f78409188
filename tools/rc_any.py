#!/usr/bin/env python3
"""The decoder's 16-lane AVX-512 loop for planes of ANY statistics against the scalar loop of four, one thread, noise planes
(7.5 bits per symbol) and, for comparison, dominant-symbol planes.  CPU only.  usage: rc_any.py [blocks per plane]"""
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
    if kind == "gauss":    # ~6 bits per symbol
        return np.clip(np.rint(rs.normal(128, 20, n)), 0, 255).astype(np.uint8)
    raise ValueError(kind)


for kind in ("noise", "gauss", "two"):
    base = [plane(kind) for _ in range(4)]
    ss4 = [api.range_encode(p) for p in base]
    best = 1e9
    for _ in range(2):
        t = time.time(); api.range_decode_multi(ss4, n); best = min(best, time.time() - t)
    print("scalar loop of four,       4 planes of kind %-5s decode %7.1f Msym/s per thread (%.1f per stream)" % (kind, 4 * n / best / 1e6, n / best / 1e6), flush=True)
    for k in (4, 8, 12, 16, 32):
        ss = [ss4[i % 4] for i in range(k)]
        best = 1e9
        for _ in range(2):
            t = time.time(); api.range_decode_vec(ss, [n] * k, any_statistics=True); best = min(best, time.time() - t)
        print("any-statistics loop,      %2d planes of kind %-5s decode %7.1f Msym/s per thread (%.1f per stream)" % (k, kind, k * n / best / 1e6, min(k, 16) * n / best / 1e6 / k * (k / min(k, 16))), flush=True)
