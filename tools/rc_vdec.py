#!/usr/bin/env python3
"""The decoder's 16-lane AVX-512 loop for dominant-symbol planes on one thread: 4 / 8 / 16 planes of kind "two" (p = 0.8 / 0.2)
and "one" (p = 0.9997).  CPU only.  usage: rc_vdec.py [blocks per plane]"""
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
    return np.where(rs.random_sample(n) < 0.9997, 128, rs.randint(120, 136, n)).astype(np.uint8)


for kind in ("two", "one"):
    base = [api.range_encode(plane(kind)) for _ in range(4)]
    for k in (4, 8, 16):
        ss = [base[i % 4] for i in range(k)]
        best = 1e9
        for _ in range(3):
            t = time.time(); api.range_decode_vec(ss, [n] * k); best = min(best, time.time() - t)
        print("16-lane decoder loop, %2d planes of kind %-4s %7.1f Msym/s per thread (%.1f per stream)" % (k, kind, k * n / best / 1e6, n / best / 1e6), flush=True)
