#!/usr/bin/env python3
"""The decoder's 16-lane loop for dominant-symbol planes on one thread, one group of 16 lanes against two groups whose steps
are interleaved (WR_VEC_DUAL=1): 8 / 16 / 24 / 32 planes of kind "two" (p = 0.8 / 0.2) and "one" (p = 0.9997).  CPU only.
usage: [WR_VEC_DUAL=1] rc_vdec_dual.py [blocks per plane]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 60
n = 60000 * nb
rs = np.random.RandomState(1)


def plane(kind):
    if kind == "two":
        return rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.8, 0.2])
    return np.where(rs.random_sample(n) < 0.9997, 128, rs.randint(120, 136, n)).astype(np.uint8)


print("WR_VEC_DUAL =", os.environ.get("WR_VEC_DUAL", "0"))
for kind in ("two", "one"):
    planes = [plane(kind) for _ in range(4)]
    base = [api.range_encode(p) for p in planes]
    for k in (8, 16, 24, 32):
        ss = [base[i % 4] for i in range(k)]
        best = 1e9
        for _ in range(3):
            t = time.time(); dec, got = api.range_decode_vec(ss, [n] * k); best = min(best, time.time() - t)
        assert all(g == n for g in got) and all(np.array_equal(d, planes[i % 4]) for i, d in enumerate(dec))
        print("%2d planes of kind %-4s %7.1f Msym/s per thread (%.1f per stream)" % (k, kind, k * n / best / 1e6, n / best / 1e6), flush=True)
