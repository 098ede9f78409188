#!/usr/bin/env python3
"""The scalar decoder loop (1..4 planes interleaved) on one thread: noise planes and two-symbol planes.  CPU only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 100
n = 60000 * nb
rs = np.random.RandomState(1)
noise = [np.minimum(rs.randint(0, 256, n), rs.randint(64, 320, n)).astype(np.uint8) for _ in range(4)]
two = [rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.8, 0.2]) for _ in range(4)]
for name, planes in (("noise", noise), ("two", two), ("two+two+noise+noise", two[:2] + noise[:2])):
    ss = [api.range_encode(p) for p in planes]
    for k in (1, 2, 3, 4):
        best = 1e9
        for _ in range(3):
            t = time.time(); api.range_decode_multi(ss[:k], n); best = min(best, time.time() - t)
        print("scalar decoder loop, %d planes of kind %-20s %7.1f Msym/s per thread (%.1f per stream)" % (k, name, k * n / best / 1e6, n / best / 1e6), flush=True)
