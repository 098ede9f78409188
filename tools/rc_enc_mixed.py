#!/usr/bin/env python3
"""The encoder's 16-lane loop on sessions that mix noise planes with dominant-symbol planes: 16 planes of which m are noise,
with all lanes looking their symbols up per lane (the default when any lane needs it) against the candidate form with the
noise lanes on its scalar look-up path (WR_VEC_ENC_MISS_LANES=k: up to k such lanes).  CPU only."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

n = 60000 * (int(sys.argv[1]) if len(sys.argv) > 1 else 40)
rs = np.random.RandomState(1)
two = [rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.8, 0.2]) for _ in range(4)]
noise = [np.minimum(rs.randint(0, 256, n), rs.randint(64, 320, n)).astype(np.uint8) for _ in range(4)]
want = {id(p): api.range_encode(p) for p in two + noise}
print("WR_VEC_ENC_MISS_LANES =", os.environ.get("WR_VEC_ENC_MISS_LANES", "0"))
for m in (0, 1, 2, 4, 6, 8):
    ps = [noise[i % 4] for i in range(m)] + [two[i % 4] for i in range(16 - m)]
    best = 1e9
    for _ in range(3):
        t = time.time(); out = api.range_encode_vec(ps); best = min(best, time.time() - t)
    assert all(np.array_equal(o, want[id(p)]) for o, p in zip(out, ps))
    print("16 planes, %d of them noise: %7.1f Msym/s per thread (%.1f per stream)" % (m, 16 * n / best / 1e6, n / best / 1e6), flush=True)
