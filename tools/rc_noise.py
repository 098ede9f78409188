#!/usr/bin/env python3
"""The scalar decoder loop on noise planes (7.5 bits per symbol: no dominant symbols, no usable bucket table), 1-4 streams
interleaved on one thread: the loop a third of the pool's worker-seconds goes to.  (The general loop -- two stream bytes
and a 7-bit shift per renormalisation step -- and the compiler's register allocation of the four-stream loop were measured
against the byte-aligned, hand-allotted one in round 4: profiles/r04/e_rc_noise_*, m_rc_noise_*.)  CPU only.
usage: rc_noise.py [blocks per plane]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 200
kmax = 4
n = 60000 * nb
rs = np.random.RandomState(1)
planes = [np.minimum(rs.randint(0, 256, n), rs.randint(64, 320, n)).astype(np.uint8) for _ in range(kmax)]
streams = api.range_encode_multi(planes)
for k in range(1, kmax + 1):
    best = 1e9
    for _ in range(3):
        t = time.time()
        out, _ = api.range_decode_multi(streams[:k], n)
        best = min(best, time.time() - t)
    assert all(np.array_equal(o, p) for o, p in zip(out, planes[:k]))
    print("noise planes, %d stream(s) in the loop: decode %6.1f Msym/s per thread (%5.1f per stream)" % (k, k * n / best / 1e6, n / best / 1e6), flush=True)
