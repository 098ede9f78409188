#!/usr/bin/env python3
"""Host coder rates per thread for different GROUPINGS of plane kinds in one interleaved symbol loop: is a decoder
thread better off with the planes of one field (2 dominant-symbol planes + 2 noise planes) or with planes of one
kind taken from several fields?  CPU only.  usage: rc_mix.py [blocks per plane]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 300
n = 60000 * nb
rs = np.random.RandomState(1)


def plane(kind):
    if kind == "two":      # leading planes of a smooth field: p = 0.8 / 0.2
        return rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.8, 0.2])
    if kind == "one":      # one dominant symbol, p = 0.9997
        return np.where(rs.random_sample(n) < 0.9997, 128, rs.randint(120, 136, n)).astype(np.uint8)
    if kind == "noise":    # 7.5 bit/symbol
        return np.minimum(rs.randint(0, 256, n), rs.randint(64, 320, n)).astype(np.uint8)
    raise ValueError(kind)


def rate(kinds):
    ps = [plane(k) for k in kinds]
    best_e = best_d = 1e9
    for _ in range(2):
        t = time.time(); ss = api.range_encode_multi(ps); best_e = min(best_e, time.time() - t)
        t = time.time(); api.range_decode_multi(ss, n); best_d = min(best_d, time.time() - t)
    return len(ps) * n / best_e / 1e6, len(ps) * n / best_d / 1e6


for kinds in (["noise"], ["two"], ["one"], ["noise"] * 2, ["two"] * 2, ["noise"] * 3, ["two"] * 3, ["noise"] * 4, ["two"] * 4, ["one"] * 4,
              ["two", "two", "noise", "noise"], ["one", "two", "noise", "noise"], ["one", "two", "noise"]):
    e, d = rate(kinds)
    print("%-28s encode %6.1f Msym/s   decode %6.1f Msym/s   (per thread, all planes of the group together)" % ("+".join(kinds), e, d), flush=True)


# the coder pool with ONE worker: the same thread budget, but the worker's decoder loop takes up to 4 streams of any fields
def pool_rate(kinds, streams):
    ps = [plane(k) for k in kinds]
    api.set_coder_pool(1, streams)
    best_e = best_d = 1e9
    for _ in range(2):
        t = time.time(); ss = api.range_encode_pool(ps); best_e = min(best_e, time.time() - t)
        t = time.time(); api.range_decode_pool(ss, [n] * len(ps)); best_d = min(best_d, time.time() - t)
    api.set_coder_pool(0)
    return len(ps) * n / best_e / 1e6, len(ps) * n / best_d / 1e6


for streams in (3, 4):
    for kinds in (["two", "two", "noise", "noise"] * 3, ["noise"] * 12, ["two"] * 12, ["one", "two", "noise"] * 4):
        e, d = pool_rate(kinds, streams)
        print("pool, 1 worker, %d decoder streams, 12 planes %-22s encode %6.1f Msym/s   decode %6.1f Msym/s" % (streams, "+".join(kinds[:4]) + "..", e, d), flush=True)


# the 16-lane AVX-512 loop for dominant-symbol planes (one thread)
for k in (4, 8, 12, 16):
    for kind in ("two", "one"):
        ps = [plane(kind) for _ in range(k)]
        ss = [api.range_encode(p) for p in ps]
        best = 1e9
        try:
            for _ in range(3):
                t = time.time(); api.range_decode_vec(ss, [n] * k); best = min(best, time.time() - t)
            print("AVX-512 loop, %2d planes of kind %-4s decode %7.1f Msym/s per thread" % (k, kind, k * n / best / 1e6), flush=True)
        except api.WaveRangeError as exc:
            print("AVX-512 loop:", exc)
            break


# the 16-lane AVX-512 encoder loop (any kind of plane; one thread)
for k in (4, 8, 16):
    for kind in ("noise", "two", "one"):
        ps = [plane(kind) for _ in range(k)]
        best = 1e9
        try:
            for _ in range(3):
                t = time.time(); api.range_encode_vec(ps); best = min(best, time.time() - t)
            print("AVX-512 encoder loop, %2d planes of kind %-5s encode %7.1f Msym/s per thread" % (k, kind, k * n / best / 1e6), flush=True)
        except api.WaveRangeError as exc:
            print("AVX-512 encoder loop:", exc)
            break
