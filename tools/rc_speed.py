#!/usr/bin/env python3
"""Single-thread speed of the host plane coder (wr_range_encode / wr_range_decode) on planes of
different entropy, then of 2, 3 and 4 planes interleaved in one thread (wr_range_*_multi).
usage: rc_speed.py [libpath] [Msym]"""
import ctypes as C
import os
import sys
import time

import numpy as np

lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "waverange_amd", "libwaverange_amd.so")
n = (int(sys.argv[2]) if len(sys.argv) > 2 else 64) * 1000 * 1000
L = C.CDLL(lib)
L.wr_range_encode.restype = C.c_size_t
L.wr_range_encode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p]
L.wr_range_decode.restype = C.c_size_t
L.wr_range_decode.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
L.wr_range_encode_bound.restype = C.c_size_t
L.wr_range_encode_bound.argtypes = [C.c_size_t]
rs = np.random.RandomState(1)
# shaped like the bit planes of a smooth field: one dominant symbol, two symbols, then noise
u = rs.random_sample(n)
planes = {"dominant symbol p=0.9997 (~0.01 bit)": np.where(u < 0.9997, 121, rs.randint(0, 256, n)),
          "two symbols .74/.26 (~0.9 bit)": np.where(u < 0.737, 189, np.where(u < 0.994, 190, rs.randint(0, 256, n))),
          "uniform (8 bit)": rs.randint(0, 256, n), "centred normal sigma=12 (~5.6 bit)": np.clip(np.rint(rs.normal(128, 12, n)), 0, 255)}
for name, p in planes.items():
    p = p.astype(np.uint8)
    out = np.empty(L.wr_range_encode_bound(n), np.uint8)
    back = np.empty(n, np.uint8)
    best_e = best_d = 1e9
    for rep in range(2):
        t = time.time(); m = L.wr_range_encode(p.ctypes.data, n, out.ctypes.data); best_e = min(best_e, time.time() - t)
        t = time.time(); got = L.wr_range_decode(out.ctypes.data, m, back.ctypes.data, n); best_d = min(best_d, time.time() - t)
    assert got == n and np.array_equal(back, p)
    print("%-36s ratio %5.2f  encode %6.1f Msym/s  decode %6.1f Msym/s" % (name, n / m, n / best_e / 1e6, n / best_d / 1e6))

# interleaved: the four planes above, `count` at a time, on one thread
L.wr_range_encode_multi.restype = None
L.wr_range_encode_multi.argtypes = [C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p]
L.wr_range_decode_multi.restype = None
L.wr_range_decode_multi.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]
ps = [p.astype(np.uint8) for p in planes.values()]
outs = [np.empty(L.wr_range_encode_bound(n), np.uint8) for _ in ps]
backs = [np.empty(n, np.uint8) for _ in ps]
for count in (1, 2, 3, 4):
    ptr = lambda arrs: (C.c_void_p * count)(*[a.ctypes.data for a in arrs[:count]])  # noqa: E731
    lens = (C.c_size_t * count)()
    got = (C.c_size_t * count)()
    best_e = best_d = 1e9
    for rep in range(2):
        t = time.time(); L.wr_range_encode_multi(count, ptr(ps), n, ptr(outs), lens); best_e = min(best_e, time.time() - t)
        t = time.time(); L.wr_range_decode_multi(count, ptr(outs), lens, ptr(backs), n, got); best_d = min(best_d, time.time() - t)
    assert all(got[k] == n and np.array_equal(backs[k], ps[k]) for k in range(count))
    print("%d planes interleaved on one thread: encode %6.1f Msym/s  decode %6.1f Msym/s (aggregate)"
          % (count, count * n / best_e / 1e6, count * n / best_d / 1e6))
