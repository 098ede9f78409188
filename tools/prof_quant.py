#!/usr/bin/env python3
"""Quantizer plane kernel alone: wall time of wr_dev_quantize_plane on an n^3 field (kernel + residual min/max + one
host round trip).  usage: prof_quant.py [n] [reps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
api.set_verbosity(0)
ctx = api.Context(0)
buf = ctx.alloc(n ** 3 * 8)
q = ctx.alloc(n ** 3 + 4096)
ctx.synth_field(buf, n, n, n, 12345)
ctx.sync()
lo, hi = ctx.minmax(buf, n ** 3)
ts = []
for r in range(reps):
    deps = (hi - lo) / 255.0
    t = time.perf_counter()
    lo, hi = ctx.quantize_plane(buf, n ** 3, deps, lo, q)
    ts.append((time.perf_counter() - t) * 1e3)
print("quantize_plane ms:", " ".join("%.2f" % v for v in ts), "| min %.3f" % min(ts))
ctx.close()
