#!/usr/bin/env python3
"""One context, one n^3 synthetic field: encode + decode `reps` times (for rocprofv3 --kernel-trace --stats:
quantizer, histogram, dequantizer and transform kernels without other contexts in the way).
usage: prof_codec.py [n] [tol] [reps]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-7
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
api.set_verbosity(0)
ctx = api.Context(0)
orig = ctx.alloc(n ** 3 * 8)
work = ctx.alloc(n ** 3 * 8)
ctx.synth_field(orig, n, n, n, 12345)
ctx.sync()
for _ in range(reps):
    ctx.copy(work, orig, n ** 3 * 8)
    enc, te = ctx.encode(work, (n, n, n), tol)
    td = ctx.decode(work, (n, n, n), enc)
    print("encode", {k: round(v, 4) for k, v in te.items()})
    print("decode", {k: round(v, 4) for k, v in td.items()})
diff, amax = ctx.linf(orig, work, n ** 3)
print("linf_rel", diff / amax)
ctx.close()
