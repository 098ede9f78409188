#!/usr/bin/env python3
"""Small driver for profiling: runs forward (and inverse) transforms of an n^3 synthetic field.
usage: prof_transform.py [n] [reps] [fwd|inv|both]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api
if os.environ.get('WR_AB_LIB'):  # A/B runs of an experimental build of the library
    api.LIB_PATH = os.path.abspath(os.environ['WR_AB_LIB'])

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
what = sys.argv[3] if len(sys.argv) > 3 else "both"
api.set_verbosity(0)
ctx = api.Context(0)
buf = ctx.alloc(n ** 3 * 8)
ctx.synth_field(buf, n, n, n, 12345)
ctx.sync()
for r in range(reps):
    if what in ("fwd", "both"):
        print("fwd ms", ctx.bench_transform(buf, (n, n, n), 4, 1))
    if what in ("inv", "both"):
        print("inv ms", ctx.bench_transform(buf, (n, n, n), -4, 1))
ctx.close()
