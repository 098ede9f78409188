#!/usr/bin/env python3
"""Small driver for profiling: runs forward (and inverse) transforms of an n^3 synthetic field.
usage: prof_transform.py [n] [reps] [fwd|inv|both] [idle seconds before every transform]"""
import sys
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
what = sys.argv[3] if len(sys.argv) > 3 else "both"
idle = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
import time
api.set_verbosity(0)
ctx = api.Context(0)
buf = ctx.alloc(n ** 3 * 8)
ctx.synth_field(buf, n, n, n, 12345)
ctx.sync()
for r in range(reps):
    time.sleep(idle)
    if what in ("fwd", "both"):
        print("fwd ms", ctx.bench_transform(buf, (n, n, n), 4, 1))
    time.sleep(idle)
    if what in ("inv", "both"):
        print("inv ms", ctx.bench_transform(buf, (n, n, n), -4, 1))
ctx.close()
