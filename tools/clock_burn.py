#!/usr/bin/env python3
"""What brings the shader clock up before a kernel stage: after `idle` seconds of nothing, a burner kernel of X ms (mode 0:
fp64 arithmetic on every CU, mode 1: every CU occupied by sleeping waves) and then forward transforms back to back, each
timed with HIP events.  usage: clock_burn.py [n] [idle]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
api.set_verbosity(0)
ctx = api.Context(0)
buf = ctx.alloc(n ** 3 * 8)
ctx.synth_field(buf, n, n, n, 12345)
ctx.sync()
for lvl, name in ((4, "fwd"), (-4, "inv")):
    for mode, wgs in ((0, 1024), (1, 1024), (0, 256), (1, 256)):
        for ms in (0, 5, 10, 20, 40, 80):
            if ms == 0 and (mode, wgs) != (0, 1024):
                continue
            time.sleep(idle)
            if ms:
                ctx.burn(ms, mode, wgs)
            t = [ctx.bench_transform(buf, (n, n, n), lvl, 1) for _ in range(4)]
            print("%s after %.2f s idle + burner(mode %d, %4d workgroups, %2d ms):" % (name, idle, mode, wgs, ms), " ".join("%.2f" % m for m in t), flush=True)
ctx.close()
