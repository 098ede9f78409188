#!/usr/bin/env python3
"""How long the GPU takes to reach its working clocks after an idle gap: forward transforms back to back
after `idle` seconds of nothing, each timed with HIP events.  usage: clock_ramp.py [n] [idle] [count] [rounds]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
idle = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
count = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rounds = int(sys.argv[4]) if len(sys.argv) > 4 else 3
api.set_verbosity(0)
ctx = api.Context(0)
buf = ctx.alloc(n ** 3 * 8)
ctx.synth_field(buf, n, n, n, 12345)
ctx.sync()
for r in range(rounds):
    time.sleep(idle)
    t0 = time.perf_counter()
    ms = [ctx.bench_transform(buf, (n, n, n), 4, 1) for _ in range(count)]
    print("after %.2f s idle: fwd ms" % idle, " ".join("%.2f" % m for m in ms), "| wall %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    time.sleep(idle)
    ms = [ctx.bench_transform(buf, (n, n, n), -4, 1) for _ in range(count)]
    print("after %.2f s idle: inv ms" % idle, " ".join("%.2f" % m for m in ms), flush=True)
ctx.close()
