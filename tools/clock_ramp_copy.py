#!/usr/bin/env python3
"""Does a PCIe upload done by a copy KERNEL (a few workgroups busy for 150 ms) bring the clocks up for the transform
that follows, where an SDMA upload leaves the GPU idle?  usage: clock_ramp_copy.py [n] [workgroups]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
wgs = int(sys.argv[2]) if len(sys.argv) > 2 else 16
api.set_verbosity(0)
ctx = api.Context(0)
L = api.lib()
buf = ctx.alloc(n ** 3 * 8)
stage = ctx.alloc(n ** 3 * 8)
ctx.synth_field(buf, n, n, n, 12345)
ctx.sync()
host = api.pinned_array((n, n, n))
api._check(L.wr_dev_download(ctx.h, host.ctypes.data, buf.ptr, host.nbytes))
for mode in ("sdma", "kernel", "sdma", "kernel"):
    time.sleep(0.5)
    t0 = time.perf_counter()
    if mode == "sdma":
        api._check(L.wr_dev_upload(ctx.h, stage.ptr, host.ctypes.data, host.nbytes))
    else:
        api._check(L.wr_dev_copy_kernel(ctx.h, stage.ptr, host.ctypes.data, host.nbytes, wgs))
        ctx.sync()
    t_up = (time.perf_counter() - t0) * 1e3
    ms = [ctx.bench_transform(buf, (n, n, n), 4, 1) for _ in range(4)]
    print("upload by %-6s %.0f ms (%.1f GB/s), then fwd ms: %s" % (mode, t_up, host.nbytes / t_up / 1e6, " ".join("%.2f" % m for m in ms)), flush=True)
ctx.close()
