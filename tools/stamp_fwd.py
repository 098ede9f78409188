#!/usr/bin/env python3
"""Diagnostic: per-phase cycle shares of the fused forward level-0 kernel.
Build with WR_CXXFLAGS=-DWR_STAMP python -m waverange_amd.build --force, then run this."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
L = api.lib()
L.wr_stamp_buffer.restype = C.c_void_p
L.wr_stamp_buffer.argtypes = [C.c_size_t]
nw = 256 * 8 * 8
api.set_verbosity(0)
ctx = api.Context(0)
ptr = L.wr_stamp_buffer(nw)
buf = ctx.alloc(n ** 3 * 8)
ctx.synth_field(buf, n, n, n, 12345)
ctx.sync()
inv = len(sys.argv) > 2 and sys.argv[2] == "inv"
for r in range(3):
    print("ms", ctx.bench_transform(buf, (n, n, n), -4 if inv else 4, 1))
out = np.zeros(nw * 8, dtype=np.uint64)
L.wr_dev_download(ctx.h, out.ctypes.data, ptr, out.nbytes)
a = out.reshape(-1, 8).astype(np.float64)
a = a[a.sum(axis=1) > 0] / 3.0   # three launches accumulated? no: overwritten each launch
names = ["wait_dma", "barrier_top", "xlift", "fetch_issue", "ylift", "zstep+stores", "barriers_mid", "loop_top"]
if inv:
    names = ["wait_dma", "barrier_top", "zstep(+zb write)", "fetch_issue", "y+x stages", "odd plane -> zb", "barriers_mid", "loop_top"]
tot = a.sum(axis=1).mean()
print("waves with stamps:", len(a), "mean cycles/wave:", tot * 3)
for i, nm in enumerate(names):
    print("%-14s %6.1f %%" % (nm, 100 * a[:, i].mean() / tot))
