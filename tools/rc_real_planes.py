#!/usr/bin/env python3
"""The coder loops on the bench field's OWN planes: the field of bench.py (synthetic generator, seed 12345) is transformed and
quantized on the GPU (wr_dev_encode_planes), its planes come to the host, and every plane kind goes through the loops of the
pool on one thread: the 16-lane decoder loop (16 and 8 copies), the scalar decoder loop of four, the 16-lane encoder loop.
Tells what each plane of a field costs a worker, and how often the 16-lane decoder has to leave its candidates.
usage: rc_real_planes.py [n = 512]   (one GPU)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
nelem = n ** 3
ctx = api.Context(0)
pitch = api.lib().wr_plane_pitch(nelem)
fld = ctx.alloc(nelem * 8)
planes_dev = ctx.alloc(pitch * 8)


def best_of(fn, reps=2):
    best = 1e9
    for _ in range(reps):
        t = time.time(); fn(); best = min(best, time.time() - t)
    return best


for tol in (1e-3, 1e-7):
    ctx.synth_field(fld, n, n, n, 12345)
    info = ctx.encode_planes(fld, (n, n, n), tol, planes_dev)
    for l in range(info.nlay):
        q = planes_dev.download(np.uint8, nelem, offset=l * pitch)
        blocks = q[: nelem // 60000 * 60000].reshape(-1, 60000)
        # share of the block held by its four most frequent symbols, over the blocks
        cover = []
        for b in blocks[:: max(1, len(blocks) // 200)]:
            h = np.bincount(b, minlength=256)
            cover.append(np.sort(h)[-4:].sum() / 60000.0)
        cover = np.array(cover)
        s = api.range_encode(q)
        bits = 8.0 * s.size / nelem
        line = "tol %g plane %d: %.3f bits/symbol, four most frequent symbols hold %.2f %% of a block (min %.2f %%)" % (
            tol, l, bits, 100 * cover.mean(), 100 * cover.min())
        if bits < 2:
            for k in (16, 8):
                dt = best_of(lambda: api.range_decode_vec([s] * k, [nelem] * k))
                line += " | vector decoder, %d lanes: %.0f Msym/s per thread" % (k, k * nelem / dt / 1e6)
        dt = best_of(lambda: api.range_decode_multi([s] * 4, nelem))
        line += " | scalar decoder, 4 planes: %.0f" % (4 * nelem / dt / 1e6)
        dt = best_of(lambda: api.range_encode_vec([q] * 16))
        line += " | vector encoder, 16 lanes (CPU histograms included): %.0f" % (16 * nelem / dt / 1e6)
        print(line, flush=True)
