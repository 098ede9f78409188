#!/usr/bin/env python3
"""Host coder rates on the REAL bit planes of the synthetic field (quantized on the GPU with
wr_dev_encode_planes, then coded on one host thread): every plane alone, then all planes of the field
interleaved.  usage: rc_speed_real.py [n] [tol]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
api.set_verbosity(0)
ctx = api.Context(0)
buf = ctx.alloc(n ** 3 * 8)
ctx.synth_field(buf, n, n, n, 12345)
pitch = api.lib().wr_plane_pitch(n ** 3)
dplanes = ctx.alloc(pitch * 8)
info = ctx.encode_planes(buf, (n, n, n), tol, dplanes)
planes = [dplanes.download(np.uint8, n ** 3, offset=l * pitch) for l in range(info.nlay)]
ctx.close()
N = n ** 3
for l, p in enumerate(planes):
    h = np.bincount(p, minlength=256)
    srt = np.sort(h)[::-1]
    t = time.time(); s = api.range_encode(p); te = time.time() - t
    t = time.time(); api.range_decode(s, N); td = time.time() - t
    print("plane %d: %.3f bit/sym, p1 %.4f p2 %.4f, %3d symbols  encode %6.1f Msym/s  decode %6.1f Msym/s"
          % (l, 8.0 * s.size / N, srt[0] / N, srt[1] / N, (h > 0).sum(), N / te / 1e6, N / td / 1e6))
t = time.time(); ss = api.range_encode_multi(planes); te = time.time() - t
t = time.time(); api.range_decode_multi(ss, N); td = time.time() - t
print("all %d planes interleaved on one thread: encode %6.1f Msym/s  decode %6.1f Msym/s (aggregate; the encoder counts its own histograms here)"
      % (len(planes), len(planes) * N / te / 1e6, len(planes) * N / td / 1e6))

# the 16-lane AVX-512 loops on the real dominant-symbol planes (16 copies of plane 0 / plane 1 alternating), one thread
dom = [p for l, p in enumerate(planes) if 8.0 * api.range_encode(p).size / N < 2.0]
if dom:
    ps = [dom[i % len(dom)] for i in range(16)]
    blocks = N // 60000
    for l, p in enumerate(dom):  # share of the blocks the vector encoder takes (top four symbols hold >= 99 %)
        q = p[:blocks * 60000].reshape(blocks, 60000)
        ok = 0
        for b in range(0, blocks, max(1, blocks // 200)):
            h = np.sort(np.bincount(q[b], minlength=256))[::-1]
            ok += h[:4].sum() * 100 >= 60000 * 99
        print("dominant plane %d: %.0f %% of the sampled blocks are held >= 99 %% by four symbols" % (l, 100.0 * ok / len(range(0, blocks, max(1, blocks // 200)))))
    try:
        t = time.time(); venc = api.range_encode_vec(ps); te = time.time() - t
        t = time.time(); api.range_decode_vec(venc, [N] * 16); td = time.time() - t
        t = time.time(); api.range_encode_multi(ps[:3]); t3 = time.time() - t
        print("16 dominant-symbol planes in the AVX-512 loops: encode %6.1f Msym/s  decode %6.1f Msym/s per thread; scalar encoder loop of three: %6.1f"
              % (16 * N / te / 1e6, 16 * N / td / 1e6, 3 * N / t3 / 1e6))
    except api.WaveRangeError as exc:
        print("AVX-512 loops:", exc)
