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
