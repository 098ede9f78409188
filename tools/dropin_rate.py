#!/usr/bin/env python3
"""What one caller of the drop-in symbols gets: encoding_wrap / decoding_wrap on ordinary (pageable) host arrays, the
way the reference's CLI calls them -- upload, kernels, planes back, host range coder with a thread per plane,
residual write-back included.  usage: dropin_rate.py [n] [tol]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from waverange_amd import api, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
tol = float(sys.argv[2]) if len(sys.argv) > 2 else 1e-5
api.set_verbosity(0)
with api.Context(0) as ctx:   # the synthetic field comes from the device generator (numpy would take minutes at 1024^3)
    buf = ctx.alloc(n ** 3 * 8)
    ctx.synth_field(buf, n, n, n, 12345)
    ctx.sync()
    f = buf.download(np.float64, n ** 3).reshape(n, n, n)
    buf.free()
mb = f.nbytes / 1e6
for wb in (1, 0):
    api.set_writeback_residual(wb)
    for rep in range(2):
        t = time.time(); enc = api.encoding_wrap(f, tol); te = time.time() - t   # includes the wrapper's copy of the field
        t = time.time(); rec = api.decoding_wrap(enc, f.shape); td = time.time() - t
    print("n=%d tol=%g residual write-back %d: encoding_wrap %.3f s (%.0f MB/s)  decoding_wrap %.3f s (%.0f MB/s)  both %.0f MB/s  planes %d ratio %.2f"
          % (n, tol, wb, te, mb / te, td, mb / td, mb / (te + td), enc["nlay"], f.nbytes / enc["ntot_enc"]), flush=True)
