import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
from waverange_amd import api, synth
api.set_verbosity(0)
n = 512
f = synth.field(n, n, n)
for rep in range(2):
    t = time.time(); enc = api.encoding_wrap(f, 1e-5); te = time.time() - t
    t = time.time(); rec = api.decoding_wrap(enc, f.shape); td = time.time() - t
    print("host-pointer API 512^3: encode %.3f s (%.0f MB/s)  decode %.3f s (%.0f MB/s)  [includes numpy copy of the field + 8N-byte output buffer]" % (te, f.nbytes/1e6/te, td, f.nbytes/1e6/td))
import ctypes as C
L = api.lib()
ctx = api.Context(0)
buf = ctx.alloc(f.nbytes)
for rep in range(3):
    t = time.time(); L.wr_dev_upload(ctx.h, buf.ptr, f.ctypes.data, f.nbytes); tu = time.time() - t
    out = np.empty_like(f)
    t = time.time(); L.wr_dev_download(ctx.h, out.ctypes.data, buf.ptr, f.nbytes); tdn = time.time() - t
    print("pageable H2D %.1f GB/s  D2H %.1f GB/s" % (f.nbytes/1e9/tu, f.nbytes/1e9/tdn))
