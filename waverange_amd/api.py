"""ctypes binding of libwaverange_amd.so (include/waverange_amd.h).

Host-side mirror of the reference's codec interface (src/core/wrappers.h): the module-level
``setup_wr`` / ``encoding_wrap`` / ``decoding_wrap`` take numpy arrays and return the same
quantities the reference writes through its reference parameters.  ``Context`` exposes the
device-resident entry points used by the parity tests and bench.py.

There is no CPU fallback: loading fails loudly when the HIP library has not been built, and
every compute call fails loudly when no GPU is usable.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# WAVERANGE_AMD_LIB points at another build of the library (A/B runs of experimental builds)
LIB_PATH = os.environ.get("WAVERANGE_AMD_LIB") or os.path.join(HERE, "libwaverange_amd.so")
NLAYMAX = 8

_vp = C.c_void_p
_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_ubyte)
_ulp = C.POINTER(C.c_ulong)


class EncInfo(C.Structure):
    """wr_enc_info: the per-field header record (reference .wrh fields)."""
    _fields_ = [("tolabs", C.c_double), ("midval", C.c_double), ("halfspanval", C.c_double),
                ("wlev", C.c_ubyte), ("nlay", C.c_ubyte), ("ntot_enc", C.c_ulong),
                ("deps_vec", C.c_double * NLAYMAX), ("minval_vec", C.c_double * NLAYMAX),
                ("len_enc_vec", C.c_ulong * NLAYMAX)]

    def as_dict(self):
        L = self.nlay
        return dict(tolabs=self.tolabs, midval=self.midval, halfspanval=self.halfspanval,
                    wlev=self.wlev, nlay=L, ntot_enc=self.ntot_enc,
                    deps_vec=np.array(self.deps_vec[:L]), minval_vec=np.array(self.minval_vec[:L]),
                    len_enc_vec=[int(v) for v in self.len_enc_vec[:L]])

    @classmethod
    def from_dict(cls, d):
        s = cls()
        s.tolabs, s.midval, s.halfspanval = d.get("tolabs", 0.0), d["midval"], d.get("halfspanval", 0.0)
        s.wlev, s.nlay, s.ntot_enc = d["wlev"], d["nlay"], d["ntot_enc"]
        for i in range(d["nlay"]):
            s.deps_vec[i] = d["deps_vec"][i]
            s.minval_vec[i] = d["minval_vec"][i]
            s.len_enc_vec[i] = d["len_enc_vec"][i]
        return s


class Timings(C.Structure):
    _fields_ = [("total", C.c_double), ("gpu", C.c_double), ("transfer", C.c_double),
                ("rangecoder", C.c_double), ("transform_ms", C.c_float), ("quant_ms", C.c_float),
                ("minmax_ms", C.c_float), ("wait", C.c_double), ("h2d_ms", C.c_float), ("d2h_ms", C.c_float),
                ("plane_coder_s", C.c_double * NLAYMAX)]

    def as_dict(self):
        return {k: (list(getattr(self, k)) if k == "plane_coder_s" else getattr(self, k)) for k, _ in self._fields_}


class WaveRangeError(RuntimeError):
    pass


_lib = None


def lib():
    """Load the shared library (once).  Raises if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise WaveRangeError(
            "%s is missing: build it with `python -m waverange_amd.build` (hipcc, gfx950); "
            "there is no CPU fallback" % LIB_PATH)
    L = C.CDLL(LIB_PATH)
    L.wr_last_error.restype = C.c_char_p
    L.wr_ctx_create.argtypes = [C.POINTER(_vp), C.c_int, _vp]
    L.wr_ctx_destroy.argtypes = [_vp]
    L.wr_ctx_sync.argtypes = [_vp]
    L.wr_ctx_set_keep_residual.argtypes = [_vp, C.c_int]
    L.wr_dev_alloc.argtypes = [_vp, C.POINTER(_vp), C.c_size_t]
    L.wr_dev_free.argtypes = [_vp, _vp]
    L.wr_dev_upload.argtypes = [_vp, _vp, _vp, C.c_size_t]
    L.wr_dev_download.argtypes = [_vp, _vp, _vp, C.c_size_t]
    L.wr_dev_copy.argtypes = [_vp, _vp, _vp, C.c_size_t]
    L.wr_dev_copy_kernel.argtypes = [_vp, _vp, _vp, C.c_size_t, C.c_int]
    L.wr_dev_linf.argtypes = [_vp, _vp, _vp, C.c_size_t, _dp, _dp]
    L.wr_dev_transform.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.wr_dev_minmax.argtypes = [_vp, _vp, C.c_size_t, _dp, _dp]
    L.wr_dev_quantize_plane.argtypes = [_vp, _vp, C.c_size_t, C.c_double, C.c_double, _vp, _dp, _dp]
    L.wr_dev_dequant_accum.argtypes = [_vp, _vp, C.c_size_t, C.c_int, C.POINTER(_vp), _dp, _dp]
    L.wr_dev_synth_field.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_ulonglong]
    L.wr_plane_pitch.restype = C.c_size_t
    L.wr_plane_pitch.argtypes = [C.c_size_t]
    L.wr_dev_encode_planes.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double, _vp,
                                       C.POINTER(EncInfo)]
    L.wr_dev_decode_planes.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, _vp, C.POINTER(EncInfo)]
    L.wr_encode_device.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_double,
                                   C.POINTER(EncInfo), _vp, C.c_size_t, C.POINTER(Timings)]
    L.wr_encode_device_local.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp,
                                         C.POINTER(EncInfo), _vp, C.c_size_t, C.POINTER(Timings)]
    L.wr_decode_device.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(EncInfo), _vp, C.c_size_t,
                                   C.POINTER(Timings)]
    L.wr_encode_host.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp,
                                 C.POINTER(EncInfo), _vp, C.c_size_t, C.POINTER(Timings)]
    L.wr_decode_host.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.POINTER(EncInfo), _vp, C.c_size_t,
                                 C.POINTER(Timings)]
    L.wr_transform_host.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int]
    L.wr_decode_begin.argtypes = [_vp, C.c_int, C.c_int, C.c_int, C.POINTER(EncInfo), _vp, C.c_size_t, C.POINTER(Timings)]
    L.wr_decode_finish_host.argtypes = [_vp, _vp, C.POINTER(Timings)]
    L.wr_decode_finish_device.argtypes = [_vp, _vp, C.POINTER(Timings)]
    L.wr_host_alloc.argtypes = [C.POINTER(_vp), C.c_size_t]
    L.wr_host_free.argtypes = [_vp]
    L.wr_host_register.argtypes = [_vp, C.c_size_t]
    L.wr_autotune_batch.argtypes = [C.c_size_t, C.c_int]
    L.wr_test_stale_window.argtypes = [_vp, C.c_size_t]
    L.wr_host_unregister.argtypes = [_vp]
    L.wr_set_device_slots.argtypes = [C.c_int, C.c_int]
    L.wr_set_writeback_residual.argtypes = [C.c_int]
    L.wr_set_coder_pool.argtypes = [C.c_int, C.c_int]
    L.wr_stat.restype = C.c_ulong
    L.wr_stat.argtypes = [C.c_int]
    L.wr_pool_loop_stats.restype = None
    L.wr_pool_loop_stats.argtypes = [_vp, _vp]
    L.wr_range_encode_bound.restype = C.c_size_t
    L.wr_range_encode_bound.argtypes = [C.c_size_t]
    L.wr_range_encode.restype = C.c_size_t
    L.wr_range_encode.argtypes = [_vp, C.c_size_t, _vp]
    L.wr_range_decode.restype = C.c_size_t
    L.wr_range_decode.argtypes = [_vp, C.c_size_t, _vp, C.c_size_t]
    L.wr_range_encode_multi.restype = None
    L.wr_range_encode_multi.argtypes = [C.c_int, _vp, C.c_size_t, _vp, _vp]
    L.wr_ctx_trim.restype = C.c_int
    L.wr_ctx_trim.argtypes = [C.c_void_p]
    L.wr_range_decode_multi.restype = None
    L.wr_range_decode_multi.argtypes = [C.c_int, _vp, _vp, _vp, C.c_size_t, _vp]
    L.wr_range_encode_pool.argtypes = [C.c_int, _vp, _vp, _vp, _vp]
    L.wr_range_decode_pool.argtypes = [C.c_int, _vp, _vp, _vp, _vp, _vp]
    L.wr_range_decode_vec.argtypes = [C.c_int, _vp, _vp, _vp, _vp, _vp]
    L.wr_range_encode_vec.argtypes = [C.c_int, _vp, _vp, _vp, _vp]
    L.wr_range_encode_windowed.argtypes = [C.c_int, C.c_int, _vp, C.c_size_t, C.c_size_t, _vp, _vp]
    L.wr_range_decode_windowed.argtypes = [C.c_int, C.c_int, _vp, _vp, _vp, C.c_size_t, C.c_size_t, _vp]
    L.wr_bench_transform.argtypes = [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _dp]
    # drop-in symbols (reference src/core/wrappers.h:53,70,75)
    L.setup_wr.argtypes = [C.c_int] * 3 + [_u8p, _ulp]
    L.encoding_wrap.argtypes = [C.c_int] * 3 + [_dp] + [C.c_int] * 4 + [_dp] + [_dp] * 3 + [
        _u8p, _u8p, _ulp, _dp, _dp, _ulp, _u8p]
    L.decoding_wrap.argtypes = [C.c_int] * 3 + [_dp] + [_dp] * 3 + [_u8p, _u8p, _ulp, _dp, _dp, _ulp, _u8p]
    L.waveletcdf97_3d.argtypes = [C.c_int] * 4 + [_dp]
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        raise WaveRangeError("libwaverange_amd error %d: %s" % (rc, lib().wr_last_error().decode()))


def device_count():
    return lib().wr_device_count()


def set_verbosity(level):
    lib().wr_set_verbosity(int(level))


def set_threads(n, encoder=0):
    """coder threads per call; `encoder` > 0 gives the encoder its own count"""
    lib().wr_set_threads(int(n))
    lib().wr_set_encoder_threads(int(encoder))


def set_coder_pool(nthreads, decoder_streams=0):
    """process-wide coder pool (wr_set_coder_pool): nthreads workers code the planes of all concurrent calls;
    0 stops it"""
    lib().wr_set_coder_pool(int(nthreads), int(decoder_streams))


def set_device_slots(device, nslots):
    _check(lib().wr_set_device_slots(int(device), int(nslots)))


STAT_EARLY_DECODES, STAT_SLOTS_POPULATED, STAT_DEVICE_PLANE_BYTES, STAT_POOL_IDLE_MS, STAT_POOL_STREAMS_MOVED = 0, 1, 2, 3, 4
STAT_POOL_QUEUE_MS, STAT_PLANE_WAIT_MS, STAT_HANDOVER_ERRORS, STAT_CLOCK_WARMUP_MS, STAT_DECODE_GATE_MS = 5, 6, 7, 8, 9
STAT_WINDOW_WAIT_MS = 10


def stat(what):
    return lib().wr_stat(int(what))


def pool_loop_stats():
    """{loop kind: (worker seconds in block steps, stream-blocks advanced)} since the process started."""
    sec, blk = (C.c_double * 4)(), (C.c_double * 4)()
    lib().wr_pool_loop_stats(sec, blk)
    return {k: (sec[i], blk[i]) for i, k in enumerate(("scalar_encoder", "scalar_decoder", "vector_decoder", "vector_encoder"))}


def set_writeback_residual(on):
    lib().wr_set_writeback_residual(int(on))


class _Pinned:
    def __init__(self, nbytes):
        p = _vp()
        _check(lib().wr_host_alloc(C.byref(p), nbytes))
        self.ptr, self.nbytes = p.value, nbytes

    def __del__(self):
        if self.ptr and _lib is not None:
            _lib.wr_host_free(self.ptr)
            self.ptr = None


def pinned_array(shape, dtype=np.float64):
    """numpy array on pinned host memory (wr_host_alloc): moves over PCIe by DMA without a staging copy.
    The memory is released when the last view of the array is gone."""
    dt = np.dtype(dtype)
    count = int(np.prod(shape))
    owner = _Pinned(max(16, count * dt.itemsize))
    buf = (C.c_ubyte * owner.nbytes).from_address(owner.ptr)
    buf._owner = owner  # keeps the allocation alive as long as the ctypes buffer (numpy's base)
    return np.frombuffer(buf, dtype=dt, count=count).reshape(shape)


class registered:
    """`with api.registered(arr):` pins a numpy array the caller owns for the duration of the block (wr_host_register)."""

    def __init__(self, arr):
        self.arr = arr

    def __enter__(self):
        _check(lib().wr_host_register(self.arr.ctypes.data, self.arr.nbytes))
        return self.arr

    def __exit__(self, *exc):
        _check(lib().wr_host_unregister(self.arr.ctypes.data))
        return False


def range_encode_windowed(planes, chunk, mode=0):
    """Equally long planes coded through windows of `chunk` symbols (test hook of the device-resident plane path).
    mode 0: interleaved loops on this thread, 1: the coder pool, 2: the 16-lane loops."""
    ps = [np.ascontiguousarray(p, dtype=np.uint8).ravel() for p in planes]
    k, n = len(ps), ps[0].size
    assert all(p.size == n for p in ps)
    outs = [np.empty(lib().wr_range_encode_bound(n), dtype=np.uint8) for _ in ps]
    lens = (C.c_size_t * k)()
    _check(lib().wr_range_encode_windowed(mode, k, (C.c_void_p * k)(*[p.ctypes.data for p in ps]), n, chunk,
                                          (C.c_void_p * k)(*[o.ctypes.data for o in outs]), lens))
    return [o[:lens[i]].copy() for i, o in enumerate(outs)]


def range_decode_windowed(streams, n, chunk, mode=0):
    ss = [np.ascontiguousarray(s, dtype=np.uint8).ravel() for s in streams]
    k = len(ss)
    outs = [np.zeros(max(n, 1), dtype=np.uint8) for _ in ss]
    got = (C.c_size_t * k)()
    _check(lib().wr_range_decode_windowed(mode, k, (C.c_void_p * k)(*[s.ctypes.data for s in ss]), (C.c_size_t * k)(*[s.size for s in ss]),
                                          (C.c_void_p * k)(*[o.ctypes.data for o in outs]), n, chunk, got))
    return [o[:n] for o in outs], [got[i] for i in range(k)]


# ---------------------------------------------------------------------------------------
# host range coder (product code, runs without a GPU)
# ---------------------------------------------------------------------------------------
def range_encode(plane):
    p = np.ascontiguousarray(plane, dtype=np.uint8).ravel()
    out = np.empty(lib().wr_range_encode_bound(p.size), dtype=np.uint8)
    n = lib().wr_range_encode(p.ctypes.data, p.size, out.ctypes.data)
    return out[:n].copy()


def range_encode_bound_hist(plane):
    """wr_range_encode_bound_hist on the plane's own per-block histograms (counted here with numpy)."""
    p = np.ascontiguousarray(plane, dtype=np.uint8).ravel()
    nb = p.size // 60000 + 1
    hist = np.zeros((nb, 256), dtype=np.uint16)
    for b in range(nb):
        hist[b] = np.bincount(p[b * 60000:(b + 1) * 60000], minlength=256)
    lib().wr_range_encode_bound_hist.restype = C.c_size_t
    lib().wr_range_encode_bound_hist.argtypes = [C.c_void_p, C.c_size_t]
    return int(lib().wr_range_encode_bound_hist(hist.ctypes.data, p.size))


def range_decode(stream, n):
    s = np.ascontiguousarray(stream, dtype=np.uint8).ravel()
    out = np.zeros(max(n, 1), dtype=np.uint8)
    got = lib().wr_range_decode(s.ctypes.data, s.size, out.ctypes.data, n)
    return out[:n], got


def range_encode_multi(planes):
    """Code several equally long planes on this thread with interleaved symbol loops (wr_range_encode_multi)."""
    ps = [np.ascontiguousarray(p, dtype=np.uint8).ravel() for p in planes]
    n, k = ps[0].size, len(ps)
    assert all(p.size == n for p in ps)
    outs = [np.empty(lib().wr_range_encode_bound(n), dtype=np.uint8) for _ in ps]
    lens = (C.c_size_t * k)()
    lib().wr_range_encode_multi(k, (C.c_void_p * k)(*[p.ctypes.data for p in ps]), n,
                                (C.c_void_p * k)(*[o.ctypes.data for o in outs]), lens)
    return [o[:lens[i]].copy() for i, o in enumerate(outs)]


def range_decode_multi(streams, n):
    ss = [np.ascontiguousarray(s, dtype=np.uint8).ravel() for s in streams]
    k = len(ss)
    outs = [np.zeros(max(n, 1), dtype=np.uint8) for _ in ss]
    got = (C.c_size_t * k)()
    lib().wr_range_decode_multi(k, (C.c_void_p * k)(*[s.ctypes.data for s in ss]), (C.c_size_t * k)(*[s.size for s in ss]),
                                (C.c_void_p * k)(*[o.ctypes.data for o in outs]), n, got)
    return [o[:n] for o in outs], [got[i] for i in range(k)]


def range_encode_pool(planes):
    """Planes of any lengths through the coder pool (set_coder_pool first)."""
    ps = [np.ascontiguousarray(p, dtype=np.uint8).ravel() for p in planes]
    k = len(ps)
    outs = [np.empty(lib().wr_range_encode_bound(p.size), dtype=np.uint8) for p in ps]
    lens = (C.c_size_t * k)()
    _check(lib().wr_range_encode_pool(k, (C.c_void_p * k)(*[p.ctypes.data for p in ps]), (C.c_size_t * k)(*[p.size for p in ps]),
                                      (C.c_void_p * k)(*[o.ctypes.data for o in outs]), lens))
    return [o[:lens[i]].copy() for i, o in enumerate(outs)]


def range_decode_pool(streams, ns):
    ss = [np.ascontiguousarray(s, dtype=np.uint8).ravel() for s in streams]
    k = len(ss)
    outs = [np.zeros(max(n, 1), dtype=np.uint8) for n in ns]
    got = (C.c_size_t * k)()
    _check(lib().wr_range_decode_pool(k, (C.c_void_p * k)(*[s.ctypes.data for s in ss]), (C.c_size_t * k)(*[s.size for s in ss]),
                                      (C.c_void_p * k)(*[o.ctypes.data for o in outs]), (C.c_size_t * k)(*ns), got))
    return [o[:n] for o, n in zip(outs, ns)], [got[i] for i in range(k)]


def range_encode_vec(planes):
    """Planes of any kind and length through the 16-lane AVX-512 encoder loop on this thread."""
    ps = [np.ascontiguousarray(p, dtype=np.uint8).ravel() for p in planes]
    k = len(ps)
    outs = [np.empty(lib().wr_range_encode_bound(p.size), dtype=np.uint8) for p in ps]
    lens = (C.c_size_t * k)()
    _check(lib().wr_range_encode_vec(k, (C.c_void_p * k)(*[p.ctypes.data for p in ps]), (C.c_size_t * k)(*[p.size for p in ps]),
                                     (C.c_void_p * k)(*[o.ctypes.data for o in outs]), lens))
    return [o[:lens[i]].copy() for i, o in enumerate(outs)]


def range_decode_vec(streams, ns):
    """Planes through the 16-lane AVX-512 decoder loop on this thread (raises on CPUs without AVX-512)."""
    ss = [np.ascontiguousarray(s, dtype=np.uint8).ravel() for s in streams]
    k = len(ss)
    outs = [np.zeros(max(n, 1), dtype=np.uint8) for n in ns]
    got = (C.c_size_t * k)()
    _check(lib().wr_range_decode_vec(k, (C.c_void_p * k)(*[s.ctypes.data for s in ss]), (C.c_size_t * k)(*[s.size for s in ss]),
                                     (C.c_void_p * k)(*[o.ctypes.data for o in outs]), (C.c_size_t * k)(*ns), got))
    return [o[:n] for o, n in zip(outs, ns)], [got[i] for i in range(k)]


# ---------------------------------------------------------------------------------------
# drop-in interface on host arrays (reference src/core/wrappers.h)
# ---------------------------------------------------------------------------------------
def setup_wr(nx, ny, nz):
    nl, cap = C.c_ubyte(), C.c_ulong()
    lib().setup_wr(nx, ny, nz, C.byref(nl), C.byref(cap))
    return nl.value, cap.value


def encoding_wrap(fld, tolrel, wtflag=1):
    """fld: float64 array shaped (nz, ny, nx).  Returns the reference's outputs as a dict
    (tolabs, midval, halfspanval, wlev, nlay, ntot_enc, deps_vec, minval_vec, len_enc_vec, data)."""
    fld = np.ascontiguousarray(fld, dtype=np.float64)
    nz, ny, nx = fld.shape
    work = fld.copy()
    _, cap = setup_wr(nx, ny, nz)
    data = np.empty(cap, dtype=np.uint8)
    cut = np.array([tolrel], dtype=np.float64)
    tolabs, midval, halfspan = C.c_double(), C.c_double(), C.c_double()
    wlev, nlay, ntot_enc = C.c_ubyte(), C.c_ubyte(), C.c_ulong()
    deps, mins = np.zeros(NLAYMAX), np.zeros(NLAYMAX)
    lens = np.zeros(NLAYMAX, dtype=np.uint64)
    p = lambda a, t: a.ctypes.data_as(t)  # noqa: E731
    lib().encoding_wrap(nx, ny, nz, p(work, _dp), wtflag, 1, 1, 1, p(cut, _dp), C.byref(tolabs),
                        C.byref(midval), C.byref(halfspan), C.byref(wlev), C.byref(nlay),
                        C.byref(ntot_enc), p(deps, _dp), p(mins, _dp), p(lens, _ulp), p(data, _u8p))
    L = nlay.value
    return dict(tolabs=tolabs.value, midval=midval.value, halfspanval=halfspan.value, wlev=wlev.value,
                nlay=L, ntot_enc=ntot_enc.value, deps_vec=deps[:L].copy(), minval_vec=mins[:L].copy(),
                len_enc_vec=[int(v) for v in lens[:L]], data=data[:ntot_enc.value].copy(), residual=work)


def decoding_wrap(enc, shape):
    nz, ny, nx = shape
    out = np.empty(nz * ny * nx, dtype=np.float64)
    deps, mins = np.zeros(NLAYMAX), np.zeros(NLAYMAX)
    lens = np.zeros(NLAYMAX, dtype=np.uint64)
    L = enc["nlay"]
    deps[:L], mins[:L], lens[:L] = enc["deps_vec"], enc["minval_vec"], enc["len_enc_vec"]
    data = np.ascontiguousarray(enc["data"], dtype=np.uint8)
    if data.size == 0:
        data = np.zeros(1, dtype=np.uint8)
    tolabs, midval, halfspan = C.c_double(enc.get("tolabs", 0.0)), C.c_double(enc["midval"]), \
        C.c_double(enc.get("halfspanval", 0.0))
    wlev, nlay, ntot_enc = C.c_ubyte(enc["wlev"]), C.c_ubyte(L), C.c_ulong(enc["ntot_enc"])
    p = lambda a, t: a.ctypes.data_as(t)  # noqa: E731
    lib().decoding_wrap(nx, ny, nz, p(out, _dp), C.byref(tolabs), C.byref(midval), C.byref(halfspan),
                        C.byref(wlev), C.byref(nlay), C.byref(ntot_enc), p(deps, _dp), p(mins, _dp),
                        p(lens, _ulp), p(data, _u8p))
    return out.reshape(shape)


def waveletcdf97_3d(x, lvl):
    y = np.ascontiguousarray(x, dtype=np.float64).copy()
    nz, ny, nx = y.shape
    lib().waveletcdf97_3d(nx, ny, nz, lvl, y.ctypes.data_as(_dp))
    return y


# ---------------------------------------------------------------------------------------
# device-resident interface
# ---------------------------------------------------------------------------------------
class DeviceBuffer:
    def __init__(self, ctx, nbytes):
        self.ctx, self.nbytes = ctx, nbytes
        p = _vp()
        _check(lib().wr_dev_alloc(ctx.h, C.byref(p), nbytes))
        self.ptr = p.value

    def upload(self, a):
        a = np.ascontiguousarray(a)
        assert a.nbytes <= self.nbytes
        _check(lib().wr_dev_upload(self.ctx.h, self.ptr, a.ctypes.data, a.nbytes))
        return self

    def download(self, dtype, count, offset=0):
        out = np.empty(count, dtype=dtype)
        _check(lib().wr_dev_download(self.ctx.h, out.ctypes.data, self.ptr + offset, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().wr_dev_free(self.ctx.h, self.ptr)
            self.ptr = None


class Context:
    """One GPU context (device + stream + work space): wr_ctx of include/waverange_amd.h."""

    def __init__(self, device=0, stream=None):
        h = _vp()
        _check(lib().wr_ctx_create(C.byref(h), device, stream))
        self.h = h

    def close(self):
        if self.h:
            lib().wr_ctx_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def alloc(self, nbytes):
        return DeviceBuffer(self, nbytes)

    def to_device(self, a):
        a = np.ascontiguousarray(a)
        return self.alloc(max(a.nbytes, 16)).upload(a)

    def sync(self):
        _check(lib().wr_ctx_sync(self.h))

    def trim(self):
        """idle buffers of the device's plane pool go back to the device (wr_ctx_trim)"""
        _check(lib().wr_ctx_trim(self.h))

    def set_keep_residual(self, keep):
        lib().wr_ctx_set_keep_residual(self.h, int(keep))

    def copy(self, dst, src, nbytes):
        _check(lib().wr_dev_copy(self.h, dst.ptr, src.ptr, nbytes))

    def linf(self, a, b, n):
        """(max|a-b|, max|a|) over n doubles."""
        d, m = C.c_double(), C.c_double()
        _check(lib().wr_dev_linf(self.h, a.ptr, b.ptr, n, C.byref(d), C.byref(m)))
        return d.value, m.value

    def transform(self, buf, shape, lvl):
        nz, ny, nx = shape
        _check(lib().wr_dev_transform(self.h, buf.ptr, nx, ny, nz, lvl))

    def minmax(self, buf, n):
        a, b = C.c_double(), C.c_double()
        _check(lib().wr_dev_minmax(self.h, buf.ptr, n, C.byref(a), C.byref(b)))
        return a.value, b.value

    def quantize_plane(self, buf, n, deps, minval, qbuf):
        a, b = C.c_double(), C.c_double()
        _check(lib().wr_dev_quantize_plane(self.h, buf.ptr, n, deps, minval, qbuf.ptr, C.byref(a), C.byref(b)))
        return a.value, b.value

    def dequant_accum(self, acc, n, planes, deps, minval):
        arr = (_vp * len(planes))(*[p if isinstance(p, int) else p.ptr for p in planes])
        d = np.ascontiguousarray(deps, dtype=np.float64)
        m = np.ascontiguousarray(minval, dtype=np.float64)
        _check(lib().wr_dev_dequant_accum(self.h, acc.ptr, n, len(planes), arr, d.ctypes.data_as(_dp),
                                          m.ctypes.data_as(_dp)))

    def synth_field(self, buf, nx, ny, nz, seed):
        _check(lib().wr_dev_synth_field(self.h, buf.ptr, nx, ny, nz, seed))

    def encode_planes(self, buf, shape, tolrel, planes, wtflag=1):
        nz, ny, nx = shape
        info = EncInfo()
        _check(lib().wr_dev_encode_planes(self.h, buf.ptr, nx, ny, nz, wtflag, tolrel, planes.ptr, C.byref(info)))
        return info

    def decode_planes(self, buf, shape, planes, info):
        nz, ny, nx = shape
        _check(lib().wr_dev_decode_planes(self.h, buf.ptr, nx, ny, nz, planes.ptr, C.byref(info)))

    def encode(self, buf, shape, tolrel, wtflag=1, out=None):
        """Whole encode with the field resident on the device.  `buf` is consumed: it holds the residual in wavelet space
        afterwards if set_keep_residual(True) was called on the context, otherwise its contents are unspecified.  Returns
        (info dict incl. data, timings)."""
        nz, ny, nx = shape
        _, cap = setup_wr(nx, ny, nz)
        data = out if out is not None else np.empty(cap, dtype=np.uint8)
        info, tm = EncInfo(), Timings()
        _check(lib().wr_encode_device(self.h, buf.ptr, nx, ny, nz, wtflag, tolrel, C.byref(info),
                                      data.ctypes.data, data.size, C.byref(tm)))
        d = info.as_dict()
        d["data"] = data[:info.ntot_enc]
        return d, tm.as_dict()

    def encode_local(self, buf, shape, cutoff, m, wtflag=1):
        """Encode with the reference's non-uniform cutoff: cutoff has mx*my*mz entries, m = (mx, my, mz)."""
        nz, ny, nx = shape
        _, cap = setup_wr(nx, ny, nz)
        data = np.empty(cap, dtype=np.uint8)
        cut = np.ascontiguousarray(cutoff, dtype=np.float64)
        info, tm = EncInfo(), Timings()
        _check(lib().wr_encode_device_local(self.h, buf.ptr, nx, ny, nz, wtflag, m[0], m[1], m[2],
                                            cut.ctypes.data_as(_dp), C.byref(info), data.ctypes.data, data.size,
                                            C.byref(tm)))
        d = info.as_dict()
        d["data"] = data[:info.ntot_enc].copy()
        return d, tm.as_dict()

    def decode(self, buf, shape, enc):
        nz, ny, nx = shape
        info = EncInfo.from_dict(enc)
        tm = Timings()
        data = np.ascontiguousarray(enc["data"], dtype=np.uint8)
        if data.size == 0:
            data = np.zeros(1, dtype=np.uint8)
        _check(lib().wr_decode_device(self.h, buf.ptr, nx, ny, nz, C.byref(info), data.ctypes.data, data.size,
                                      C.byref(tm)))
        return tm.as_dict()

    # ---- host buffer to host buffer (what encoding_wrap / decoding_wrap run on)
    def encode_host(self, fld, tolrel, wtflag=1, out=None, cutoff=None, m=(1, 1, 1)):
        """fld: C-contiguous float64 array shaped (nz, ny, nx), pinned (pinned_array) or pageable; left
        untouched unless set_keep_residual(True).  Returns (info dict incl. data, timings)."""
        assert fld.dtype == np.float64 and fld.flags["C_CONTIGUOUS"]
        nz, ny, nx = fld.shape
        _, cap = setup_wr(nx, ny, nz)
        data = out if out is not None else np.empty(cap, dtype=np.uint8)
        cut = np.ascontiguousarray([tolrel] if cutoff is None else cutoff, dtype=np.float64)
        info, tm = EncInfo(), Timings()
        _check(lib().wr_encode_host(self.h, fld.ctypes.data, nx, ny, nz, wtflag, m[0], m[1], m[2],
                                    cut.ctypes.data_as(_dp), C.byref(info), data.ctypes.data, data.size, C.byref(tm)))
        d = info.as_dict()
        d["data"] = data[:info.ntot_enc]
        return d, tm.as_dict()

    def decode_host(self, out, enc):
        """out: C-contiguous float64 array shaped (nz, ny, nx) that receives the reconstruction."""
        assert out.dtype == np.float64 and out.flags["C_CONTIGUOUS"]
        nz, ny, nx = out.shape
        info = EncInfo.from_dict(enc)
        tm = Timings()
        data = np.ascontiguousarray(enc["data"], dtype=np.uint8)
        if data.size == 0:
            data = np.zeros(1, dtype=np.uint8)
        _check(lib().wr_decode_host(self.h, out.ctypes.data, nx, ny, nz, C.byref(info), data.ctypes.data, data.size,
                                    C.byref(tm)))
        return tm.as_dict()

    def decode_begin(self, shape, enc):
        """Host half of a decode (range decoding into the context's staging); no output buffer needed yet."""
        nz, ny, nx = shape
        info = EncInfo.from_dict(enc)
        tm = Timings()
        data = np.ascontiguousarray(enc["data"], dtype=np.uint8)
        if data.size == 0:
            data = np.zeros(1, dtype=np.uint8)
        _check(lib().wr_decode_begin(self.h, nx, ny, nz, C.byref(info), data.ctypes.data, data.size, C.byref(tm)))
        return tm.as_dict()

    def decode_finish_host(self, out):
        """Device half of the decode begun on this context: upload, kernels, download into `out`."""
        assert out.dtype == np.float64 and out.flags["C_CONTIGUOUS"]
        tm = Timings()
        _check(lib().wr_decode_finish_host(self.h, out.ctypes.data, C.byref(tm)))
        return tm.as_dict()

    def decode_finish(self, buf):
        tm = Timings()
        _check(lib().wr_decode_finish_device(self.h, buf.ptr, C.byref(tm)))
        return tm.as_dict()

    def burn(self, ms, mode=0, workgroups=1024):
        lib().wr_dev_burn.argtypes = [_vp, C.c_double, C.c_int, C.c_int]
        _check(lib().wr_dev_burn(self.h, ms, mode, workgroups))

    def test_stale_window(self, n):
        """Test hook: the window handle of a finished call against the plane of the next one (must be refused)."""
        _check(lib().wr_test_stale_window(self.h, n))

    def transform_host(self, fld, lvl):
        nz, ny, nx = fld.shape
        _check(lib().wr_transform_host(self.h, fld.ctypes.data, nx, ny, nz, lvl))

    def bench_transform(self, buf, shape, lvl, reps=1):
        nz, ny, nx = shape
        ms = C.c_double()
        _check(lib().wr_bench_transform(self.h, buf.ptr, nx, ny, nz, lvl, reps, C.byref(ms)))
        return ms.value
