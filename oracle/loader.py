"""ctypes loaders for the CPU oracle and for the compiled reference.

TEST INFRASTRUCTURE ONLY (see oracle/wr_oracle.h): imported by tests/, by
__graft_entry__.smoke() and by bench.py's cpu_baseline leg, never by the product
package ``waverange_amd``.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ORACLE_SO = os.path.join(HERE, "libwr_oracle.so")
REF_SO = os.path.join(HERE, "_ref", "libwaverange_ref.so")
REF_WRENC = os.path.join(HERE, "_ref", "wrenc_ref")
REF_WRDEC = os.path.join(HERE, "_ref", "wrdec_ref")

NLAYMAX = 8
BLOCKSIZE = 60000

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_ubyte)
_ulp = C.POINTER(C.c_ulong)


def build(ref=True):
    """(Re)build the oracle and, when the reference sources are present, oracle/_ref."""
    subprocess.check_call(["make", "-s", "-C", HERE, "oracle"])
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.check_call(["make", "-s", "-C", HERE, "ref", "cross"])


def have_ref():
    return os.path.exists(REF_SO)


def _p(a, t):
    return a.ctypes.data_as(t)


class Oracle:
    """The plain-C restatement (oracle/wr_oracle.c)."""

    def __init__(self):
        if not os.path.exists(ORACLE_SO):
            build(ref=False)
        L = self.lib = C.CDLL(ORACLE_SO)
        L.wro_cdf97_3d.argtypes = [C.c_int] * 4 + [_dp]
        L.wro_range_encode.argtypes = [_u8p, C.c_size_t, _u8p]
        L.wro_range_encode.restype = C.c_size_t
        L.wro_range_decode.argtypes = [_u8p, C.c_size_t, _u8p, C.c_size_t]
        L.wro_range_decode.restype = C.c_size_t
        L.wro_minmax.argtypes = [_dp, C.c_size_t, _dp, _dp]
        L.wro_quantize_plane.argtypes = [_dp, C.c_size_t, C.c_double, C.c_double, _u8p]
        L.wro_dequant_accum.argtypes = [_dp, C.c_size_t, _u8p, C.c_double, C.c_double]
        L.wro_encode.argtypes = [C.c_int] * 3 + [_dp] + [C.c_int] * 4 + [_dp] + [_dp] * 3 + [
            _u8p, _u8p, _ulp, _dp, _dp, _ulp, _u8p]
        L.wro_decode.argtypes = [C.c_int] * 3 + [_dp, C.c_double, C.c_ubyte, C.c_ubyte, C.c_ulong,
                                                 _dp, _dp, _ulp, _u8p]
        L.wro_ind_p2w_3d.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int)] * 4

    def cdf97_3d(self, x, lvl):
        """x: float64 array shaped (nz, ny, nx) (x fastest); returns a transformed copy."""
        y = np.ascontiguousarray(x, dtype=np.float64).copy()
        nz, ny, nx = y.shape
        self.lib.wro_cdf97_3d(nx, ny, nz, lvl, _p(y, _dp))
        return y

    def range_encode(self, sym):
        sym = np.ascontiguousarray(sym, dtype=np.uint8)
        out = np.empty(2 * max(1024, sym.size), dtype=np.uint8)
        n = self.lib.wro_range_encode(_p(sym, _u8p), sym.size, _p(out, _u8p))
        return out[:n].copy()

    def range_decode(self, stream, n):
        stream = np.ascontiguousarray(stream, dtype=np.uint8)
        out = np.zeros(max(n, 1), dtype=np.uint8)
        got = self.lib.wro_range_decode(_p(stream, _u8p), stream.size, _p(out, _u8p), n)
        return out[:n], got

    def minmax(self, x):
        x = np.ascontiguousarray(x, dtype=np.float64).ravel()
        a, b = C.c_double(), C.c_double()
        self.lib.wro_minmax(_p(x, _dp), x.size, C.byref(a), C.byref(b))
        return a.value, b.value

    def quantize_plane(self, x, deps, minval):
        r = np.ascontiguousarray(x, dtype=np.float64).ravel().copy()
        q = np.empty(r.size, dtype=np.uint8)
        self.lib.wro_quantize_plane(_p(r, _dp), r.size, deps, minval, _p(q, _u8p))
        return q, r

    def dequant_accum(self, acc, q, deps, minval):
        acc = np.ascontiguousarray(acc, dtype=np.float64).ravel().copy()
        q = np.ascontiguousarray(q, dtype=np.uint8).ravel()
        self.lib.wro_dequant_accum(_p(acc, _dp), acc.size, _p(q, _u8p), deps, minval)
        return acc

    def ind_p2w(self, lvl, n1, n2, n3, i1, i2, i3):
        o = [C.c_int() for _ in range(4)]
        self.lib.wro_ind_p2w_3d(lvl, n1, n2, n3, i1, i2, i3, *[C.byref(v) for v in o])
        return tuple(v.value for v in o)

    def encode(self, fld, tol, wtflag=1, cutoff=None, m=(1, 1, 1)):
        """fld shaped (nz, ny, nx).  Returns a dict with every output of encoding_wrap."""
        return _encode_common(self.lib.wro_encode, fld, tol, wtflag, cutoff, m, byref=False)

    def decode(self, enc, shape):
        nz, ny, nx = shape
        out = np.empty(nz * ny * nx, dtype=np.float64)
        deps = np.zeros(NLAYMAX)
        mins = np.zeros(NLAYMAX)
        lens = np.zeros(NLAYMAX, dtype=np.uint64)
        deps[:len(enc["deps_vec"])] = enc["deps_vec"]
        mins[:len(enc["minval_vec"])] = enc["minval_vec"]
        lens[:len(enc["len_enc_vec"])] = enc["len_enc_vec"]
        data = np.ascontiguousarray(enc["data"], dtype=np.uint8)
        if data.size == 0:
            data = np.zeros(1, dtype=np.uint8)
        self.lib.wro_decode(nx, ny, nz, _p(out, _dp), enc["midval"], enc["wlev"], enc["nlay"],
                            enc["ntot_enc"], _p(deps, _dp), _p(mins, _dp), _p(lens, _ulp),
                            _p(data, _u8p))
        return out.reshape(shape)


def _encode_common(fn, fld, tol, wtflag, cutoff, m, byref):
    fld = np.ascontiguousarray(fld, dtype=np.float64)
    nz, ny, nx = fld.shape
    work = fld.copy()
    n = work.size
    data = np.empty(NLAYMAX * max(1024, n), dtype=np.uint8)
    cut = np.array([tol] if cutoff is None else cutoff, dtype=np.float64)
    tolabs, midval, halfspan = C.c_double(), C.c_double(), C.c_double()
    wlev, nlay = C.c_ubyte(), C.c_ubyte()
    ntot_enc = C.c_ulong()
    deps = np.zeros(NLAYMAX)
    mins = np.zeros(NLAYMAX)
    lens = np.zeros(NLAYMAX, dtype=np.uint64)
    fn(nx, ny, nz, _p(work, _dp), wtflag, m[0], m[1], m[2], _p(cut, _dp), C.byref(tolabs),
       C.byref(midval), C.byref(halfspan), C.byref(wlev), C.byref(nlay), C.byref(ntot_enc),
       _p(deps, _dp), _p(mins, _dp), _p(lens, _ulp), _p(data, _u8p))
    L = nlay.value
    return dict(tolabs=tolabs.value, midval=midval.value, halfspanval=halfspan.value,
                wlev=wlev.value, nlay=L, ntot_enc=ntot_enc.value, deps_vec=deps[:L].copy(),
                minval_vec=mins[:L].copy(), len_enc_vec=[int(v) for v in lens[:L]],
                data=data[:ntot_enc.value].copy(), residual=work)


class Reference:
    """The reference itself (oracle/_ref/libwaverange_ref.so), strict-IEEE build.

    Its C++ reference parameters are plain pointers at the ABI level
    (src/core/wrappers.h:53,70,75)."""

    def __init__(self):
        if not have_ref():
            raise FileNotFoundError(REF_SO)
        L = self.lib = C.CDLL(REF_SO)
        L.waveletcdf97_3d.argtypes = [C.c_int] * 4 + [_dp]
        L.encoding_wrap.argtypes = [C.c_int] * 3 + [_dp] + [C.c_int] * 4 + [_dp] + [_dp] * 3 + [
            _u8p, _u8p, _ulp, _dp, _dp, _ulp, _u8p]
        L.decoding_wrap.argtypes = [C.c_int] * 3 + [_dp] + [_dp] * 3 + [_u8p, _u8p, _ulp, _dp,
                                                                        _dp, _ulp, _u8p]
        L.ind_p2w_3d.argtypes = [C.c_int] * 7 + [C.POINTER(C.c_int)] * 4

    def cdf97_3d(self, x, lvl):
        y = np.ascontiguousarray(x, dtype=np.float64).copy()
        nz, ny, nx = y.shape
        self.lib.waveletcdf97_3d(nx, ny, nz, lvl, _p(y, _dp))
        return y

    def ind_p2w(self, lvl, n1, n2, n3, i1, i2, i3):
        o = [C.c_int() for _ in range(4)]
        self.lib.ind_p2w_3d(lvl, n1, n2, n3, i1, i2, i3, *[C.byref(v) for v in o])
        return tuple(v.value for v in o)

    def encode(self, fld, tol, wtflag=1, cutoff=None, m=(1, 1, 1)):
        return _encode_common(self.lib.encoding_wrap, fld, tol, wtflag, cutoff, m, byref=True)

    def decode(self, enc, shape):
        nz, ny, nx = shape
        out = np.empty(nz * ny * nx, dtype=np.float64)
        deps = np.zeros(NLAYMAX)
        mins = np.zeros(NLAYMAX)
        lens = np.zeros(NLAYMAX, dtype=np.uint64)
        deps[:len(enc["deps_vec"])] = enc["deps_vec"]
        mins[:len(enc["minval_vec"])] = enc["minval_vec"]
        lens[:len(enc["len_enc_vec"])] = enc["len_enc_vec"]
        data = np.ascontiguousarray(enc["data"], dtype=np.uint8)
        if data.size == 0:
            data = np.zeros(1, dtype=np.uint8)
        tolabs, midval, halfspan = C.c_double(enc["tolabs"]), C.c_double(enc["midval"]), \
            C.c_double(enc["halfspanval"])
        wlev, nlay = C.c_ubyte(enc["wlev"]), C.c_ubyte(enc["nlay"])
        ntot_enc = C.c_ulong(enc["ntot_enc"])
        self.lib.decoding_wrap(nx, ny, nz, _p(out, _dp), C.byref(tolabs), C.byref(midval),
                               C.byref(halfspan), C.byref(wlev), C.byref(nlay),
                               C.byref(ntot_enc), _p(deps, _dp), _p(mins, _dp), _p(lens, _ulp),
                               _p(data, _u8p))
        return out.reshape(shape)
