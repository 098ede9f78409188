"""Shared helpers for the parity tests."""
import hashlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def bits_equal(a, b):
    a = np.ascontiguousarray(a, dtype=np.float64).ravel()
    b = np.ascontiguousarray(b, dtype=np.float64).ravel()
    return a.shape == b.shape and np.array_equal(a.view(np.uint64), b.view(np.uint64))


def hexes(v):
    return [float(x).hex() for x in v]


def kat_plane(kind, n):
    """Same planes as tools/make_golden.py::kat_plane (guarded by plane_sha256 in golden.json)."""
    from waverange_amd import synth
    idx = np.arange(n, dtype=np.uint64)
    h = synth.splitmix64(777, idx)
    if kind == "uniform":
        b = (h & np.uint64(0xFF)).astype(np.uint8)
    elif kind == "skewed":
        b = np.minimum((h & np.uint64(0xFF)), (h >> np.uint64(8)) & np.uint64(0xFF))
        b = np.minimum(b, (h >> np.uint64(16)) & np.uint64(0xFF)).astype(np.uint8)
    elif kind == "sparse":
        b = np.zeros(n, dtype=np.uint8)
        b[:: max(1, n // 37)] = 3
    else:
        raise ValueError(kind)
    b = b.copy()
    b[0], b[-1] = 0, 255
    return b


def check_enc_record(e, rec, what=""):
    """Compare an encode result dict (oracle.loader format) with a golden.json record."""
    assert float(e["tolabs"]).hex() == rec["tolabs"], what
    assert float(e["midval"]).hex() == rec["midval"], what
    assert float(e["halfspanval"]).hex() == rec["halfspanval"], what
    assert e["wlev"] == rec["wlev"] and e["nlay"] == rec["nlay"], what
    assert hexes(e["deps_vec"]) == rec["deps_vec"], what
    assert hexes(e["minval_vec"]) == rec["minval_vec"], what
    assert list(e["len_enc_vec"]) == rec["len_enc_vec"], what
    assert e["ntot_enc"] == rec["ntot_enc"], what
    assert sha(e["data"]) == rec["data_sha256"], what


def sha_big(a, chunk=1 << 28):
    """SHA-256 of an array's bytes without a copy of the whole array (fields of gigabytes)."""
    h = hashlib.sha256()
    b = np.ascontiguousarray(a).reshape(-1).view(np.uint8)
    for o in range(0, b.size, chunk):
        h.update(b[o:o + chunk])
    return h.hexdigest()


def large_golden():
    """tests/golden/large.json: outputs of the compiled reference at 512^3 and 1024^3 (tools/make_golden_large.py)."""
    import json
    with open(os.path.join(GOLDEN, "large.json")) as fh:
        return json.load(fh)


def check_large_record(e, rec, what=""):
    """An encode result (dict with the encoding_wrap outputs and `data`) against a large.json record: header scalars as
    bit patterns, plane lengths, SHA-256 of every plane stream and of all coded bytes."""
    assert e["nlay"] == rec["nlay"] and e["wlev"] == rec["wlev"], what
    assert list(int(v) for v in e["len_enc_vec"]) == rec["len_enc_vec"], (what, list(e["len_enc_vec"]), rec["len_enc_vec"])
    assert int(e["ntot_enc"]) == rec["ntot_enc"], what
    for k in ("tolabs", "midval", "halfspanval"):
        assert float(e[k]).hex() == rec[k], (what, k)
    assert hexes(e["deps_vec"]) == rec["deps_vec"], what
    assert hexes(e["minval_vec"]) == rec["minval_vec"], what
    data = np.ascontiguousarray(e["data"]).reshape(-1)[:rec["ntot_enc"]]
    off = 0
    for l, n in enumerate(rec["len_enc_vec"]):
        assert sha_big(data[off:off + n]) == rec["plane_sha256"][l], (what, "coded bytes of plane %d differ from the reference's" % l)
        off += n
    assert sha_big(data) == rec["data_sha256"], what


def g7_field(rec):
    """The reference's Fortran-example field (examples/fortran/example_fort.f90:82-91) rebuilt from the 1-D factors the
    fixture carries: ((10 sin x) (sin y)^2) cos z, multiplied in the example's order."""
    sx = np.array([float.fromhex(v) for v in rec["sin_factor"]])
    cz = np.array([float.fromhex(v) for v in rec["cos_factor"]])
    f = ((10.0 * sx)[None, None, :] * (sx * sx)[None, :, None]) * cz[:, None, None]
    assert sha(f) == rec["input_sha256"]
    return f
