"""The drop-in symbols called from plain C (tests/native/shim_test.c) on the GPU, compared with the oracle:
  * the Fortran shims setup_wr_f / encoding_wrap_f / decoding_wrap_f (reference src/core/wrappers.cpp:545-594,
    call pattern of examples/fortran/example_fort.f90:74-121): by-pointer scalars, signed long arrays of 8;
  * two threads inside encoding_wrap / decoding_wrap at the same time on their own buffers (the reference
    is re-entrant on distinct buffers, SURVEY.md 8b "Threading");
  * the residual the reference leaves in fld_1d (wrappers.cpp:397-398)."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from util import ROOT, bits_equal
from waverange_amd import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def shim_exe(tmp_path_factory):
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    lib = os.path.join(ROOT, "waverange_amd")
    exe = str(tmp_path_factory.mktemp("shim") / "shim_test")
    subprocess.check_call(["gcc", "-O2", "-std=gnu99", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "native", "shim_test.c"), "-o", exe, "-L" + lib, "-lwaverange",
                           "-lpthread", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib"])
    return exe


def read_records(path, n):
    raw = open(path, "rb").read()
    recs, pos = [], 0
    while pos < len(raw):
        s = np.frombuffer(raw, np.float64, 3, pos); pos += 24
        u = np.frombuffer(raw, np.uint64, 3, pos); pos += 24
        deps = np.frombuffer(raw, np.float64, 8, pos); pos += 64
        mins = np.frombuffer(raw, np.float64, 8, pos); pos += 64
        lens = np.frombuffer(raw, np.int64, 8, pos); pos += 64
        ne = int(u[2])
        data = np.frombuffer(raw, np.uint8, ne, pos); pos += ne
        resid = np.frombuffer(raw, np.float64, n, pos); pos += 8 * n
        rec = np.frombuffer(raw, np.float64, n, pos); pos += 8 * n
        recs.append(dict(tolabs=s[0], midval=s[1], halfspanval=s[2], wlev=int(u[0]), nlay=int(u[1]), ntot_enc=ne,
                         deps_vec=deps, minval_vec=mins, len_enc_vec=lens, data=data, residual=resid, rec=rec))
    return recs


def check(r, f, tol, oracle):
    want = oracle.encode(f, tol)
    L = want["nlay"]
    for k in ("tolabs", "midval", "halfspanval", "wlev", "nlay", "ntot_enc"):
        assert r[k] == want[k], k
    assert bits_equal(r["deps_vec"][:L], want["deps_vec"]) and bits_equal(r["minval_vec"][:L], want["minval_vec"])
    assert list(r["len_enc_vec"][:L]) == want["len_enc_vec"] and not r["len_enc_vec"][L:].any()
    assert np.array_equal(r["data"], want["data"])
    assert bits_equal(r["residual"], want["residual"]), "fld_1d after encoding_wrap is not the reference's residual"
    assert bits_equal(r["rec"], oracle.decode(want, f.shape))


def test_fortran_shims_by_pointer(shim_exe, oracle, tmp_path):
    f = synth.field(40, 36, 28, seed=31)
    f.tofile(tmp_path / "in.bin")
    env = dict(os.environ, WR_QUIET="1")
    subprocess.run([shim_exe, "f", str(tmp_path / "in.bin"), "40", "36", "28", "1e-6", str(tmp_path / "out.bin")],
                   check=True, env=env)
    (r,) = read_records(tmp_path / "out.bin", f.size)
    check(r, f, 1e-6, oracle)


def test_two_threads_in_the_drop_in_symbols(shim_exe, oracle, tmp_path):
    fa = synth.field(64, 64, 64, seed=12345)   # fused transform path
    fb = synth.field(37, 21, 13, seed=9)       # general path, odd sizes
    fa.tofile(tmp_path / "a.bin")
    fb.tofile(tmp_path / "b.bin")
    env = dict(os.environ, WR_QUIET="1")
    reps = 6
    subprocess.run([shim_exe, "t", str(tmp_path / "a.bin"), "64", "64", "64", "1e-7", str(tmp_path / "a.out"),
                    str(tmp_path / "b.bin"), "37", "21", "13", "1e-4", str(tmp_path / "b.out"), str(reps)], check=True, env=env)
    ra, rb = read_records(tmp_path / "a.out", fa.size), read_records(tmp_path / "b.out", fb.size)
    assert len(ra) == reps and len(rb) == reps
    for r in ra:
        check(r, fa, 1e-7, oracle)
    for r in rb:
        check(r, fb, 1e-4, oracle)


def test_python_threads_in_the_drop_in_symbols(oracle):
    """The same through ctypes (which releases the GIL): four threads, mixed shapes, encoding_wrap +
    decoding_wrap + waveletcdf97_3d on host arrays."""
    import threading
    from waverange_amd import api
    api.set_verbosity(0)
    jobs = [((64, 64, 64), 1e-5, 1), ((48, 40, 24), 1e-6, 2), ((96, 64, 64), 1e-3, 3), ((33, 5, 1), 1e-4, 4)]
    errors = []

    def worker(shape, tol, seed):
        try:
            f = synth.field(*shape, seed=seed)
            want = oracle.encode(f, tol)
            rec = oracle.decode(want, f.shape)
            fw = oracle.cdf97_3d(f, 4)
            for _ in range(3):
                enc = api.encoding_wrap(f, tol)
                assert np.array_equal(enc["data"], want["data"]) and enc["len_enc_vec"] == want["len_enc_vec"]
                assert bits_equal(enc["residual"], want["residual"])
                assert bits_equal(api.decoding_wrap(enc, f.shape), rec)
                assert bits_equal(api.waveletcdf97_3d(f, 4), fw)
        except Exception as exc:  # noqa: BLE001
            errors.append((shape, exc))

    ths = [threading.Thread(target=worker, args=j) for j in jobs]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    assert not errors, errors
