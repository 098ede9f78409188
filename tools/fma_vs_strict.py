#!/usr/bin/env python3
"""SURVEY.md 8c "report both comparisons": the reference AS SHIPPED (its own flags: -march=native, gcc contracts a*b+c into
fused multiply-adds; oracle/_ref_fma, `make -C oracle ref_fma`) against the strict-IEEE build every parity claim of this
repository is made against (oracle/_ref, -ffp-contract=off, otherwise the same flags) -- on the inputs and tolerances of
the G1 golden vectors, on the 512^3 bench field, and on G7: the field of the reference's Fortran example
(examples/fortran/example_fort.f90:82-91, evaluated in C: tools/native/g7_field.c) at its tolrel 1e-6.
Informational, CPU only, build container only; writes Markdown to stdout:

    python tools/fma_vs_strict.py > profiles/r03/fma_vs_strict.md"""
import ctypes as C
import hashlib
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import loader  # noqa: E402
from waverange_amd import synth  # noqa: E402


class Fma(loader.Reference):
    def __init__(self):
        so = os.path.join(ROOT, "oracle", "_ref_fma", "libwaverange_ref_fma.so")
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref_fma"])
        L = self.lib = C.CDLL(so)
        dp, u8p, ulp = loader._dp, loader._u8p, loader._ulp
        L.encoding_wrap.argtypes = [C.c_int] * 3 + [dp] + [C.c_int] * 4 + [dp] + [dp] * 3 + [u8p, u8p, ulp, dp, dp, ulp, u8p]
        L.decoding_wrap.argtypes = [C.c_int] * 3 + [dp] + [dp] * 3 + [u8p, u8p, ulp, dp, dp, ulp, u8p]
        L.waveletcdf97_3d.argtypes = [C.c_int] * 4 + [dp]


def quiet(fn, *a, **k):
    fd = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    sys.stdout.flush()
    os.dup2(devnull, 1)
    try:
        return fn(*a, **k)
    finally:
        sys.stdout.flush()
        os.dup2(fd, 1)
        os.close(devnull)
        os.close(fd)


def digits_equal(a, b):
    """leading significant decimal digits two doubles share"""
    if a == b:
        return 17
    if a == 0 or b == 0:
        return 0
    return max(0, int(np.floor(-np.log10(abs(a - b) / max(abs(a), abs(b))))))


def compare(strict, fma, f, tol, label):
    es, ef = quiet(strict.encode, f, tol), quiet(fma.encode, f, tol)
    rs, rf = quiet(strict.decode, es, f.shape), quiet(fma.decode, ef, f.shape)
    hdr = []
    for k in ("tolabs", "midval", "halfspanval"):
        if es[k] != ef[k]:
            hdr.append("%s (%d digits equal)" % (k, digits_equal(es[k], ef[k])))
    for k in ("deps_vec", "minval_vec"):
        n = min(len(es[k]), len(ef[k]))
        diff = [digits_equal(float(es[k][i]), float(ef[k][i])) for i in range(n) if es[k][i] != ef[k][i]]
        if diff or len(es[k]) != len(ef[k]):
            hdr.append("%s: %d of %d entries (>= %d digits equal)" % (k, len(diff), n, min(diff) if diff else 17))
    ds, df = es["data"], ef["data"]
    m = min(ds.size, df.size)
    nd = int(np.count_nonzero(ds[:m] != df[:m])) + abs(ds.size - df.size)
    first = int(np.argmax(ds[:m] != df[:m])) if nd and np.any(ds[:m] != df[:m]) else None
    amax = float(np.abs(f).max())
    return "| %s | %g | %d / %d | %s | %s | %d vs %d | %s | %.3g / %.3g | %.3g |" % (
        label, tol, es["nlay"], ef["nlay"], "identical" if not hdr else "; ".join(hdr), "identical" if es["len_enc_vec"] == ef["len_enc_vec"] else "differ",
        ds.size, df.size, "identical" if nd == 0 else "%d bytes differ (first at %s)" % (nd, first),
        np.abs(rs - f).max() / amax, np.abs(rf - f).max() / amax, np.abs(rs - rf).max() / amax)


def main():
    strict, fma = loader.Reference(), Fma()
    print("# As-shipped (FMA-contracted) reference build vs the strict-IEEE build\n")
    print("Written by `tools/fma_vs_strict.py` in the build container (%s, gcc %s)." % (
        [l.split(":", 1)[1].strip() for l in open("/proc/cpuinfo") if l.startswith("model name")][0],
        subprocess.check_output(["gcc", "-dumpfullversion"], text=True).strip()))
    print("`strict` = `oracle/_ref` (the reference's sources, its flags without `-march=native`, plus `-ffp-contract=off`): the build every parity")
    print("test of this repository is pinned to.  `as shipped` = `oracle/_ref_fma` (`config.mk:25,30` verbatim: `-O2 -ftree-vectorize ... -march=native`,")
    print("gcc's default `-ffp-contract=fast`): %s `vfm*` instructions in its `waveletcdf97_3d.o`.  The two differ only in rounding; which one a user gets"
          % subprocess.check_output("objdump -d %s | grep -c vfm" % os.path.join(ROOT, "oracle", "_ref_fma", "waveletcdf97_3d.o"), shell=True, text=True).strip())
    print("depends on the compiler and the host it was built on, so the as-shipped build is reported, not gated on.\n")
    print("| input | tol | nlay strict / shipped | header scalars | plane lengths | ntot_enc strict vs shipped | coded bytes | L-inf rel strict / shipped | max abs diff of the two reconstructions / max abs field |")
    print("|---|---|---|---|---|---|---|---|---|")
    f64 = synth.field(64, 64, 64, seed=12345)
    for tol in (1e-3, 1e-5, 1e-7, 1e-16):
        print(compare(strict, fma, f64, tol, "G1 field 64^3 (seed 12345)"))
    fodd = synth.field(37, 21, 13, seed=12345)
    print(compare(strict, fma, fodd, 1e-6, "odd 37x21x13 (seed 12345)"))
    f256 = synth.field(256, 256, 256, seed=12345)
    for tol in (1e-3, 1e-7):
        print(compare(strict, fma, f256, tol, "bench generator 256^3 (seed 12345)"))
    # G7
    with tempfile.TemporaryDirectory() as d:
        exe, raw = os.path.join(d, "g7"), os.path.join(d, "g7.raw")
        subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", os.path.join(ROOT, "tools", "native", "g7_field.c"), "-o", exe, "-lm"])
        subprocess.check_call([exe, "64", "64", "64", raw])
        g7 = np.fromfile(raw, dtype=np.float64).reshape(64, 64, 64)
    print(compare(strict, fma, g7, 1e-6, "G7 Fortran-example field 64^3"))
    print("\nG7 input: `tools/native/g7_field.c` (the formula of `examples/fortran/example_fort.f90:82-91` with this container's libm), SHA-256 of the 64^3 doubles `%s`;"
          % hashlib.sha256(g7.tobytes()).hexdigest())
    print("the example prints the relative L-inf error at `tolrel = 1e-6`: the strict column above is what this repository's GPU path reproduces bit for bit")
    print("(`tests/test_gpu_parity.py::test_fortran_example_field`).")
    # transform alone
    a, b = quiet(strict.cdf97_3d, f64, 4), quiet(fma.cdf97_3d, f64, 4)
    nd = int(np.count_nonzero(a.view(np.uint64) != b.view(np.uint64)))
    print("\nForward transform alone, 64^3: %d of %d coefficients differ in their bit patterns, largest difference %.3g relative to the largest coefficient."
          % (nd, a.size, float(np.abs(a - b).max() / np.abs(a).max())))


if __name__ == "__main__":
    main()
