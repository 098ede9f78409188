"""Build libwaverange_amd.so (HIP kernels for gfx950 + host pipeline + C ABI) in-tree.

    python -m waverange_amd.build [--force]

hipcc cross-compiles without a GPU.  Strict IEEE arithmetic: -ffp-contract=off everywhere
(DESIGN.md "Arithmetic").  The .so stays in-tree (git-ignored) so that it travels with gpurun.
"""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libwaverange_amd.so")
ALIAS = os.path.join(HERE, "libwaverange.so")  # the reference's library name (drop-in link target)
BIN = os.path.join(HERE, "bin")

SOURCES = ["wr_kernels.hip", "wr_fused.hip", "wr_pipeline.cpp", "wr_codec.cpp", "wr_coder_hooks.cpp", "wr_dropin.cpp", "wr_dma.cpp", "wr_rangecoder.cpp", "wr_rangecoder_avx512.cpp", "wr_compat.cpp"]
# per-file extra flags: the AVX-512 coder loop is only entered when the CPU has the instructions (vec_available)
EXTRA = {"wr_rangecoder_avx512.cpp": ["-mavx512f", "-mavx512bw", "-mavx512dq", "-mavx512vl"]}
CLI = {"wrenc": ["cli/wrenc.cpp", "cli/gen_io.cpp"], "wrdec": ["cli/wrdec.cpp", "cli/gen_io.cpp"],
       # MSSG front-end (GrADS regular output, restart sets united / divided)
       "wrenc_mssg": ["cli/mssg_enc.cpp", "cli/mssg_io.cpp"], "wrdec_mssg": ["cli/mssg_dec.cpp", "cli/mssg_io.cpp"]}
# FluSI HDF5 front-end (libhdf5 is looked up at run time: HDF5_ROOT, the default path, /opt/conda)
FLUSI = {"wrenc_flusi": ["cli/flusi_enc.cpp", "cli/flusi_h5.cpp"], "wrdec_flusi": ["cli/flusi_dec.cpp", "cli/flusi_h5.cpp"]}
# -fvisibility=hidden: the library exports what include/waverange_amd.h declares and nothing else
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC", "-Wall",
          "-Wno-unused-result", "-march=x86-64-v3", "-fvisibility=hidden", "-fvisibility-inlines-hidden"]


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libwaverange_amd cannot be built (there is no CPU fallback)")


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def _deps():
    out = [os.path.join(HERE, "..", "include", "waverange_amd.h"), os.path.abspath(__file__)]
    for root, _, files in os.walk(CSRC):
        out += [os.path.join(root, f) for f in files if f.endswith((".hip", ".cpp", ".h", ".map"))]
    return out


def build(force=False, verbose=True):
    hipcc = _hipcc()
    deps = _deps()
    if force or _stale(LIB, deps):
        objs = []
        os.makedirs(os.path.join(HERE, "build"), exist_ok=True)
        for s in SOURCES:
            o = os.path.join(HERE, "build", s.replace("/", "_") + ".o")
            cmd = [hipcc] + COMMON + EXTRA.get(s, []) + os.environ.get("WR_CXXFLAGS", "").split() + ["-c", os.path.join(CSRC, s), "-o", o]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            objs.append(o)
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-o", LIB] + objs + ["-lhsa-runtime64", "-lpthread", "-Wl,-rpath,/opt/rocm/lib",
                                                                             "-Wl,--version-script=" + os.path.join(CSRC, "exports.map")]
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd)
        shutil.copyfile(LIB, ALIAS)
    for name, srcs in CLI.items():
        paths = [os.path.join(CSRC, s) for s in srcs]
        if not all(os.path.exists(p) for p in paths):
            continue
        exe = os.path.join(BIN, name)
        if force or _stale(exe, deps + [LIB]):
            os.makedirs(BIN, exist_ok=True)
            cmd = [hipcc, "-O2", "-std=c++17", "-ffp-contract=off", "-x", "c++"] + paths + [
                "-x", "none", "-o", exe, "-L" + HERE, "-lwaverange_amd", "-Wl,-rpath," + HERE,
                "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    # FluSI HDF5 front-end: libhdf5 is bound with dlopen at run time, so it always builds
    for name, srcs in FLUSI.items():
        paths = [os.path.join(CSRC, s) for s in srcs]
        if not all(os.path.exists(p) for p in paths):
            continue
        exe = os.path.join(BIN, name)
        if force or _stale(exe, deps + [LIB]):
            os.makedirs(BIN, exist_ok=True)
            cmd = ["g++", "-O2", "-std=c++17", "-ffp-contract=off"] + paths + [
                "-o", exe, "-L" + HERE, "-lwaverange_amd", "-ldl", "-lpthread", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib"]
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
