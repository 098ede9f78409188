#!/usr/bin/env python3
"""BASELINE configs[4] at size: a synthetic FluSI backup set -- ux, uy, uz at n^3 (seeds 12345..12347), fp64 and fp32
datasets, attribute bckp = (time, dt1, dt0, n1, it, nx, ny, nz) -- through wrenc_flusi / wrdec_flusi at tol 1e-16 (8 bit
planes, near-lossless) on one GPU.  Wall clock and MB/s of either tool, the per-dataset phase lines of WR_CLI_TIMING=1
(dataset k+1's read, upload and kernels inside dataset k's codec call), and checks: every dataset's payload and coding
attributes equal what the library's own wr_encode_host gives for the same array (and, with --oracle, what the CPU oracle
gives for ux), the reconstruction is within tolerance.

    python tools/flusi_rate.py [--size 512] [--tol 1e-16] [--precisions 8,4] [--dir DIR] [--oracle] [--keep]
Reference loops being replaced: src/flusi/main_enc.cpp:469-505, main_dec.cpp:191,290."""
import argparse
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "waverange_amd", "bin")
H5ROOT = next((r for r in (os.environ.get("HDF5_ROOT"), "/opt/conda", "/usr") if r and os.path.exists(os.path.join(r, "include", "hdf5.h"))), None)


def build_h5tool(d):
    exe = os.path.join(d, "h5tool")
    subprocess.check_call(["gcc", "-O1", "-I" + os.path.join(H5ROOT, "include"), os.path.join(ROOT, "tests", "native", "h5tool.c"),
                           "-o", exe, "-L" + os.path.join(H5ROOT, "lib"), "-lhdf5", "-Wl,-rpath," + os.path.join(H5ROOT, "lib")])
    return exe


def timed(cmd, cwd, env):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
    dt = time.perf_counter() - t0
    if r.returncode:
        raise SystemExit("%s failed (%d): %s" % (cmd[0], r.returncode, r.stderr[-2000:]))
    return dt, [l for l in r.stderr.splitlines() if l.startswith("timing ")]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--tol", default="1e-16")
    ap.add_argument("--precisions", default="8,4")
    ap.add_argument("--dir", default=None)
    ap.add_argument("--oracle", action="store_true", help="also compare ux (fp64) with the CPU oracle (minutes at 512^3)")
    ap.add_argument("--keep", action="store_true", help="leave backup8.h5 / backup4.h5 in --dir (for a profiler run)")
    ap.add_argument("--make-only", action="store_true")
    args = ap.parse_args()
    if H5ROOT is None:
        raise SystemExit("no HDF5 C library found (HDF5_ROOT, /opt/conda, /usr)")
    n = args.size
    d = args.dir or tempfile.mkdtemp(prefix="wr_flusi_")
    os.makedirs(d, exist_ok=True)
    h5tool = build_h5tool(d)
    from waverange_amd import api
    api.set_verbosity(0)
    comps = {"ux": 12345, "uy": 12346, "uz": 12347}
    out = {"workload": "synthetic FluSI backup set: ux, uy, uz at %d^3, tol %s, one GPU" % (n, args.tol), "cpus": len(os.sched_getaffinity(0)), "runs": {}}
    mb = 3 * n ** 3 * 8 / 1e6   # what the codec sees: fp32 datasets are widened to fp64 on read (hdf5_interfaces.cpp:716-738)
    env = dict(os.environ, WR_QUIET="1", WR_CLI_TIMING="1")
    with api.Context(0) as ctx:
        buf = ctx.alloc(n ** 3 * 8)
        host = api.pinned_array((n, n, n))
        fields = {}
        for name, seed in comps.items():
            ctx.synth_field(buf, n, n, n, seed)
            ctx.sync()
            api._check(api.lib().wr_dev_download(ctx.h, host.ctypes.data, buf.ptr, host.nbytes))
            fields[name] = host.copy()
            fields[name].tofile(os.path.join(d, name + ".raw"))
        buf.free()
        for nbytes in [int(v) for v in args.precisions.split(",")]:
            h5 = "backup%d.h5" % nbytes
            argv = [h5tool, "make", os.path.join(d, h5), "backup", str(nbytes), str(n), str(n), str(n)]
            for name in comps:
                argv += [name, os.path.join(d, name + ".raw")]
            subprocess.check_call(argv)
            if args.make_only:
                continue
            te, enc_lines = timed([os.path.join(BIN, "wrenc_flusi"), h5, "comp.h5", "1", args.tol], d, env)
            td, dec_lines = timed([os.path.join(BIN, "wrdec_flusi"), "comp.h5", "rec.h5", "1", "2"], d, env)
            run = {"wrenc_flusi_s": round(te, 3), "wrdec_flusi_s": round(td, 3), "wrenc_MBps": round(mb / te, 1), "wrdec_MBps": round(mb / td, 1),
                   "roundtrip_MBps": round(mb / (te + td), 1), "comp_h5_bytes": os.path.getsize(os.path.join(d, "comp.h5")),
                   "phases_encode": enc_lines, "phases_decode": dec_lines}
            # what the tool stored against the library called directly on the same arrays, and the reconstruction
            dumpdir = os.path.join(d, "dump")
            os.makedirs(dumpdir, exist_ok=True)
            subprocess.check_call([h5tool, "dump", os.path.join(d, "comp.h5"), dumpdir], stdout=subprocess.DEVNULL)
            same, nlay = True, {}
            for name in comps:
                f = fields[name] if nbytes == 8 else fields[name].astype(np.float32).astype(np.float64)
                enc, _ = ctx.encode_host(f, float(args.tol))
                raw = np.fromfile(os.path.join(dumpdir, name + ".bin"), dtype=np.uint8)
                same = same and raw.size == enc["ntot_enc"] and np.array_equal(raw, enc["data"])
                nlay[name] = enc["nlay"]
                if args.oracle and name == "ux" and nbytes == 8:
                    from oracle.loader import Oracle
                    want = Oracle().encode(f, float(args.tol))
                    run["ux_payload_equals_oracle"] = bool(want["ntot_enc"] == raw.size and np.array_equal(want["data"], raw))
            run["payload_equals_library_call"] = bool(same)
            run["nlay"] = nlay
            shutil.rmtree(dumpdir)
            os.makedirs(dumpdir, exist_ok=True)
            subprocess.check_call([h5tool, "dump", os.path.join(d, "rec.h5"), dumpdir], stdout=subprocess.DEVNULL)
            worst = 0.0
            for name in comps:
                f = fields[name] if nbytes == 8 else fields[name].astype(np.float32).astype(np.float64)
                rec = np.fromfile(os.path.join(dumpdir, name + ".bin"), dtype=np.float64).reshape(f.shape)
                worst = max(worst, float(np.abs(rec - f).max() / np.abs(f).max()))
            run["linf_rel_worst"] = worst
            shutil.rmtree(dumpdir)
            out["runs"]["fp%d" % (nbytes * 8)] = run
            for fn in ("comp.h5", "rec.h5"):
                os.remove(os.path.join(d, fn))
    if not args.keep and not args.make_only:
        shutil.rmtree(d, ignore_errors=True)
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
