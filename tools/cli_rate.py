#!/usr/bin/env python3
"""Wall clock of the generic command-line tools on BASELINE configs[3]'s file (NF independent n^3 fp64 fields in one raw
file, seeds 12345 .. 12345+NF-1, one tolerance) on ONE GPU: our wrenc / wrdec (waverange_amd/bin, the library's coder pool
and as many fields in flight as wr_autotune_batch allows) next to the reference's own wrenc / wrdec (oracle/_ref, compiled
from /root/reference by oracle/Makefile) on the same host, and a byte comparison of the three files either pair writes.

    python tools/cli_rate.py [--nf 8] [--size 512] [--tol 1e-5] [--dir /tmp/wr_cli] [--no-ref] [--pipeline K]

Prints one JSON object.  MB/s = field bytes (NF x 8 n^3, 10^6 B per MB) / wall seconds of the process, file I/O included
(input and outputs in --dir: whatever file system that is, usually the page cache)."""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
BIN = os.path.join(ROOT, "waverange_amd", "bin")
REF = os.path.join(ROOT, "oracle", "_ref")


def timed(cmd, cwd, env=None):
    t0 = time.perf_counter()
    r = subprocess.run(cmd, cwd=cwd, stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True, env=env)
    dt = time.perf_counter() - t0
    if r.returncode:
        raise SystemExit("%s failed (%d): %s" % (cmd[0], r.returncode, r.stderr[-2000:]))
    return dt


def same(a, b):
    return subprocess.run(["cmp", "-s", a, b]).returncode == 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--nf", type=int, default=8)
    ap.add_argument("--size", type=int, default=512)
    ap.add_argument("--tol", default="1e-5")
    ap.add_argument("--dir", default=os.path.join(os.environ.get("TMPDIR", "/tmp"), "wr_cli"))
    ap.add_argument("--no-ref", action="store_true")
    ap.add_argument("--pipeline", default=None, help="WR_CLI_PIPELINE for our tools (default: the library decides)")
    ap.add_argument("--repeat", type=int, default=2, help="runs of our tools (the first one pays the page-cache and driver warm-up)")
    args = ap.parse_args()
    n, nf = args.size, args.nf
    os.makedirs(args.dir, exist_ok=True)
    d = args.dir
    from waverange_amd import api
    import numpy as np
    api.set_verbosity(0)
    t0 = time.perf_counter()
    with api.Context(0) as ctx, open(os.path.join(d, "data.bin"), "wb") as fh:
        buf = ctx.alloc(n ** 3 * 8)
        host = api.pinned_array((n, n, n))
        for k in range(nf):  # the fields of bench.py's ranks: seed 12345 + k
            ctx.synth_field(buf, n, n, n, 12345 + k)
            ctx.sync()
            api._check(api.lib().wr_dev_download(ctx.h, host.ctypes.data, buf.ptr, host.nbytes))
            host.tofile(fh)
        buf.free()
        del host
    t_make = time.perf_counter() - t0
    mb = nf * n ** 3 * 8 / 1e6
    enc_args = ["data.bin", "X.wrb", "X.wrh", "2", "0", str(nf), "2", str(n), str(n), str(n), args.tol]
    out = {"workload": "NF = %d independent %d^3 fp64 fields in one raw file (TYPE 2), tol %s, one GPU" % (nf, n, args.tol),
           "field_MB": mb, "make_input_s": round(t_make, 2), "dir": d, "cpus": len(os.sched_getaffinity(0))}
    env = dict(os.environ, WR_QUIET="1")
    if args.pipeline is not None:
        env["WR_CLI_PIPELINE"] = args.pipeline
    runs = []
    for r in range(args.repeat):
        te = timed([os.path.join(BIN, "wrenc")] + [a.replace("X", "ours") for a in enc_args], d, env)
        td = timed([os.path.join(BIN, "wrdec"), "ours.wrb", "ours.wrh", "ours_rec.bin", "2", "0"], d, env)
        runs.append({"wrenc_s": round(te, 3), "wrdec_s": round(td, 3), "wrenc_MBps": round(mb / te, 1), "wrdec_MBps": round(mb / td, 1),
                     "roundtrip_MBps": round(mb / (te + td), 1)})
    out["ours"] = {"runs": runs, "best": max(runs, key=lambda x: x["roundtrip_MBps"]), "pipeline": args.pipeline or "library (wr_autotune_batch)",
                   "wrb_bytes": os.path.getsize(os.path.join(d, "ours.wrb"))}
    if not args.no_ref and os.path.exists(os.path.join(REF, "wrenc_ref")):
        te = timed([os.path.join(REF, "wrenc_ref")] + [a.replace("X", "ref") for a in enc_args], d)
        td = timed([os.path.join(REF, "wrdec_ref"), "ref.wrb", "ref.wrh", "ref_rec.bin", "2", "0"], d)
        out["reference_cli"] = {"wrenc_s": round(te, 2), "wrdec_s": round(td, 2), "roundtrip_MBps": round(mb / (te + td), 2),
                                "build": "oracle/_ref (gcc -O2, -ffp-contract=off), one thread, as the reference runs"}
        # the header names its .wrb file on line 3: everything else must be identical
        ho = open(os.path.join(d, "ours.wrh")).read().replace("ours.wrb", "X.wrb")
        hr = open(os.path.join(d, "ref.wrh")).read().replace("ref.wrb", "X.wrb")
        out["identical_to_reference"] = {"wrh": ho == hr, "wrb": same(os.path.join(d, "ours.wrb"), os.path.join(d, "ref.wrb")),
                                         "decoded_file": same(os.path.join(d, "ours_rec.bin"), os.path.join(d, "ref_rec.bin"))}
        out["speedup_vs_reference_cli"] = round((out["reference_cli"]["wrenc_s"] + out["reference_cli"]["wrdec_s"]) /
                                                (out["ours"]["best"]["wrenc_s"] + out["ours"]["best"]["wrdec_s"]), 1)
    for f in os.listdir(d):
        os.remove(os.path.join(d, f))
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
