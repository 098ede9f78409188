#!/usr/bin/env python3
"""Turn two rocprofv3 PMC passes over tools/prof_transform.py into profiles/rNN/traffic_<n>.json.

  rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch --output-format csv -- python3 tools/prof_transform.py 1024 2 both
  rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write --output-format csv -- python3 tools/prof_transform.py 1024 2 both
  python tools/pmc_traffic.py 1024 gpurun_out/pmc_fetch gpurun_out/pmc_write > profiles/r01/traffic_1024.json

Per transform = the four level launches of k_fwd_fused / k_inv_fused of the LAST repetition.  FETCH_SIZE
is doubled (gfx950 tallies the 128-B requests of 16-B/lane streaming reads at 64 B; MI355X_MICROARCH.md,
HBM section); both counters are in KiB."""
import csv
import glob
import json
import os
import sys


def rows(d, counter):
    out = []
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                k = r.get("Kernel_Name", "")
                kind = "fwd" if "k_fwd_fused" in k else "inv" if "k_inv_fused" in k else None
                if kind:
                    out.append({"dispatch": int(r["Dispatch_Id"]), "kernel": kind, "KiB": float(r["Counter_Value"])})
    # one row per dispatch (rocprofv3 may emit one row per XCD/instance: sum them)
    agg = {}
    for r in out:
        a = agg.setdefault(r["dispatch"], {"dispatch": r["dispatch"], "kernel": r["kernel"], "KiB": 0.0})
        a["KiB"] += r["KiB"]
    return [agg[k] for k in sorted(agg)]


def last_transform(rs, kind, launches=4):
    sel = [r for r in rs if r["kernel"] == kind][-launches:]
    return sum(r["KiB"] for r in sel) * 1024.0


def main():
    n, dfetch, dwrite = int(sys.argv[1]), sys.argv[2], sys.argv[3]
    fe, wr = rows(dfetch, "FETCH_SIZE"), rows(dwrite, "WRITE_SIZE")
    out = {"size": n,
           "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 tools/prof_transform.py %d 2 both; last repetition" % n,
           "correction": "FETCH_SIZE doubled (gfx950: 128-B requests tallied at 64 B for 16-B/lane streaming reads, MI355X_MICROARCH.md HBM section); WRITE_SIZE as reported; KiB = 1024 B"}
    import hashlib
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out["kernel_sources_sha256"] = {}
    for name in ("wr_fused.hip",):  # the kernels the figure belongs to: k_fwd_fused, k_inv_fused  # bench.py marks the figure stale once these change
        with open(os.path.join(here, "waverange_amd", "csrc", name), "rb") as fh:
            out["kernel_sources_sha256"][name] = hashlib.sha256(fh.read()).hexdigest()
    for kind in ("fwd", "inv"):
        f, w = 2.0 * last_transform(fe, kind), last_transform(wr, kind)
        out[kind] = {"fetch_bytes": f, "write_bytes": w, "total_bytes": f + w}
    out["raw"] = {"FETCH_SIZE": fe, "WRITE_SIZE": wr}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
