#!/usr/bin/env python3
"""HBM traffic per launch of every kernel whose name contains one of the given substrings, from two rocprofv3 PMC passes:

  rocprofv3 --pmc FETCH_SIZE -d DIR_F --output-format csv -- python3 tools/prof_codec.py 1024 1e-7 1
  rocprofv3 --pmc WRITE_SIZE -d DIR_W --output-format csv -- python3 tools/prof_codec.py 1024 1e-7 1
  python tools/pmc_kernels.py DIR_F DIR_W k_quant_blk k_dequant_lds k_hist > profiles/rNN/..._traffic.json

FETCH_SIZE doubled (gfx950 tallies the 128-B requests of 16-B/lane streaming reads at 64 B: MI355X_MICROARCH.md, HBM
section), WRITE_SIZE as reported, both in KiB."""
import csv
import glob
import json
import os
import sys


def per_dispatch(d, counter):
    agg = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                if r.get("Counter_Name") != counter:
                    continue
                a = agg.setdefault(int(r["Dispatch_Id"]), {"kernel": r.get("Kernel_Name", ""), "KiB": 0.0})
                a["KiB"] += float(r["Counter_Value"])
    return agg


def main():
    df, dw, names = sys.argv[1], sys.argv[2], sys.argv[3:]
    fe, wr = per_dispatch(df, "FETCH_SIZE"), per_dispatch(dw, "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes); FETCH_SIZE doubled, WRITE_SIZE as reported; KiB = 1024 B", "kernels": {}}
    for nm in names:
        f = [v["KiB"] * 2048.0 for k, v in sorted(fe.items()) if nm in v["kernel"]]
        w = [v["KiB"] * 1024.0 for k, v in sorted(wr.items()) if nm in v["kernel"]]
        out["kernels"][nm] = {"launches": len(f), "fetch_bytes": f, "write_bytes": w,
                              "mean_total_bytes": (sum(f) / max(1, len(f))) + (sum(w) / max(1, len(w)))}
    json.dump(out, sys.stdout, indent=1)


if __name__ == "__main__":
    main()
