#!/usr/bin/env python3
"""Sums rocprofv3 PMC counters per launch of k_fwd_fused / k_inv_fused (last transform of the run).
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES -d D1 --output-format csv -- python3 tools/prof_transform.py 1024 2 both
  rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d D2 ...
  rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS -d D3 ...
  python tools/pmc_sq.py D1 D2 D3 > profiles/rNN/..._sq_counters.txt"""
import csv
import glob
import os
import sys

for d in sys.argv[1:]:
    rows = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                k = r.get("Kernel_Name", "")
                kind = "fwd" if "k_fwd_fused" in k else "inv" if "k_inv_fused" in k else None
                if kind:
                    key = (kind, r["Counter_Name"], int(r["Dispatch_Id"]))
                    rows[key] = rows.get(key, 0.0) + float(r["Counter_Value"])
    for kind in ("fwd", "inv"):
        for c in sorted({k[1] for k in rows if k[0] == kind}):
            ds = sorted(k[2] for k in rows if k[0] == kind and k[1] == c)[-4:]
            print(os.path.basename(d.rstrip("/")), kind, c, ["%.4g" % rows[(kind, c, x)] for x in ds])
