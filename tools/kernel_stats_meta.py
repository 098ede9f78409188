#!/usr/bin/env python3
"""Copies the rocprofv3 --kernel-trace --memory-copy-trace --stats summaries of a bench.py run into profiles/rNN/ as
final_kernel_stats.csv / final_memory_copy_stats.csv and writes final_kernel_stats.meta.json beside them: the hashes of the
kernel sources the run was taken on (bench.py marks the figures stale once those change, like `traffic_stale`), the commit,
the command and whether the clock warm-up hook was set.

  cd /tmp && export TMPDIR=/tmp
  rocprofv3 --kernel-trace --memory-copy-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof --output-format csv -- \
      python3 $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline --secondary-steps 0
  python3 tools/kernel_stats_meta.py gpurun_out/prof profiles/r05 "bench.py --steps 3 --warmup 1 ..."
"""
import glob
import hashlib
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    src, dst, cmd = sys.argv[1], sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else ""
    os.makedirs(dst, exist_ok=True)
    found = {}
    for kind in ("kernel_stats", "memory_copy_stats"):
        fs = sorted(glob.glob(os.path.join(src, "**", "*_%s.csv" % kind), recursive=True), key=os.path.getsize)
        if fs:
            shutil.copyfile(fs[-1], os.path.join(dst, "final_%s.csv" % kind))
            found[kind] = os.path.relpath(fs[-1], src)
    if "kernel_stats" not in found:
        raise SystemExit("no *_kernel_stats.csv under " + src)
    meta = {"command": cmd, "files": found, "kernel_sources_sha256": {}, "clock_warmup_ms": float(os.environ.get("WR_CLOCK_WARMUP_MS", "0") or 0)}
    for name in ("wr_fused.hip", "wr_kernels.hip"):
        with open(os.path.join(ROOT, "waverange_amd", "csrc", name), "rb") as fh:
            meta["kernel_sources_sha256"][name] = hashlib.sha256(fh.read()).hexdigest()
    try:
        meta["head"] = subprocess.run(["git", "-C", ROOT, "rev-parse", "--short", "HEAD"], capture_output=True, text=True).stdout.strip() or None
    except OSError:
        meta["head"] = None
    with open(os.path.join(dst, "final_kernel_stats.meta.json"), "w") as fh:
        json.dump(meta, fh, indent=1, sort_keys=True)
        fh.write("\n")
    print(json.dumps(meta))


if __name__ == "__main__":
    main()
