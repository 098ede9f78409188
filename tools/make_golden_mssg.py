#!/usr/bin/env python3
"""Golden outputs of the MSSG front-end: runs the REFERENCE's wrmssgenc / wrmssgdec, compiled from
/root/reference/src/mssg by oracle/Makefile (oracle/_ref/wrmssgenc_ref, wrmssgdec_ref, linked with the
reference codec), on the synthetic data sets of tests/mssg_cases.py and stores the header texts and
the hashes of every other output in tests/golden/mssg.json."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import mssg_cases  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref")


def run_case(case, enc, dec, workdir):
    """Encode + decode `case` in workdir with the given binaries; returns {file: bytes}."""
    c = mssg_cases.CASES[case]
    inputs = mssg_cases.write_inputs(case, workdir)
    for proc in c.get("procs", [0]):
        argv, stdin, inmeta = mssg_cases.enc_invocation(case, proc)
        meta = os.path.join(workdir, "inmeta")
        if inmeta is not None:
            with open(meta, "w") as fh:
                fh.write(inmeta)
        subprocess.run([enc] + argv, cwd=workdir, input=stdin, text=True, check=True, stdout=subprocess.DEVNULL)
        if os.path.exists(meta):
            os.remove(meta)
    for proc in c.get("procs", [0]):
        argv, stdin = mssg_cases.dec_invocation(case, proc)
        subprocess.run([dec] + argv, cwd=workdir, input=stdin, text=True, check=True, stdout=subprocess.DEVNULL)
    out = {}
    for group in mssg_cases.output_files(case):
        for f in group:
            with open(os.path.join(workdir, f), "rb") as fh:
                out[f] = fh.read()
    out["__inputs__"] = b"".join(open(os.path.join(workdir, f), "rb").read() for f in inputs)
    return out


def digest(files):
    rec = {}
    for name, data in sorted(files.items()):
        entry = {"size": len(data), "sha256": hashlib.sha256(data).hexdigest()}
        if "_h" in name and name.endswith(".enc"):
            entry["text"] = data.decode()
        rec[name] = entry
    return rec


def main():
    enc, dec = os.path.join(REF, "wrmssgenc_ref"), os.path.join(REF, "wrmssgdec_ref")
    golden = {}
    for case in sorted(mssg_cases.CASES):
        with tempfile.TemporaryDirectory() as d:
            golden[case] = digest(run_case(case, enc, dec, d))
        print(case, {k: v["size"] for k, v in golden[case].items()})
    with open(os.path.join(ROOT, "tests", "golden", "mssg.json"), "w") as fh:
        json.dump(golden, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
