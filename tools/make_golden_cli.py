#!/usr/bin/env python3
"""Golden vectors G5 for the generic CLI: run the compiled REFERENCE wrenc/wrdec
(oracle/_ref/wrenc_ref, wrdec_ref; strict-IEEE build) on the cases of tests/cli_cases.py and
store the .wrh text, the SHA-256/size of the .wrb and of the decoded file in
tests/golden/cli.json.  Build container only."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cli_cases  # noqa: E402
from oracle import loader  # noqa: E402


def sha_file(p):
    return hashlib.sha256(open(p, "rb").read()).hexdigest()


def main():
    out = {}
    for case in cli_cases.CASES:
        with tempfile.TemporaryDirectory() as d:
            argv, stdin = cli_cases.write_inputs(case, d)
            subprocess.run([loader.REF_WRENC] + argv, cwd=d, input=stdin, text=True, check=True,
                           stdout=subprocess.DEVNULL)
            if os.path.exists(os.path.join(d, "inmeta")):
                os.remove(os.path.join(d, "inmeta"))
            subprocess.run([loader.REF_WRDEC] + cli_cases.dec_argv(case), cwd=d, check=True, stdout=subprocess.DEVNULL)
            out[case] = dict(
                input_sha256=sha_file(os.path.join(d, "data.bin")),
                wrh=open(os.path.join(d, "data.wrh")).read(),
                wrb_size=os.path.getsize(os.path.join(d, "data.wrb")),
                wrb_sha256=sha_file(os.path.join(d, "data.wrb")),
                rec_size=os.path.getsize(os.path.join(d, "datarec.bin")),
                rec_sha256=sha_file(os.path.join(d, "datarec.bin")))
            print(case, out[case]["wrb_size"], out[case]["rec_size"])
    with open(os.path.join(ROOT, "tests", "golden", "cli.json"), "w") as fh:
        json.dump(out, fh, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
