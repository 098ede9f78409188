"""The command-line tools at a size where rates mean something, in front of the driver: the generic
tools on BASELINE configs[3]'s file shape (NF = 8 fields in one raw file) against the REFERENCE's own wrenc / wrdec
(compiled by oracle/Makefile, present on the GPU box as binaries) -- identical .wrh / .wrb / decoded file -- and the FluSI
tools on a backup set at tol 1e-16 (configs[4]) against the library called directly and the CPU oracle.  The rates land in
the test log as warnings (they survive -q); 256^3 keeps both tests under a minute.
Loops replaced: src/generic/gen_enc.cpp:538-605, src/flusi/main_enc.cpp:469-505."""
import json
import os
import subprocess
import sys
import warnings

import pytest

from util import ROOT

pytestmark = pytest.mark.gpu
H5ROOT = next((r for r in (os.environ.get("HDF5_ROOT"), "/opt/conda", "/usr") if r and os.path.exists(os.path.join(r, "include", "hdf5.h"))), None)


def _run(tool, args, tmp_path):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", tool)] + args, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])


def test_generic_tools_nf8_256_identical_to_the_reference_tools(tmp_path):
    out = _run("cli_rate.py", ["--nf", "8", "--size", "256", "--tol", "1e-5", "--dir", str(tmp_path / "cli"), "--repeat", "2"], tmp_path)
    best = out["ours"]["best"]
    msg = "wrenc + wrdec on NF = 8 x 256^3 fp64, tol 1e-5: %.2f s + %.2f s = %.0f MB/s round trip" % (best["wrenc_s"], best["wrdec_s"], best["roundtrip_MBps"])
    if "reference_cli" in out:
        assert out["identical_to_reference"] == {"wrh": True, "wrb": True, "decoded_file": True}, out["identical_to_reference"]
        msg += "; the reference's tools on the same host %.1f s + %.1f s (%.1f x), .wrh / .wrb / decoded file identical" % (
            out["reference_cli"]["wrenc_s"], out["reference_cli"]["wrdec_s"], out["speedup_vs_reference_cli"])
    else:
        msg += "; oracle/_ref/wrenc_ref not present: no comparison with the reference's tools"
    warnings.warn(UserWarning("tools at size: " + msg))
    assert best["roundtrip_MBps"] > 0


def test_flusi_tools_backup_set_256_tol_1e_16(tmp_path):
    if H5ROOT is None:
        pytest.skip("no HDF5 C library for the FluSI tools")
    out = _run("flusi_rate.py", ["--size", "256", "--tol", "1e-16", "--dir", str(tmp_path / "flusi"), "--oracle"], tmp_path)
    for prec, run in out["runs"].items():
        assert run["payload_equals_library_call"], prec
        assert run["linf_rel_worst"] < 1e-13, (prec, run["linf_rel_worst"])
        assert all(v == 8 for v in run["nlay"].values()), run["nlay"]
        if prec == "fp64":
            assert run["ux_payload_equals_oracle"], "ux payload differs from the oracle's"
        warnings.warn(UserWarning("tools at size: FluSI backup set (ux, uy, uz at 256^3, %s, tol 1e-16, 8 planes): wrenc_flusi %.2f s, wrdec_flusi %.2f s, %.0f MB/s "
                                  "round trip" % (prec, run["wrenc_flusi_s"], run["wrdec_flusi_s"], run["roundtrip_MBps"])))
