"""Generic CLI (wrenc / wrdec) parity: .wrh text, .wrb bytes and the decoded file against the
golden vectors produced by the compiled reference CLI (tools/make_golden_cli.py).

  * CPU: OUR CLI sources linked against the REFERENCE codec (oracle/_ref/*_ours_refcodec) --
    proves the file formats, the three parameter modes and the C-ABI call compatibility.
  * GPU: OUR CLI on OUR library (waverange_amd/bin), and the REFERENCE's compiled CLI running
    on OUR library under the reference's library name (oracle/_ref/*_ref_dyn) -- the drop-in.
"""
import hashlib
import json
import os
import subprocess
import tempfile

import pytest

import cli_cases
from util import GOLDEN, ROOT

REFDIR = os.path.join(ROOT, "oracle", "_ref")
BINDIR = os.path.join(ROOT, "waverange_amd", "bin")


@pytest.fixture(scope="module")
def golden_cli():
    with open(os.path.join(GOLDEN, "cli.json")) as fh:
        return json.load(fh)


def sha_file(p):
    return hashlib.sha256(open(p, "rb").read()).hexdigest()


def run_case(case, wrenc, wrdec, g, **extra_env):
    env = dict(os.environ, WR_QUIET="1", **extra_env)
    with tempfile.TemporaryDirectory() as d:
        argv, stdin = cli_cases.write_inputs(case, d)
        assert sha_file(os.path.join(d, "data.bin")) == g["input_sha256"]
        subprocess.run([wrenc] + argv, cwd=d, input=stdin, text=True, check=True, stdout=subprocess.DEVNULL, env=env)
        if os.path.exists(os.path.join(d, "inmeta")):
            os.remove(os.path.join(d, "inmeta"))
        assert open(os.path.join(d, "data.wrh")).read() == g["wrh"], "header text differs"
        assert os.path.getsize(os.path.join(d, "data.wrb")) == g["wrb_size"]
        assert sha_file(os.path.join(d, "data.wrb")) == g["wrb_sha256"], ".wrb bytes differ"
        subprocess.run([wrdec] + cli_cases.dec_argv(case), cwd=d, check=True, stdout=subprocess.DEVNULL, env=env)
        assert os.path.getsize(os.path.join(d, "datarec.bin")) == g["rec_size"]
        assert sha_file(os.path.join(d, "datarec.bin")) == g["rec_sha256"], "decoded file differs"


@pytest.mark.parametrize("case", sorted(cli_cases.CASES))
def test_our_cli_on_reference_codec(case, golden_cli):
    enc, dec = os.path.join(REFDIR, "wrenc_ours_refcodec"), os.path.join(REFDIR, "wrdec_ours_refcodec")
    if not (os.path.exists(enc) and os.path.exists(dec)):
        pytest.skip("oracle/_ref cross binaries not built")
    run_case(case, enc, dec, golden_cli[case])


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(cli_cases.CASES))
def test_our_cli_on_gpu(case, golden_cli):
    run_case(case, os.path.join(BINDIR, "wrenc"), os.path.join(BINDIR, "wrdec"), golden_cli[case])


@pytest.mark.gpu
@pytest.mark.parametrize("threads", ["1", "2"])
def test_our_cli_on_gpu_with_few_coder_threads(threads, golden_cli):
    """WR_THREADS < number of planes: the planes are coded in groups with interleaved symbol loops; the
    files must not change."""
    run_case("config1_64cube", os.path.join(BINDIR, "wrenc"), os.path.join(BINDIR, "wrdec"), golden_cli["config1_64cube"],
             WR_THREADS=threads)


@pytest.mark.gpu
@pytest.mark.parametrize("case", ["config1_64cube", "inmeta_new_type0"])
def test_reference_cli_on_our_library(case, golden_cli):
    enc, dec = os.path.join(REFDIR, "wrenc_ref_dyn"), os.path.join(REFDIR, "wrdec_ref_dyn")
    if not (os.path.exists(enc) and os.path.exists(dec)):
        pytest.skip("oracle/_ref/*_ref_dyn not built")
    run_case(case, enc, dec, golden_cli[case])


@pytest.mark.gpu
def test_sharded_driver_with_gpu_codec(golden_cli):
    """waverange_amd.sharded with the product codec (Context.encode), single process."""
    import numpy as np
    from waverange_amd import api, sharded
    api.set_verbosity(0)
    case = "inmeta_new_type0"
    c = cli_cases.CASES[case]
    specs = []
    for fd in c["fields"]:
        nbytes, nx, ny, nz, nh, idinv = fd["spec"]
        specs.append(dict(nbytes=nbytes, nx=nx, ny=ny, nz=nz, nh=nh, idinv=idinv, icomp=fd["icomp"], tol_base=float(fd["tol"])))
    with api.Context(0) as ctx:
        def codec(fld, tol):
            buf = ctx.to_device(fld)
            enc, _ = ctx.encode(buf, fld.shape, tol)
            enc["data"] = enc["data"].copy()
            buf.free()
            return enc
        with tempfile.TemporaryDirectory() as d:
            cli_cases.write_inputs(case, d)
            sharded.wrenc_sharded(os.path.join(d, "data.bin"), os.path.join(d, "data.wrb"), os.path.join(d, "data.wrh"),
                                  specs, c["file_type"], bool(c["flip"]), codec)
            assert open(os.path.join(d, "data.wrh")).read() == golden_cli[case]["wrh"]
            assert sha_file(os.path.join(d, "data.wrb")) == golden_cli[case]["wrb_sha256"]


@pytest.mark.gpu
def test_c_example_links_against_library_under_reference_name():
    """examples/example_roundtrip.c: a plain C client, linked with -lwaverange (the reference's
    library name), runs on the GPU and meets the tolerance."""
    import shutil
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    lib = os.path.join(ROOT, "waverange_amd")
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "example_roundtrip")
        subprocess.check_call(["gcc", "-O2", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "example_roundtrip.c"),
                               "-o", exe, "-L" + lib, "-lwaverange", "-Wl,-rpath," + lib, "-Wl,-rpath,/opt/rocm/lib", "-lm"])
        r = subprocess.run([exe, "1e-6"], capture_output=True, text=True, env=dict(os.environ, WR_QUIET="1"))
        assert r.returncode == 0, r.stdout + r.stderr
        assert "Linf_rel" in r.stdout
