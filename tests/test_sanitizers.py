"""AddressSanitizer + UBSan runs of the host-side native code (CPU only: GPU sanitizers are not
available on the pool).  Covers the range coder (round trips, truncated and corrupted streams) and
the CLI file I/O + parameter parsing (our CLI sources on the reference codec)."""
import json
import os
import shutil
import subprocess
import tempfile

import pytest

import cli_cases
from util import GOLDEN, ROOT

CSRC = os.path.join(ROOT, "waverange_amd", "csrc")
SAN = ["-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined"]


def _have_san():
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "t.cpp")
        open(src, "w").write("int main(){return 0;}\n")
        return subprocess.run(["g++"] + SAN + [src, "-o", os.path.join(d, "t")], capture_output=True).returncode == 0


pytestmark = pytest.mark.skipif(shutil.which("g++") is None or not _have_san(), reason="g++ with ASan/UBSan not available")


def test_range_coder_under_sanitizers():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "rc_fuzz")
        vec_o = os.path.join(d, "vec.o")
        subprocess.check_call(["g++"] + SAN + ["-mavx512f", "-mavx512bw", "-mavx512dq", "-mavx512vl", "-c",
                                               os.path.join(CSRC, "wr_rangecoder_avx512.cpp"), "-o", vec_o])
        subprocess.check_call(["g++"] + SAN + ["-I" + CSRC, "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "rc_fuzz.cpp"),
                                               os.path.join(CSRC, "wr_rangecoder.cpp"), os.path.join(CSRC, "wr_compat.cpp"), vec_o, "-o", exe, "-lpthread"])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert "sanitizer run OK" in r.stdout


@pytest.mark.parametrize("san", ["thread", "address,undefined"])
def test_plane_handover_under_sanitizers(san):
    """The coder pool (streams changing workers forced) against a mock of the pipeline's plane streams that keeps the
    product's hand-over rules (csrc/wr_handover.h) and frees chunks the way the product does: ThreadSanitizer sees a worker
    that touches a stream after pool_wait, AddressSanitizer a window request that reaches behind a released chunk; then
    every violation of the rules one by one must be refused (tests/native/handover_tsan.cpp)."""
    flags = ["-std=c++17", "-O1", "-g", "-fsanitize=" + san]
    with tempfile.TemporaryDirectory() as d:
        probe = os.path.join(d, "t.cpp")
        open(probe, "w").write("int main(){return 0;}\n")
        if subprocess.run(["g++"] + flags + [probe, "-o", os.path.join(d, "t")], capture_output=True).returncode != 0:
            pytest.skip("g++ -fsanitize=%s not available" % san)
        exe = os.path.join(d, "handover")
        vec_o = os.path.join(d, "vec.o")
        subprocess.check_call(["g++"] + flags + ["-mavx512f", "-mavx512bw", "-mavx512dq", "-mavx512vl", "-c",
                                                 os.path.join(CSRC, "wr_rangecoder_avx512.cpp"), "-o", vec_o])
        subprocess.check_call(["g++"] + flags + ["-I" + CSRC, "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "native", "handover_tsan.cpp"),
                                                 os.path.join(CSRC, "wr_rangecoder.cpp"), vec_o, "-o", exe, "-lpthread"])
        r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
        assert "hand-over run OK" in r.stdout and "refused: 0" in r.stdout
        assert "ThreadSanitizer" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-4000:]


def test_cli_io_under_sanitizers():
    refso = os.path.join(ROOT, "oracle", "_ref", "libwaverange_ref.so")
    if not os.path.exists(refso):
        pytest.skip("oracle/_ref not built")
    with open(os.path.join(GOLDEN, "cli.json")) as fh:
        golden = json.load(fh)
    with tempfile.TemporaryDirectory() as b:
        exes = {}
        for name in ("wrenc", "wrdec"):
            exes[name] = os.path.join(b, name)
            subprocess.check_call(["g++"] + SAN + [os.path.join(CSRC, "cli", name + ".cpp"), os.path.join(CSRC, "cli", "gen_io.cpp"),
                                                   "-o", exes[name], "-L" + os.path.dirname(refso), "-lwaverange_ref",
                                                   "-Wl,-rpath," + os.path.dirname(refso)])
        env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")  # the reference codec itself leaks (wrappers.cpp:553)
        for case in ("inmeta_new_type0", "inmeta_old_type1_bigendian", "stdin_trivial"):
            with tempfile.TemporaryDirectory() as d:
                argv, stdin = cli_cases.write_inputs(case, d)
                r = subprocess.run([exes["wrenc"]] + argv, cwd=d, input=stdin, text=True, capture_output=True, env=env)
                assert r.returncode == 0, r.stderr[-3000:]
                if os.path.exists(os.path.join(d, "inmeta")):
                    os.remove(os.path.join(d, "inmeta"))
                assert open(os.path.join(d, "data.wrh")).read() == golden[case]["wrh"]
                r = subprocess.run([exes["wrdec"]] + cli_cases.dec_argv(case), cwd=d, capture_output=True, text=True, env=env)
                assert r.returncode == 0, r.stderr[-3000:]


def test_mssg_frontend_under_sanitizers():
    """wrenc_mssg / wrdec_mssg sources (control-file parsers, field I/O, header text) built with ASan + UBSan
    on the reference codec, over every MSSG case; outputs still equal the golden ones."""
    import hashlib
    import sys
    refso = os.path.join(ROOT, "oracle", "_ref", "libwaverange_ref.so")
    if not os.path.exists(refso):
        pytest.skip("oracle/_ref not built")
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_golden_mssg
    import mssg_cases
    with open(os.path.join(GOLDEN, "mssg.json")) as fh:
        golden = json.load(fh)
    with tempfile.TemporaryDirectory() as b:
        exes = {}
        for name in ("mssg_enc", "mssg_dec"):
            exes[name] = os.path.join(b, name)
            subprocess.check_call(["g++"] + SAN + [os.path.join(CSRC, "cli", name + ".cpp"), os.path.join(CSRC, "cli", "mssg_io.cpp"),
                                                   "-o", exes[name], "-L" + os.path.dirname(refso), "-lwaverange_ref",
                                                   "-Wl,-rpath," + os.path.dirname(refso)])
        os.environ["ASAN_OPTIONS"] = "detect_leaks=0"  # the reference codec itself leaks (wrappers.cpp:553)
        try:
            for case in sorted(mssg_cases.CASES):
                with tempfile.TemporaryDirectory() as d:
                    files = make_golden_mssg.run_case(case, exes["mssg_enc"], exes["mssg_dec"], d)
                for name, data in files.items():
                    assert hashlib.sha256(data).hexdigest() == golden[case][name]["sha256"], (case, name)
        finally:
            del os.environ["ASAN_OPTIONS"]
