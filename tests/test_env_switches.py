"""Every environment switch the library still reads (20 of them) is set by at least one test.  The
ones no other test file covers are here; each runs in a child process, because the library reads its environment once."""
import os
import subprocess
import sys

import numpy as np
import pytest

from util import GOLDEN, ROOT

PRE = "import sys; sys.path.insert(0, %r)\nimport numpy as np\nfrom waverange_amd import api, synth\nfrom oracle.loader import Oracle\n" % ROOT


def child(tmp_path, body, timeout=600, **env):
    script = tmp_path / "child.py"
    script.write_text(PRE + body)
    return subprocess.run([sys.executable, str(script)], env=dict(os.environ, **env), capture_output=True, text=True, timeout=timeout)


POOL_PLANES = r"""
o = Oracle()
rs = np.random.RandomState(5)
planes = []
for k in range(20):   # more planes than a scalar loop holds, all kinds, lengths around block boundaries
    n = 60000 * (4 + k % 3) + (0, 1, 59999, 4321)[k % 4]
    planes.append(rs.randint(0, 256, n).astype(np.uint8) if k % 3 == 0 else
                  rs.choice(np.array([127, 128], np.uint8), size=n, p=[0.8, 0.2]) if k % 3 == 1 else
                  np.where(rs.random_sample(n) < 0.999, 255, rs.randint(0, 256, n)).astype(np.uint8))
want = [o.range_encode(p) for p in planes]
api.set_coder_pool(3, 4)
enc = api.range_encode_pool(planes)
assert all(np.array_equal(a, b) for a, b in zip(enc, want)), "pool encode"
dec, got = api.range_decode_pool(want, [p.size for p in planes])
assert all(g == p.size and np.array_equal(d, p) for d, g, p in zip(dec, got, planes)), "pool decode"
st = api.pool_loop_stats()
api.set_coder_pool(0)
print("blocks", {k: int(v[1]) for k, v in st.items()})
"""


def test_pool_scalar_encoder_sessions_same_bytes(tmp_path):
    """WR_VEC_ENCODE=0: the pool's encoder sessions are scalar loops of three planes (what a CPU without AVX-512 runs) -- the
    oracle's bytes, and the loop statistics show which loop did the work."""
    r = child(tmp_path, POOL_PLANES + "assert st['vector_encoder'][1] == 0 and st['scalar_encoder'][1] > 0, st\nprint('ok')\n", WR_VEC_ENCODE="0")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]


def test_coder_without_avx512_same_bytes(tmp_path):
    """WR_NO_AVX512=1: no 16-lane loop is entered anywhere -- the vector hooks say "unsupported", the pool codes every plane on
    its scalar loops -- and the bytes and symbols are the oracle's."""
    body = POOL_PLANES + ("assert st['vector_encoder'][1] == 0 and st['vector_decoder'][1] == 0, st\n"
                          "assert api.lib().wr_range_decode_vec(0, None, None, None, None, None) != 0\n"
                          "try:\n    api.range_encode_vec(planes[:2]); raise SystemExit('vector encoder ran')\nexcept api.WaveRangeError:\n    pass\nprint('ok')\n")
    r = child(tmp_path, body, WR_NO_AVX512="1")
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]


DROPIN = r"""
api.set_verbosity(0)
f = synth.field(40, 36, 28, seed=77)
want = Oracle().encode(f, 1e-6)
enc = api.encoding_wrap(f, 1e-6)
assert enc["len_enc_vec"] == want["len_enc_vec"] and np.array_equal(enc["data"], want["data"]), "coded bytes"
print("residual_written", not np.array_equal(enc["residual"].view(np.uint64), f.view(np.uint64)))
rec = api.decoding_wrap(enc, f.shape)
assert np.array_equal(rec.view(np.uint64), Oracle().decode(want, f.shape).view(np.uint64))
print("ok")
"""


@pytest.mark.gpu
def test_env_switches_of_the_drop_in_symbols(tmp_path):
    """WR_DEVICE picks the GPU of the contexts the drop-in symbols create (0 works; a GPU that is not there fails loudly, there
    is no fallback); WR_WRITEBACK_RESIDUAL=0 leaves fld_1d as it was (default: the residual in wavelet space, as the
    reference leaves it, wrappers.cpp:397-398)."""
    r = child(tmp_path, DROPIN, WR_DEVICE="0")
    assert r.returncode == 0 and "residual_written True" in r.stdout and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]
    r = child(tmp_path, DROPIN, WR_DEVICE="0", WR_WRITEBACK_RESIDUAL="0")
    assert r.returncode == 0 and "residual_written False" in r.stdout and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]
    r = child(tmp_path, DROPIN, WR_DEVICE="63")
    assert r.returncode != 0 and "ok" not in r.stdout.split() and "device" in (r.stdout + r.stderr).lower(), r.stdout[-2000:] + r.stderr[-3000:]


RESERVE = r"""
import time
api.set_verbosity(0)
o = Oracle()
f = synth.field(96, 80, 64, seed=3)
want = o.encode(f, 1e-16)   # eight planes: every plane after the first is allocated while the caller holds planes itself
t0 = time.time()
with api.Context(0) as c:
    enc, _ = c.encode_host(f, 1e-16)
    assert enc["nlay"] == 8 and np.array_equal(enc["data"], want["data"]), "coded bytes"
    out = np.empty_like(f); enc["data"] = enc["data"].copy()
    c.decode_host(out, enc)
    assert np.array_equal(out.view(np.uint64), o.decode(want, f.shape).view(np.uint64)), "reconstruction"
print("seconds %.1f waited_ms %d" % (time.time() - t0, api.stat(api.STAT_PLANE_WAIT_MS)))
print("ok")
"""


@pytest.mark.gpu
@pytest.mark.parametrize("chunk_mb", [None, "1"])
def test_plane_reserve_does_not_stall_a_lone_caller(tmp_path, chunk_mb):
    """WR_PLANE_RESERVE_MB larger than the whole device: every plane allocation is "below the reserve".  With no plane of
    another call outstanding nothing can come back, so the allocation is tried as it is (until round 4 the call waited five
    minutes for nothing and failed) -- planes as one array and in 1 MiB chunks."""
    env = dict(WR_PLANE_RESERVE_MB="400000")
    if chunk_mb:
        env["WR_PLANE_CHUNK_MB"] = chunk_mb
    r = child(tmp_path, RESERVE, timeout=120, **env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), r.stdout[-2000:] + r.stderr[-3000:]
    assert "waited_ms 0" in r.stdout, r.stdout


WARMUP = r"""
api.set_verbosity(0)
f = synth.field(64, 64, 64, seed=9)
want = Oracle().encode(f, 1e-5)
with api.Context(0) as c:
    for _ in range(2):
        enc, _ = c.encode_host(f, 1e-5)
        assert np.array_equal(enc["data"], want["data"])
        out = np.empty_like(f); enc["data"] = enc["data"].copy()
        import time; time.sleep(0.05)
        c.decode_host(out, enc)
print("warmup_ms", api.stat(api.STAT_CLOCK_WARMUP_MS))
"""


@pytest.mark.gpu
def test_clock_warmup_hook_is_opt_in(tmp_path):
    """The clock warm-up in front of a kernel stage (a burner kernel, wr_pipeline.cpp: clock_warmup) is a measurement hook:
    nothing of it runs unless WR_CLOCK_WARMUP_MS is set (round 4 shipped it on; it raised the reported kernel rate and left
    the throughput where it was), and with it set the bytes are the same."""
    r = child(tmp_path, WARMUP)
    assert r.returncode == 0 and "warmup_ms 0" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]
    r = child(tmp_path, WARMUP, WR_CLOCK_WARMUP_MS="3")
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    ms = int(r.stdout.split("warmup_ms")[1].split()[0])
    assert ms >= 3, r.stdout


@pytest.mark.gpu
def test_cli_serial_and_pipelined_same_files(tmp_path):
    """WR_CLI_PIPELINE=0: wrenc / wrdec code one field at a time (the reference's loop, gen_enc.cpp:281-372); the default
    keeps several in flight.  Same .wrh, .wrb and decoded file, on a file of four fields of three kinds."""
    import json
    import test_cli
    with open(os.path.join(GOLDEN, "cli.json")) as fh:
        g = json.load(fh)
    enc, dec = os.path.join(test_cli.BINDIR, "wrenc"), os.path.join(test_cli.BINDIR, "wrdec")
    for pipeline in ("0", "3"):
        test_cli.run_case("inmeta_new_type0", enc, dec, g["inmeta_new_type0"], WR_CLI_PIPELINE=pipeline)


@pytest.mark.gpu
def test_flusi_tools_timing_lines(tmp_path):
    """WR_CLI_TIMING=1: the FluSI tools print one line per dataset with its read / codec call / write phases on stderr (what
    tools/flusi_rate.py builds its overlap evidence from); without it stderr stays empty."""
    import shutil
    import test_flusi
    if test_flusi.H5ROOT is None or shutil.which("gcc") is None:
        pytest.skip("no HDF5 C library / gcc for the test helper")
    from waverange_amd import synth
    h5tool = str(tmp_path / "h5tool")
    subprocess.check_call(["gcc", "-O1", "-I" + os.path.join(test_flusi.H5ROOT, "include"), os.path.join(ROOT, "tests", "native", "h5tool.c"),
                           "-o", h5tool, "-L" + os.path.join(test_flusi.H5ROOT, "lib"), "-lhdf5", "-Wl,-rpath," + os.path.join(test_flusi.H5ROOT, "lib")])
    nx, ny, nz = 32, 24, 16
    args = []
    for name, seed in (("ux", 1), ("uy", 2), ("uz", 3)):
        synth.field(nx, ny, nz, seed=seed).tofile(str(tmp_path / (name + ".raw")))
        args += [name, str(tmp_path / (name + ".raw"))]
    subprocess.check_call([h5tool, "make", str(tmp_path / "backup.h5"), "backup", "8", str(nx), str(ny), str(nz)] + args)
    for timing in ("1", None):
        env = dict(os.environ, WR_QUIET="1")
        if timing:
            env["WR_CLI_TIMING"] = timing
        r = subprocess.run([os.path.join(test_flusi.BIN, "wrenc_flusi"), "backup.h5", "comp%s.h5" % (timing or "0"), "1", "1e-16"], cwd=str(tmp_path), env=env,
                           capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
        lines = [l for l in r.stderr.splitlines() if l.startswith("timing dset=")]
        assert (len(lines) == 3 and all("slowest plane coder" in l for l in lines)) if timing else not lines, r.stderr
    # (the two files differ in the HDF5 object headers' modification times only: the payload is compared in test_flusi.py)
    assert os.path.getsize(tmp_path / "comp1.h5") == os.path.getsize(tmp_path / "comp0.h5")
