"""Property-based checks (hypothesis) on the CPU side: product range coder == oracle range coder on
arbitrary planes, stream anchors, transform round trips of the oracle on arbitrary small shapes."""
import os

import pytest
import numpy as np
from hypothesis import given, settings, strategies as st

from oracle.loader import Oracle
from waverange_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_o = Oracle()


@settings(max_examples=60, deadline=None)
@given(st.integers(1, 70000), st.integers(0, 2 ** 31 - 1), st.sampled_from(["uniform", "two", "skew", "runs"]))
def test_range_coder_matches_oracle(n, seed, kind):
    rs = np.random.RandomState(seed)
    if kind == "uniform":
        p = rs.randint(0, 256, n)
    elif kind == "two":
        p = rs.choice([3, 250], size=n, p=[0.97, 0.03])
    elif kind == "skew":
        p = np.minimum(rs.geometric(rs.uniform(0.02, 0.9), n), 255)
    else:
        p = np.repeat(rs.randint(0, 256, n // 50 + 1), 50)[:n]
    p = p.astype(np.uint8)
    s = api.range_encode(p)
    assert np.array_equal(s, _o.range_encode(p))
    assert s[0] == 0 and (int(s[-3]) << 16 | int(s[-2]) << 8 | int(s[-1])) == s.size % (1 << 24)
    back, got = api.range_decode(s, n)
    assert got == n and np.array_equal(back, p)


@settings(max_examples=40, deadline=None)
@given(st.integers(1, 40), st.integers(1, 24), st.integers(1, 20), st.integers(0, 1000))
def test_oracle_transform_round_trip(nx, ny, nz, seed):
    rs = np.random.RandomState(seed)
    f = rs.standard_normal((nz, ny, nx))
    w = _o.cdf97_3d(f, 4)
    r = _o.cdf97_3d(w, -4)
    assert np.abs(r - f).max() <= 1e-12 * max(1.0, np.abs(f).max())
    # linearity of the transform (it is a linear map; round-off level agreement)
    g = rs.standard_normal((nz, ny, nx))
    lhs = _o.cdf97_3d(f + 2.0 * g, 4)
    rhs = w + 2.0 * _o.cdf97_3d(g, 4)
    assert np.abs(lhs - rhs).max() <= 1e-11 * max(1.0, np.abs(lhs).max())


def test_bench_batch_sizing():
    """bench.py sizes its batch (fields in flight) to the rank's CPUs, host memory and free HBM."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    fb = 1024 ** 3 * 8
    big = 2048 * 2 ** 30  # (host memory given, so that the container's own free memory does not decide the test)
    # fields resident in HBM: two field buffers and the planes of two contexts per lane next to the three work-space slots
    jobs, lim = bench.fit_jobs(5, 2, fb, 288 * 10 ** 9, host_mode=False, host_mem=big)
    assert 1 <= jobs <= 5 and jobs <= max(1, lim["jobs_by_cpu"]) and jobs <= max(1, lim["jobs_by_hbm"])
    assert lim["jobs_by_hbm"] == int((0.92 * 288e9 - 3 * 2.2 * fb) // (3.0 * fb * 2))
    # HBM nearly full: refused with the arithmetic (round 4 clamped a negative budget to one job and let the rank die allocating)
    with pytest.raises(bench.SizingRefused):
        bench.fit_jobs(5, 2, fb, 40 * 10 ** 9, host_mode=False, host_mem=big)
    # host buffer to host buffer (the default): the planes of the fields in flight live in HBM -- a decoder's until its field is
    # done (and only from the moment it is admitted to the coder pool), an encoder's draining chunk by chunk: 0.6 of the two
    # contexts' worst case per lane (--hbm-per-lane) -- the host holds their coded streams
    jobs, lim = bench.fit_jobs(64, 2, fb, 288 * 10 ** 9, host_mem=big)
    assert lim["jobs_by_hbm"] == int((0.92 * 288e9 - 3 * 2.2 * fb) // (0.6 * fb * 2)) and jobs <= max(1, lim["jobs_by_host_mem"])
    # two slots leave room for more lanes; tol 1e-16 (8 planes per field) for half as many
    assert bench.fit_jobs(64, 2, fb, 288 * 10 ** 9, nslots=2, host_mem=big)[1]["jobs_by_hbm"] == int((0.92 * 288e9 - 2 * 2.2 * fb) // (0.6 * fb * 2))
    assert bench.fit_jobs(64, 1, fb, 288 * 10 ** 9, planes_per_field=8, host_mem=big)[1]["jobs_by_hbm"] == int((0.92 * 288e9 - 3 * 2.2 * fb) // (1.2 * fb))
    # ... and host memory for two field sizes per lane (coded streams of most of a field size, held twice): on a box that allows
    # 270 GiB, 11 lanes -- sized like a 4-plane run (0.6) it took more than that and the box killed it (round 5)
    jobs, lim = bench.fit_jobs(20, 1, fb, 288 * 10 ** 9, planes_per_field=8, pooled=True, host_mem=270 * 2 ** 30, cpus=16)
    assert lim["jobs_by_host_mem"] == int((0.8 * 270 * 2 ** 30 - 5 * fb) // (2.0 * fb)) == 11 and jobs == 11
    # with the coder pool a lane is not a thread: two and a half fields in flight per CPU (--fields-per-cpu)
    jobs, lim = bench.fit_jobs(64, 2, fb, 288 * 10 ** 9, pooled=True, host_mem=big)
    assert lim["jobs_by_cpu"] == max(1, int(2.5 * lim["cpus_per_rank"] // 2))
    assert bench.fit_jobs(64, 2, fb, 288 * 10 ** 9, pooled=True, fields_per_cpu=1.5, host_mem=big)[1]["jobs_by_cpu"] == max(1, int(1.5 * lim["cpus_per_rank"] // 2))
    # where host memory is what holds the lanes back, consumed coded streams hand their pages back and more lanes fit
    plenty, tight = 2048 * 2 ** 30, 120 * 2 ** 30
    jobs, lim = bench.fit_jobs(16, 2, fb, 288 * 10 ** 9, pooled=True, host_mem=plenty)
    assert not lim["host_pages_of_consumed_streams_dropped"] and lim["jobs_by_host_mem"] == int((0.8 * plenty - 5 * fb) // (0.6 * fb * 2))
    jobs, lim = bench.fit_jobs(16, 2, fb, 288 * 10 ** 9, pooled=True, host_mem=tight)
    assert lim["host_pages_of_consumed_streams_dropped"] and lim["jobs_by_host_mem"] == int((0.8 * tight - 5 * fb) // (0.25 * fb * 2))
    assert jobs == min(16, lim["jobs_by_cpu"], lim["jobs_by_host_mem"], lim["jobs_by_hbm"]) >= 1


def test_bench_cpu_share_pinning():
    """bench.py pins a rank to its GPU's slice of the allowed CPUs on multi-GPU nodes (in a child process:
    the affinity change must not leak into the test runner)."""
    import subprocess
    import sys
    code = ("import os, sys; sys.path.insert(0, %r); import bench; a = sorted(os.sched_getaffinity(0)); "
            "n = bench.take_cpu_share(1, 2); b = sorted(os.sched_getaffinity(0)); "
            "print(len(a), n, b == a[len(a)//2:2*(len(a)//2)] if n else b == a); "
            "print(bench.take_cpu_share(0, 1))" % ROOT)
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, check=True).stdout.split("\n")
    total, share, ok = out[0].split()
    assert ok == "True"
    if int(total) >= 16:
        assert int(share) == int(total) // 2
    else:
        assert share == "None"  # fewer than 8 CPUs per GPU: left alone
    assert out[1] == "None"  # a single visible GPU: nothing to share


def test_bench_cpu_share_follows_cores_and_numa_nodes():
    """cpu_share_of on the topology of the MI355X boxes of this pool (profiles/r03/ao_cpu_probe_gpu_box.txt: 2 x 64 cores,
    hardware threads c and c + 128 share a core, node 0 = cores 0-63, node 1 = cores 64-127, four GPUs on either node):
    every rank gets whole cores of its GPU's node, no two ranks meet on a core, all CPUs are handed out; without the
    GPUs' nodes it falls back to slices of whole cores; small CPU sets and single GPUs are left alone."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    allowed = list(range(256))
    topo = {"allowed": allowed, "siblings": {c: (c % 128, c % 128 + 128) for c in allowed},
            "node_of_cpu": {c: (c % 128) // 64 for c in allowed}, "gpu_nodes": [0, 0, 1, 1, 0, 1, 1, 0]}
    shares = [bench.cpu_share_of(r, 8, topo) for r in range(8)]
    seen = set()
    for r, cpus in enumerate(shares):
        assert len(cpus) == 32 and not seen & set(cpus)
        seen |= set(cpus)
        cores = {c % 128 for c in cpus}
        assert len(cores) == 16 and all(c % 128 + 128 in cpus and c % 128 in cpus for c in cpus)  # whole cores
        assert {topo["node_of_cpu"][c] for c in cpus} == {topo["gpu_nodes"][r]}                   # on the GPU's node
    assert seen == set(allowed)
    # one rank alone on the node takes the same share as with eight (weak scaling: the host behind a GPU does not change)
    assert bench.cpu_share_of(0, 8, topo) == shares[0]
    # nodes unknown: slices of whole cores in device order
    blind = dict(topo, gpu_nodes=[-1] * 8)
    got = [bench.cpu_share_of(r, 8, blind) for r in range(8)]
    assert got[0] == sorted(list(range(0, 16)) + list(range(128, 144))) and got[7] == sorted(list(range(112, 128)) + list(range(240, 256)))
    assert sorted(c for g in got for c in g) == allowed
    # all GPUs on one node of a two-node host whose CPU set spans both: that node's cores are split among them
    onenode = dict(topo, gpu_nodes=[1] * 8)
    got = [bench.cpu_share_of(r, 8, onenode) for r in range(8)]
    assert all(len(g) == 16 and {topo["node_of_cpu"][c] for c in g} == {1} for g in got) and len({c for g in got for c in g}) == 128
    # a CPU set that was cut down without regard to the nodes (all of it on node 0, GPU on node 1): plain slices
    cut = {"allowed": list(range(32)), "siblings": {c: (c,) for c in range(32)}, "node_of_cpu": {c: 0 for c in range(32)}, "gpu_nodes": [0, 1]}
    assert bench.cpu_share_of(1, 2, cut) == list(range(16, 32)) and bench.cpu_share_of(0, 2, cut) == list(range(16))
    # too few CPUs per GPU, or one GPU: left alone
    assert bench.cpu_share_of(0, 8, {"allowed": list(range(32)), "siblings": {}, "node_of_cpu": {}, "gpu_nodes": []}) is None
    assert bench.cpu_share_of(0, 1, topo) is None
