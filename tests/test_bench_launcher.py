"""bench.py --gpus N started without a launcher becomes the parent of N ranks: the ranks get the
torch.distributed.run environment, nothing GPU-related runs in the parent, the exit status is the ranks'."""
import json
import os
import subprocess
import sys

from util import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


def test_gpus_flag_starts_that_many_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=_clean_env(), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == [0, 1]
    assert all(l["world_size"] == 2 and l["launched_by"] == "bench.py" and l["master"].startswith("127.0.0.1:") for l in lines)
    assert len({l["master"] for l in lines}) == 1
    # every rank also says what it would take of the host (lanes, CPUs, host-memory share): the rehearsal of an N-rank sizing
    assert all(l["sizing"]["lanes"] >= 2 and "cpus_per_rank" in l["sizing"] and "host_mem_per_rank_gib" in l["sizing"] for l in lines)


def test_under_an_external_launcher_it_is_one_rank():
    env = dict(_clean_env(), RANK="1", LOCAL_RANK="1", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29999")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1 and lines[0]["rank"] == 1 and lines[0]["launched_by"] == "external launcher"


def test_mismatch_and_missing_gpus_fail_loudly():
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="2")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--dry-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr
    import torch
    if torch.cuda.device_count() < 3:
        r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--steps", "1"], env=_clean_env(), capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and "GPU(s) visible" in r.stderr


def test_a_failing_rank_fails_the_launch():
    # without a GPU (or without the gloo rehearsal backend on a 1-GPU box) a real rank cannot start: the parent must
    # report that instead of hanging or printing a line
    env = dict(_clean_env(), WR_BENCH_BACKEND="gloo", CUDA_VISIBLE_DEVICES="", HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--size", "64"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not any(l.startswith('{"metric"') for l in r.stdout.splitlines())


def test_handover_buffers_are_reused_by_tolerance():
    """bench.py's hand-over buffers: a field takes the buffer its own tolerance gave back last (so that the small streams of
    a loose tolerance never grow into the 2 GB a tight one leaves resident), a new one if there is none, and a leak is an
    error instead of unbounded growth."""
    import importlib.util
    import pytest
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    made = []
    pool = bench.HandoverBuffers(lambda: made.append(object()) or made[-1], limit=5)
    a, b = pool.take(1e-3), pool.take(1e-7)
    assert a is not b and len(pool.all) == 2
    pool.give(1e-7, b)
    c = pool.take(1e-3)                     # nothing idle for 1e-3: a new buffer, not the 1e-7 one
    assert c is not b and len(pool.all) == 3
    assert pool.take(1e-7) is b             # the tight tolerance gets its own buffer back
    pool.give(1e-3, a); pool.give(1e-3, c)
    assert pool.take(1e-3) is c and pool.take(1e-3) is a   # LIFO: the one given back last first
    pool.take(1e-5); pool.take(1e-5)
    with pytest.raises(RuntimeError):
        pool.take(1e-5)
    pool.reset(1e-16)
    assert {id(pool.take(1e-16)) for _ in range(5)} == {id(x) for x in pool.all}


def _bench_module():
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", BENCH)
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_rank_sizing_in_three_host_memory_regimes():
    """fit_jobs on the recorded 8-GPU topology (16 cores = 32 hardware threads and an eighth of the node's memory per rank,
    1024^3 fields of 8.6 GB, tolerances 1e-3 + 1e-7): a rank never sizes itself beyond its share of the host memory.
      ample       2 TiB node: the lanes the CPUs want, coded streams stay resident, four output fields
      tight       300 GiB node (37.5 GiB per rank: what the one-GPU box's cgroup gives an 8-rank rehearsal): one lane per
                  tolerance, consumed streams hand their pages back, ONE output field -- and what that takes fits the share
      impossible  128 GiB node (16 GiB per rank): refused with the arithmetic in the message, nothing allocated;
                  fit_jobs_or_smaller then sizes the rank for 512^3 fields and says that it did."""
    import pytest
    bench = _bench_module()
    fb = 1024 ** 3 * 8
    kw = dict(pinned_share=32, host_mode=True, pooled=True, out_pool=4, gpus_on_node=8, cpus=32.0, local_world=8)
    jobs, lim = bench.fit_jobs(16, 2, fb, int(0.97 * 288e9), host_mem=2048 << 30, **kw)
    assert lim["host_memory_regime"] == "ample" and jobs == min(16, lim["jobs_by_hbm"]) and lim["out_buffers"] == 4 and not lim["host_pages_of_consumed_streams_dropped"]
    # (1.2 TiB: all 32 lanes still fit, but only with the pages of consumed streams handed back)
    jobs, lim = bench.fit_jobs(16, 2, fb, int(0.97 * 288e9), host_mem=1200 << 30, **kw)
    assert lim["host_memory_regime"] == "tight" and jobs == min(16, lim["jobs_by_hbm"]) and lim["out_buffers"] == 4 and lim["host_pages_of_consumed_streams_dropped"]
    jobs, lim = bench.fit_jobs(16, 2, fb, int(0.97 * 288e9), host_mem=300 << 30, **kw)
    assert lim["host_memory_regime"] == "tight" and lim["out_buffers"] == 1 and lim["host_pages_of_consumed_streams_dropped"]
    assert 1 <= jobs <= 3
    used = (1 + lim["out_buffers"]) * fb + jobs * 2 * 0.25 * fb   # what main() allocates from this sizing
    assert used <= 0.8 * (300 << 30) / 8, (jobs, used / 2 ** 30)
    with pytest.raises(bench.SizingRefused) as exc:
        bench.fit_jobs(16, 2, fb, int(0.97 * 288e9), host_mem=128 << 30, **kw)
    msg = str(exc.value)
    assert "128.0 GiB available / 8 ranks" in msg and "needs" in msg and "smaller --size" in msg
    size, jobs, lim = bench.fit_jobs_or_smaller(1024, 16, 2, fb, int(0.97 * 288e9), host_mem=128 << 30, **kw)
    assert size == 512 and lim["fell_back_from_size"] == 1024 and jobs >= 1 and "128.0 GiB" in lim["refusal_at_that_size"]
    with pytest.raises(bench.SizingRefused):   # a node on which not even 512^3 fits: still a refusal, never max(1, negative)
        bench.fit_jobs_or_smaller(1024, 16, 2, fb, int(0.97 * 288e9), host_mem=16 << 30, **kw)
    with pytest.raises(bench.SizingRefused):   # HBM too: three work-space slots of 2.2 field sizes do not fit 40 GB
        bench.fit_jobs(16, 2, fb, 40 * 10 ** 9, host_mem=2048 << 30, **kw)


def test_dry_launch_of_eight_ranks_prints_a_sizing_every_rank_can_allocate_or_refuses():
    """`bench.py --gpus 8 --dry-launch` on a node of 300 GiB (the memory given through the test hook): every rank prints a
    sizing that fits its eighth; on a node of 16 GiB every rank refuses and the launch fails."""
    env = dict(_clean_env(), WR_BENCH_TEST_HOST_MEM_GIB="300")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")]
    assert sorted(l["rank"] for l in lines) == list(range(8))
    for l in lines:
        s = l["sizing"]
        fb = s["size"] ** 3 * 8
        assert s["lanes"] >= 2 and s["jobs_by_host_mem"] >= 1
        assert (1 + s["out_buffers"]) * fb + s["lanes"] * (0.25 if s["host_pages_of_consumed_streams_dropped"] else 0.6) * fb <= 0.8 * (300 << 30) / 8 + 1
    env = dict(_clean_env(), WR_BENCH_TEST_HOST_MEM_GIB="16")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--dry-launch"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "cannot hold the workload" in (r.stdout + r.stderr)


def test_every_rank_gets_the_ipc_mode_and_the_fault_log():
    """A rank started by an external launcher (the driver's torch.distributed.run) has no HSA_ENABLE_IPC_MODE_LEGACY unless its
    parent exported it: bench.py sets it (and WR_FAULT_LOG) itself, before anything touches the GPU -- checked through the
    dry launch, which prints what the rank's environment holds at that point."""
    env = dict(_clean_env(), RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29998")
    env.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)
    env.pop("WR_FAULT_LOG", None)
    r = subprocess.run([sys.executable, BENCH, "--dry-launch"], env=env, capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    line = [json.loads(l) for l in r.stdout.splitlines() if l.startswith("{")][0]
    assert line["environment"] == {"HSA_ENABLE_IPC_MODE_LEGACY": "0", "WR_FAULT_LOG": "1"}
