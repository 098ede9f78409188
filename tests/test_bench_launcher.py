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
