#!/usr/bin/env python3
"""bench.py -- WaveRange hot path on MI355X: encode + decode of 3-D fp64 fields, host buffer to
host buffer (BASELINE.json metric: "encode+decode MB/s per GPU on 1024^3 fp64; % HBM roofline; L-inf vs
tol"; SURVEY.md 8d(1): the reference API takes and returns host arrays).

One step = one batch of fields per GPU (default 6 per tolerance setting = 12), K steps = K batches.  The fields of a
run are pulled from one queue by jobs x len(tols) lanes (default 20 x 2: 2.5 fields in flight per CPU of the rank, fewer
if its CPUs, host memory or HBM are short), so up to 40 fields are in flight and steps overlap; a lane is an encoder
context and a decoder context with two coded-stream buffers between them: it encodes its next field while it decodes
the previous one.  Every field starts in a pinned host buffer and is encoded (upload; min/max, forward CDF-9/7,
bit-plane quantizer on the GPU, the planes staying in HBM; rngcod13 range coder on the host, which pulls the planes
through pinned 15 MB windows -> coded bytes in host memory: wr_encode_host, what the drop-in encoding_wrap runs on)
and decoded again (range decoder on the host, its windows going straight to the planes' device buffers:
wr_decode_begin; dequantise + inverse transform on the GPU; download into a pinned host buffer:
wr_decode_finish_host -- decoding_wrap does the two in one call; in two calls the output field is only held for the
last quarter second, so four output buffers serve all lanes).  The range coder is one serial recurrence per plane
and runs on the host by design, so the whole-job rate is set by the host cores a GPU has (16 on this pool): the
plane streams of all fields in flight go to a pool of one coder thread per CPU, whose workers interleave 3-4
streams of any fields per symbol loop (dominant-symbol planes 16 at a time in an AVX-512 loop).  The device stages
of the fields (upload / kernels / download, three work-space slots per GPU, copies on the SDMA engines) overlap with
one another and with the host coding inside the library.
value = field megabytes (10^6 B) round-tripped per second, whole job (all ranks).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--size 1024] [--tols 1e-3,1e-7] [--batch 12] [--jobs 12] [--pool -1]
                  [--resident]   (fields start and end in HBM instead: the round-1 measurement)
                  [--pool 0 --threads 1]   (no coder pool: every call codes its planes on its own thread)

N > 1: one process per GPU (torch.distributed / RCCL for the timing barrier only); every rank
codes its own independent field (seed 12345 + rank): weak scaling, no data-path collective.  Either a
launcher starts the ranks (the driver: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...,
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment), or -- `python bench.py --gpus N` on its own --
this process starts them itself before it has touched the GPU, relays rank 0's line and exits with their status
(launch_ranks).  --dry-launch: the ranks only report who they are (CPU check of the launcher).
WR_BENCH_BACKEND=gloo rehearses N ranks on fewer GPUs (ranks share devices, barrier on CPU tensors).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8 TB/s spec


def box_elems(n, levels=4):
    """sum over levels of the corner-box sizes: each level reads and writes its box once."""
    tot, m = 0, n
    for _ in range(levels):
        tot += m ** 3
        m = (m + 1) // 2
    return tot


def measured_traffic(n):
    """HBM bytes per transform from the committed PMC passes (profiles/rNN/traffic_<n>.json: rocprofv3
    --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate runs, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950), mean of forward and inverse.  Counters cannot be collected from inside this
    process, so the figure is the latest committed one and carries its file name; (None, None) if absent."""
    import glob
    best, src, stale = None, None, None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "traffic_%d.json" % n))):
        try:
            with open(f) as fh:
                t = json.load(fh)
            best = 0.5 * (t["fwd"]["total_bytes"] + t["inv"]["total_bytes"])
            src = os.path.relpath(f, ROOT)
            # the file names the kernel sources it was measured on (tools/pmc_traffic.py); any edit since makes it stale
            stale = t.get("kernel_sources_sha256") != kernel_sources_sha256()
        except Exception:
            pass
    return best, src, stale


def kernel_sources_sha256(names=("wr_fused.hip",)):
    """SHA-256 of the HIP sources whose kernels a committed figure belongs to (default: the PMC traffic figure's k_fwd_fused,
    k_inv_fused)."""
    import hashlib
    out = {}
    for name in names:
        with open(os.path.join(ROOT, "waverange_amd", "csrc", name), "rb") as fh:
            out[name] = hashlib.sha256(fh.read()).hexdigest()
    return out


def sha_big(a, chunk=1 << 28):
    import hashlib
    h = hashlib.sha256()
    b = a.reshape(-1).view("uint8")
    for o in range(0, b.size, chunk):
        h.update(b[o:o + chunk])
    return h.hexdigest()


def pin_key(n, tol, seed):
    """key of a pin in tests/golden/large.json (tools/make_golden_large.py): the bench's own field (seed 12345) has the short
    form, rank r of an N-rank run codes seed 12345 + r"""
    return "%d^3_tol%g" % (n, tol) + ("" if seed == 12345 else "_seed%d" % seed)


def parity_vs_pins(n, seed, coded, decoded):
    """What the TIMED run produced against the reference's own outputs (tests/golden/large.json, written by
    tools/make_golden_large.py from the compiled reference in the build container): the coded bytes of the last field
    of every tolerance (header scalars as bit patterns, plane lengths, SHA-256 of data_enc) and the last reconstruction.
    Runs after the timed region, on every rank for its own field.  None where no pin exists for (size, tolerance, seed)."""
    try:
        with open(os.path.join(ROOT, "tests", "golden", "large.json")) as fh:
            pins = json.load(fh)
    except OSError:
        return {"fixture": None}
    out = {"fixture": "tests/golden/large.json", "seed": seed, "coded_sha_ok": None, "decoded_sha_ok": None, "checked": []}
    for tol, enc in coded.items():
        rec = pins.get(pin_key(n, tol, seed))
        if not rec or rec["seed"] != seed:
            continue
        ok = (enc["nlay"] == rec["nlay"] and [int(v) for v in enc["len_enc_vec"]] == rec["len_enc_vec"] and int(enc["ntot_enc"]) == rec["ntot_enc"]
              and all(float(enc[k]).hex() == rec[k] for k in ("tolabs", "midval", "halfspanval"))
              and [float(v).hex() for v in enc["deps_vec"]] == rec["deps_vec"] and [float(v).hex() for v in enc["minval_vec"]] == rec["minval_vec"]
              and sha_big(enc["data"][:rec["ntot_enc"]]) == rec["data_sha256"])
        out["coded_sha_ok"] = ok if out["coded_sha_ok"] is None else (out["coded_sha_ok"] and ok)
        out["checked"].append("coded %d^3 tol %g" % (n, tol))
    if decoded is not None:
        tol, arr = decoded
        rec = pins.get(pin_key(n, tol, seed))
        if rec and rec["seed"] == seed:
            out["decoded_sha_ok"] = sha_big(arr) == rec["decoded_sha256"]
            out["checked"].append("reconstruction %d^3 tol %g" % (n, tol))
    return out


def committed_kernel_stats():
    """The rocprofv3 --kernel-trace --stats summary committed under profiles/ (latest round's final_kernel_stats.csv, copied
    there by tools/kernel_stats_meta.py together with the hashes of the kernel sources it was taken on): per-transform
    durations of the same kernels this run times with HIP events, and whether the sources have changed since (stale)."""
    import csv
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", "final_kernel_stats.csv")))
    if not files:
        return None
    f = files[-1]
    out = {"source": os.path.relpath(f, ROOT), "stale": None}
    try:
        with open(f[:-4] + ".meta.json") as fh:
            meta = json.load(fh)
        out["stale"] = meta.get("kernel_sources_sha256") != kernel_sources_sha256(("wr_fused.hip", "wr_kernels.hip"))
        out["taken_on"] = meta.get("head")
        out["clock_warmup_ms"] = meta.get("clock_warmup_ms")
    except (OSError, ValueError):
        pass
    try:
        avg, calls = {}, {}
        with open(f) as fh:
            for r in csv.DictReader(fh):
                avg[r["Name"]], calls[r["Name"]] = float(r["AverageNs"]) * 1e-6, int(r["Calls"])

        def pick(sub):
            return [(k, avg[k], calls[k]) for k in avg if sub in k]
        # an encode's forward transform is one <true, true> launch (level 0, min/max riding along) and three <false, true>
        # ones; an inverse is four k_inv_fused launches (all levels in one row of the stats)
        f0, f1, inv = pick("k_fwd_fused<true, true>"), pick("k_fwd_fused<false, true>"), pick("k_inv_fused")
        if f0 and f1:
            out["fwd_ms"] = round(f0[0][1] + 3.0 * f1[0][1], 3)
        if inv:
            out["inv_ms"] = round(4.0 * inv[0][1], 3)
        burn = pick("k_burn")
        out["k_burn_calls"] = burn[0][2] if burn else 0
    except (OSError, ValueError, KeyError):
        pass
    return out


def committed_extra(name):
    """A bench line committed under profiles/ (latest round), e.g. the resident-field variant of this run."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*", name)), reverse=True):
        try:
            with open(f) as fh:
                return json.load(fh), os.path.relpath(f, ROOT)
        except Exception:
            pass
    return None, None


def _cgroup_number(path):
    try:
        with open(path) as fh:
            return fh.read().split()
    except OSError:
        return None


def _cpulist(text):
    out = []
    for part in text.strip().split(","):
        if part:
            lo, _, hi = part.partition("-")
            out += list(range(int(lo), int(hi or lo) + 1))
    return out


def read_cpu_topology(gpu_bdfs=()):
    """What cpu_share_of needs, from sysfs: the CPUs this process may run on, the hardware threads that share a core with
    each of them, the NUMA node of every CPU and of every GPU (PCI address "dddd:bb:dd.f"; -1: unknown)."""
    import glob
    allowed = sorted(os.sched_getaffinity(0))
    siblings, node_of_cpu = {}, {}
    for c in allowed:
        try:
            with open("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c) as fh:
                siblings[c] = tuple(_cpulist(fh.read()))
        except (OSError, ValueError):
            siblings[c] = (c,)
    for d in glob.glob("/sys/devices/system/node/node[0-9]*"):
        try:
            with open(d + "/cpulist") as fh:
                for c in _cpulist(fh.read()):
                    node_of_cpu[c] = int(d.rsplit("node", 1)[1])
        except (OSError, ValueError):
            pass
    gpu_nodes = []
    for bdf in gpu_bdfs:
        try:
            with open("/sys/bus/pci/devices/%s/numa_node" % bdf) as fh:
                gpu_nodes.append(int(fh.read().strip()))
        except (OSError, ValueError):
            gpu_nodes.append(-1)
    return {"allowed": allowed, "siblings": siblings, "node_of_cpu": node_of_cpu, "gpu_nodes": gpu_nodes}


def cpu_share_of(local_rank, gpus_on_node, topo):
    """The CPUs of GPU `local_rank`'s share of the host: whole physical cores (all their hardware threads: two ranks never
    meet on one core), 1 / gpus_on_node of them -- taken from the GPU's own NUMA node when the nodes of all GPUs are known,
    split evenly among the GPUs of that node; otherwise the local_rank-th slice of all cores.  None: the CPU set is too
    small to have been sized for the node (< 8 CPUs per GPU), or a single GPU."""
    allowed = set(topo["allowed"])
    if gpus_on_node < 2 or len(allowed) // gpus_on_node < 8:
        return None
    cores = sorted({tuple(sorted(c for c in topo["siblings"].get(cpu, (cpu,)) if c in allowed)) for cpu in allowed})
    d = local_rank % gpus_on_node
    gpu_nodes = list(topo.get("gpu_nodes") or [])[:gpus_on_node]
    node_of = topo.get("node_of_cpu") or {}
    even = len(cores) // gpus_on_node
    mine = None
    if len(gpu_nodes) == gpus_on_node and all(g >= 0 for g in gpu_nodes):
        # by node -- for all GPUs or for none: every node must give each of its GPUs at least half of an even share
        # (else: a CPU set that was cut down without regard to the nodes)
        plan = []
        for g in range(gpus_on_node):
            peers = [h for h in range(gpus_on_node) if gpu_nodes[h] == gpu_nodes[g]]
            local = [cs for cs in cores if node_of.get(cs[0], -1) == gpu_nodes[g]]
            per = len(local) // len(peers)
            plan.append(local[peers.index(g) * per:(peers.index(g) + 1) * per])
        if all(len(p) >= max(1, even // 2) for p in plan):
            mine = plan[d]
    if mine is None:
        mine = cores[d * even:(d + 1) * even]
    cpus = sorted(c for cs in mine for c in cs)
    return cpus or None


def take_cpu_share(local_rank, gpus_on_node, gpu_bdfs=()):
    """Pin this process -- the threads it has (runtime helpers) and the ones it starts later -- to its GPU's share of the
    host CPUs (cpu_share_of).  A rank then has the same host behind its GPU whether 1 or 8 ranks run on the node (weak
    scaling measures GPUs, not how many idle cores one rank can borrow), never shares a physical core with another rank,
    and codes on the memory of its GPU's NUMA node.  No-op with a single visible GPU or when anything about it fails."""
    try:
        mine = cpu_share_of(local_rank, gpus_on_node, read_cpu_topology(gpu_bdfs))
        if not mine:
            return None
        os.sched_setaffinity(0, mine)
        for tid in os.listdir("/proc/self/task"):
            try:
                os.sched_setaffinity(int(tid), mine)
            except OSError:
                pass
        return len(mine)
    except (OSError, ValueError, AttributeError):
        return None


class HandoverBuffers:
    """The coded streams travel from a lane's encoder to its decoder in hand-over buffers, each large enough for any field
    (setup_wr's bound: virtual memory, only coded bytes are ever touched).  They are not tied to lanes: a buffer that has held
    a tight-tolerance field keeps that field's 2 GB resident, so a field takes the buffer its own tolerance gave back last (a
    LIFO per tolerance; a new one if there is none) -- the few-hundred-MB streams of the loose tolerance keep meeting small
    buffers, and what stays resident is what each tolerance has in flight at a time instead of 2 GB in every buffer there
    is (165 -> 137 GiB at 32 lanes and 20 steps, profiles/r04/r_*)."""

    def __init__(self, make, limit):
        import collections
        import threading
        self._deque = collections.deque
        self.make, self.limit = make, limit
        self.all = []      # every buffer ever made
        self.idle = {}     # tolerance -> deque of idle buffers that held a field of it last
        self.lock = threading.Lock()

    def take(self, tol):
        with self.lock:
            mine = self.idle.setdefault(tol, self._deque())
            if mine:
                return mine.pop()
            if len(self.all) >= self.limit:
                raise RuntimeError("bench.py: hand-over buffers leak (%d made; two per lane are in flight at most)" % len(self.all))
            self.all.append(self.make())
            return self.all[-1]

    def give(self, tol, buf):
        with self.lock:
            self.idle.setdefault(tol, self._deque()).append(buf)

    def reset(self, tol):
        """every buffer idle again, for fields of `tol` (nothing may be in flight)"""
        with self.lock:
            self.idle = {tol: self._deque(self.all)}


class SizingRefused(SystemExit):
    """this rank cannot run the workload in the host memory it has: the message carries the arithmetic"""


def fit_jobs(want, ntols, field_bytes, hbm_free, pinned_share=None, host_mode=True, pooled=False, out_pool=4, gpus_on_node=1, nslots=3, planes_per_field=4,
             fields_per_cpu=2.5, host_mem=None, cpus=None, local_world=None, hbm_per_lane=0.6):
    """Largest jobs <= want whose lanes (jobs x ntols) fit this rank's share of the host CPUs, the host
    memory and the free HBM.  Returns (jobs, {what was found}).  Host memory decides in three steps (a rank never allocates
    what its share cannot hold: an 8-GPU node with little memory per GPU must not turn its first run into an OOM kill):
      ample       the lanes the CPUs want fit with coded streams left resident (0.6 field sizes per lane) next to the fixed
                  buffers (the input field and `out_pool` output fields);
      tight       fewer lanes, consumed coded streams hand their pages back (0.25 field sizes per lane); if not even one lane
                  per tolerance fits, the output pool shrinks to ONE field (decodes then take turns for it);
      impossible  2 + 0.25 x ntols field sizes do not fit 80 % of the share: SizingRefused with the arithmetic (the caller
                  may retry at a smaller --size; main() does so once, loudly, for --size 1024 -> 512)."""
    if local_world is None:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1")))
    quota = None  # CPU quota of the whole job (all ranks), if any
    if cpus is None:
        cpus = float(len(os.sched_getaffinity(0)))
        q = _cgroup_number("/sys/fs/cgroup/cpu.max")  # cgroup v2
        if q and q[0] != "max":
            quota = float(q[0]) / float(q[1])
        q1, p1 = _cgroup_number("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), _cgroup_number("/sys/fs/cgroup/cpu/cpu.cfs_period_us")  # v1
        if q1 and p1 and float(q1[0]) > 0:
            quota = min(quota, float(q1[0]) / float(p1[0])) if quota else float(q1[0]) / float(p1[0])
        if pinned_share:
            # the affinity has been cut down to this rank's GPU's share of the node; a quota is split the same way,
            # whether 1 or 8 ranks run
            cpus = min(float(pinned_share), quota / gpus_on_node) if quota else float(pinned_share)
        else:
            cpus = (min(cpus, quota) if quota else cpus) / local_world
    # host memory like the CPUs: a rank sizes itself to its GPU's share of the node whether 1 or 8 ranks run, so that
    # the per-GPU work does not change with N (weak scaling measures GPUs, not how much idle memory one rank can borrow)
    mem_share = max(local_world, gpus_on_node if pinned_share else 1)
    mem = host_mem  # (given: tests)
    if mem is None:
        try:
            with open("/proc/meminfo") as fh:
                mem = [int(l.split()[1]) * 1024 for l in fh if l.startswith("MemAvailable")][0]
        except (OSError, IndexError):
            pass
        for path in ("/sys/fs/cgroup/memory.max", "/sys/fs/cgroup/memory/memory.limit_in_bytes"):  # cgroup v2, v1
            m = _cgroup_number(path)
            if m and m[0] != "max" and int(m[0]) < 1 << 60:
                mem = min(mem, int(m[0])) if mem else int(m[0])
    # 2 coder threads per field; an encoder thread idles a third of the time (decoding takes longer and
    # sets the period of a lane), hence 1.25 threads per CPU
    # with the coder pool the lanes are not threads: a field in flight spends part of its time in copies and kernels and
    # waiting for its slowest plane, so 1.5 fields in flight per CPU keep the pool's workers busy (16 lanes on 16 CPUs:
    # 12.2 CPUs busy on average)
    by_cpu = max(1, int(fields_per_cpu * cpus // ntols) if pooled else int(1.25 * cpus // (2 * ntols)))
    # host memory per field in flight: the coded streams only (two hand-over buffers that have each held a 1e-7 field's
    # 2 GB at some point, and the coder's output while it is being produced: ~0.6 field sizes at the bench's tolerances,
    # 0.75 budgeted) and the pinned rings (0.25 GiB) -- the quantized planes stay in HBM; the input
    # field and the pool of output fields (two-phase decode) are a fixed 1 + out_pool field sizes in host-to-host mode.
    # HBM per field in flight: the planes of its encoder and decoder contexts (up to 4 + 4 at the bench's tolerances =
    # one field size) next to three work-space slots of 2.2 field sizes; resident mode: two field buffers per lane more
    # (beyond four planes per field -- `--tols 1e-16`: eight, six of them at a byte per symbol -- a field's coded streams are most
    # of a field size and every lane holds them twice, in the encoder's output and in a hand-over buffer: two field sizes per lane;
    # sized at 0.6 such a run took more than the 270 GiB a box allows and was killed, profiles/r05/NOTES.md)
    # (what handing consumed pages back saves there has not been measured: no discount for it)
    lane_ample, lane_tight = (0.6, 0.25) if planes_per_field <= 4 else (0.25 * planes_per_field, 0.25 * planes_per_field)
    regime, trim, by_mem, budget = "ample", False, want, None
    if mem:
        budget = 0.8 * mem / mem_share
        fixed = (1 + out_pool) * field_bytes if host_mode else 0
        by_mem = int((budget - fixed) // (lane_ample * field_bytes * ntols))
        if by_mem < min(want, by_cpu):
            # consumed coded streams hand their pages back to the system (drop_pages): ~0.25 field sizes per lane instead of
            # 0.6, at ~3 % of the rate (the pages are faulted in and zeroed again for every field)
            regime, trim = "tight", True
            by_mem = int((budget - fixed) // (lane_tight * field_bytes * ntols))
            if by_mem < 1 and host_mode and out_pool > 1:
                out_pool = 1
                fixed = 2 * field_bytes
                by_mem = int((budget - fixed) // (lane_tight * field_bytes * ntols))
            if by_mem < 1:
                need = fixed + lane_tight * field_bytes * ntols
                raise SizingRefused(
                    "bench.py: this rank's share of the host memory cannot hold the workload: %.1f GiB available / %d ranks on the node x 0.8 = %.1f GiB, "
                    "but the smallest configuration -- input field + one output field (2 x %.1f GiB) + one lane per tolerance (%d x 0.25 x %.1f GiB of "
                    "coded streams) -- needs %.1f GiB.  Use a smaller --size, fewer --tols, or a node with more memory per GPU."
                    % (mem / 2 ** 30, mem_share, budget / 2 ** 30, field_bytes / 2 ** 30, ntols, field_bytes / 2 ** 30, need / 2 ** 30))
    # (planes_per_field: 1 byte per element and plane, encoder and decoder context of a lane each hold a field's planes)
    # an encoder's planes drain as its coder advances (half of them are gone on average), a decoder's stay until its field is
    # done: 0.75 of the two contexts' worst case; a call that finds no room for a plane waits for chunks to come back
    by_hbm = int((0.92 * hbm_free - nslots * 2.2 * field_bytes) // ((hbm_per_lane * planes_per_field / 4.0 if host_mode else 3.0) * field_bytes * ntols))
    if by_hbm < 1:
        raise SizingRefused("bench.py: %.1f GiB of free HBM cannot hold %d work-space slots (2.2 x %.1f GiB each) and one lane per tolerance"
                            % (hbm_free / 2 ** 30, nslots, field_bytes / 2 ** 30))
    jobs = min(want, by_cpu, by_mem, by_hbm)
    return jobs, {"jobs_requested": want, "cpus_per_rank": round(cpus, 1), "host_mem_per_rank_gib": round(mem / mem_share / 2 ** 30, 1) if mem else None,
                  "hbm_free_gib": round(hbm_free / 2 ** 30, 1), "jobs_by_cpu": by_cpu, "jobs_by_host_mem": by_mem, "jobs_by_hbm": by_hbm,
                  "host_memory_regime": regime, "out_buffers": out_pool if host_mode else 0,
                  "host_pages_of_consumed_streams_dropped": trim}


def fit_jobs_or_smaller(size, *args, **kw):
    """fit_jobs at `size`; if the rank's memory refuses a 1024^3 workload, once more at 512^3 (BASELINE configs[3]'s field
    size) -- returns (size actually sized for, jobs, sizing); the sizing says so (`fell_back_from_size`).  Anything that does
    not fit 512^3 either is refused."""
    def at(n):
        a = list(args)
        a[2] = n ** 3 * 8  # field_bytes
        return fit_jobs(*a, **kw)
    try:
        jobs, info = at(size)
        return size, jobs, info
    except SizingRefused as first:
        if size <= 512:
            raise
        try:
            jobs, info = at(512)
        except SizingRefused:
            raise first
        info["fell_back_from_size"] = size
        info["refusal_at_that_size"] = str(first)
        return 512, jobs, info


def cpu_model():
    try:
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(size, tols, ncores):
    """The reference itself (oracle/_ref, kind "reference") or, if absent, the oracle port, on a bounded
    sample of the same workload: one thread (the reference is single-threaded), then one independent field
    per core on all cores of this rank (the "all host cores" figure)."""
    import threading
    import numpy as np
    from oracle import loader
    from waverange_amd import synth
    if loader.have_ref():
        impl, kind = loader.Reference(), "reference"
    else:
        impl, kind = loader.Oracle(), "port"
    f = synth.field(size, size, size, seed=12345)

    def roundtrip(fld):
        for tol in tols:
            enc = impl.encode(fld, tol)
            impl.decode(enc, fld.shape)

    fd = os.dup(1)
    devnull = os.open(os.devnull, os.O_WRONLY)
    sys.stdout.flush()
    os.dup2(devnull, 1)  # the reference prints progress lines from C++
    try:
        t0 = time.time()
        roundtrip(f)
        dt = time.time() - t0
        # all cores: one field per core, every thread inside the C library (ctypes drops the GIL); a smaller
        # field keeps this leg at the single-thread leg's duration
        size_all = max(64, int(size * 0.8) // 16 * 16)
        fa = synth.field(size_all, size_all, size_all, seed=12345)
        ths = [threading.Thread(target=roundtrip, args=(fa,)) for _ in range(ncores)]
        t1 = time.time()
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        dt_all = time.time() - t1
    finally:
        sys.stdout.flush()
        os.dup2(fd, 1)
        os.close(devnull)
        os.close(fd)
    mb = len(tols) * f.nbytes / 1e6
    mb_all = ncores * len(tols) * fa.nbytes / 1e6
    return {"value": round(mb / dt, 2), "unit": "MB/s", "cores": 1, "kind": kind, "cpu": cpu_model(),
            "sample": "%d^3 fp64 field of the same generator, tols %s, encode+decode, %.1f s"
                      % (size, ",".join("%g" % t for t in tols), dt),
            "all_cores": {"value": round(mb_all / dt_all, 2), "unit": "MB/s", "cores": ncores,
                          "sample": "one %d^3 field per core on %d cores at once, same tols, %.1f s" % (size_all, ncores, dt_all)},
            "note": "bounded samples, smaller than the workload's 1024^3: the reference's rate falls with the field size (BASELINE.md: 440 -> 233 MB/s forward "
                    "from 256^3 to 512^3), so both figures are upper bounds for what it does at size"}


def launch_ranks(nranks, argv, dry):
    """`python bench.py --gpus N` without a launcher around it: this process becomes the parent of N ranks.
    It starts them (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, the contract of
    torch.distributed.run) BEFORE anything here has touched the GPU -- no HIP call, no torch.cuda call that
    initialises the device, no libwaverange_amd load -- relays what they print (rank 0 prints the JSON line) and
    exits with the first non-zero status.  It never re-execs itself.  The analogue in the reference is one process
    per subdomain started by a shell loop (examples/mssg/divided/all_enc_dec.sh:7-11)."""
    import socket
    import subprocess
    backend = os.environ.get("WR_BENCH_BACKEND", "nccl")
    if not dry and backend == "nccl":
        import torch  # device_count() reads the topology, it does not initialise a device
        ndev = torch.cuda.device_count()
        if ndev < nranks:
            raise SystemExit("bench.py: --gpus %d but %d GPU(s) visible; one rank per GPU (WR_BENCH_BACKEND=gloo rehearses "
                             "more ranks than GPUs, the ranks then share devices)" % (nranks, ndev))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(nranks):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(nranks), LOCAL_WORLD_SIZE=str(nranks),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), WR_BENCH_CHILD="1")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env))
    status = 0
    try:
        for p in procs:
            rc = p.wait()
            if rc and not status:
                status = rc
                for q in procs:  # a rank that died would leave the others in the barrier for good
                    if q.poll() is None:
                        q.terminate()
    finally:
        for q in procs:
            if q.poll() is None:
                q.kill()
    sys.exit(status)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--dry-launch", action="store_true", help="ranks print RANK / WORLD_SIZE / the device they would take and exit "
                    "(checks the launcher without a GPU)")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=0, help="fields a step codes (0: 6 per tolerance setting); the lanes pull fields from the run's "
                    "queue, so up to --jobs x tols fields are in flight whatever the batch and steps overlap")
    ap.add_argument("--size", type=int, default=1024)
    ap.add_argument("--tols", type=str, default="1e-3,1e-7")
    ap.add_argument("--cpu-size", type=int, default=448)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--jobs", type=int, default=20, help="fields in flight per tolerance (lanes = jobs x tols); cut down to what the rank's CPUs, host memory and HBM allow")
    ap.add_argument("--fields-per-cpu", type=float, default=2.5, help="fields in flight per CPU of the rank that the coder pool is given (the upper bound the CPUs put on --jobs)")
    ap.add_argument("--hbm-per-lane", type=float, default=0.6, help="plane memory a lane is budgeted in HBM, in field sizes at four planes per field (measured: 40 lanes "
                    "hold 194-196 GiB of planes at 1024^3 since decodes wait for admission to the coder pool WITHOUT their planes: 0.61)")
    ap.add_argument("--trim-host", action="store_true", help="hand the pages of consumed coded streams back to the system (done by itself when host memory is what limits the lanes)")
    ap.add_argument("--timeline", type=str, default=None, help="write what the coder pool did every half second of the timed region to this file (fill and drain of a run)")
    ap.add_argument("--out-buffers", type=int, default=4, help="pinned output fields shared by all lanes (host mode: a decode needs one only for its last ~0.25 s)")
    ap.add_argument("--threads", type=int, default=1, help="range-coder threads per decode call; planes are interleaved when fewer than planes")
    ap.add_argument("--enc-threads", type=int, default=0, help="range-coder threads per encode call (0: as --threads; 2 was measured: no gain once the cores are full)")
    ap.add_argument("--resident", action="store_true", help="fields start and end in HBM (round-1 measurement) instead of host buffers")
    ap.add_argument("--slots", type=int, default=0, help="device work-space slots (0: library default)")
    ap.add_argument("--pool", type=int, default=-1, help="coder-pool worker threads shared by all fields in flight (-1: one per CPU of this rank; 0: no pool, "
                    "every call runs its own --threads coder threads)")
    ap.add_argument("--dec-streams", type=int, default=4, help="plane streams a pool worker's decoder loop interleaves (1..4)")
    ap.add_argument("--no-warm-transforms", action="store_true", help="skip the back-to-back transforms behind the timed region (roofline_warm): a profiled run "
                    "then holds only the pipeline's launches of k_inv_fused (tools/final_profiles.sh)")
    ap.add_argument("--secondary-steps", type=int, default=2, help="steps of the secondary pass at --secondary-tol after the timed region (N = 1 only; 0: none)")
    ap.add_argument("--secondary-tol", type=float, default=1e-16, help="BASELINE configs[4]'s near-lossless tolerance: 8 planes per field, mostly noise")
    ap.add_argument("--secondary-lanes", type=int, default=8, help="fields in flight in the secondary pass (each holds up to 2 x 6 GB of coded bytes at 1024^3)")
    args = ap.parse_args()
    tols = [float(t) for t in args.tols.split(",")]
    n = args.size
    host_mode = not args.resident
    # Before anything touches the GPU or loads a library: (1) this pool's host driver only supports dmabuf IPC -- without
    # HSA_ENABLE_IPC_MODE_LEGACY=0 RCCL's set-up between two ranks fails with `hipIpcGetMemHandle: invalid argument`; the
    # variable is read when the ROCm runtime initialises, so it must be in the environment of EVERY rank, whoever started it
    # (launch_ranks sets it for its children, an external launcher such as torch.distributed.run does not);  (2) the library's
    # launch ring (a SIGABRT handler that names the last plane-kernel launches, WR_FAULT_LOG) costs nothing and is the only
    # witness if a GPU fault ends a rank inside the runtime on a node nobody can log into.
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ.setdefault("WR_FAULT_LOG", "1")

    # N ranks: under a launcher (the driver's torch.distributed.run) WORLD_SIZE is set and this process is one of
    # them; started plainly with --gpus N > 1 this process is their parent (nothing below may run in it)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ and "RANK" not in os.environ:
        launch_ranks(args.gpus, sys.argv[1:], args.dry_launch)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        if args.gpus > 1:
            raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
        if rank == 0:
            print("bench.py: --gpus not given, following the launcher's WORLD_SIZE=%d" % world, file=sys.stderr)
    if args.dry_launch:
        # what this rank would take of the host if the node had one GPU per rank (no GPU is touched: the node's GPUs' NUMA
        # nodes are not looked up, the share is the rank's slice of whole cores; HBM assumed empty): the rehearsal of an
        # N-rank run's sizing on a box with fewer GPUs
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        sizing = None
        try:
            mine = cpu_share_of(local_rank, local_world, read_cpu_topology(()))
            if mine:
                os.sched_setaffinity(0, mine)
            hm = int(os.environ["WR_BENCH_TEST_HOST_MEM_GIB"]) << 30 if os.environ.get("WR_BENCH_TEST_HOST_MEM_GIB") else None  # (tests: a node's memory)
            size_for, jobs, sizing = fit_jobs_or_smaller(n, args.jobs, len(tols), n ** 3 * 8, int(0.97 * 288e9), len(mine) if mine else None, host_mode,
                                                         pooled=args.pool != 0, out_pool=args.out_buffers, gpus_on_node=local_world, nslots=args.slots or 3,
                                                         planes_per_field=8 if min(tols) < 1e-12 else 4, fields_per_cpu=args.fields_per_cpu, host_mem=hm)
            sizing.update(size=size_for, lanes=jobs * len(tols), cpu_affinity_share=len(mine) if mine else None,
                          cpus=",".join("%d" % c for c in sorted(os.sched_getaffinity(0))) if len(os.sched_getaffinity(0)) <= 64 else "%d CPUs" % len(os.sched_getaffinity(0)))
        except SizingRefused as e:
            print(json.dumps({"dry_launch": True, "rank": rank, "world_size": world, "sizing": {"refused": str(e)}}), flush=True)
            raise
        except (OSError, ValueError) as e:
            sizing = {"error": str(e)}
        print(json.dumps({"dry_launch": True, "rank": rank, "local_rank": local_rank, "world_size": world,
                          "master": "%s:%s" % (os.environ.get("MASTER_ADDR"), os.environ.get("MASTER_PORT")),
                          "launched_by": "bench.py" if os.environ.get("WR_BENCH_CHILD") else "external launcher" if world > 1 else "none",
                          "environment": {k: os.environ.get(k) for k in ("HSA_ENABLE_IPC_MODE_LEGACY", "WR_FAULT_LOG")},
                          "sizing": sizing}), flush=True)
        return

    import torch
    dist = None
    # WR_BENCH_BACKEND=gloo is a rehearsal mode for boxes with fewer GPUs than ranks (ranks then
    # share devices and the barrier runs on CPU tensors); the real multi-GPU run uses RCCL.
    backend = os.environ.get("WR_BENCH_BACKEND", "nccl")
    ndev = torch.cuda.device_count()
    if ndev < 1:
        raise SystemExit("bench.py: no GPU visible (libwaverange_amd has no CPU fallback)")
    dev_index = local_rank if backend == "nccl" else local_rank % ndev
    # Under a launcher (RANK / WORLD_SIZE / MASTER_* in the environment: torch.distributed.run, or launch_ranks above) the
    # process group is set up whatever the number of ranks -- with WORLD_SIZE=1 too, so that the one-GPU run the driver
    # launches through torch.distributed.run takes the same RCCL barrier and max-reduction as the 8-GPU one.
    launched = all(k in os.environ for k in ("RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"))
    if world > 1 and not launched:
        raise SystemExit("bench.py: WORLD_SIZE=%d without RANK / MASTER_ADDR / MASTER_PORT" % world)
    if launched:
        import torch.distributed as dist
        torch.cuda.set_device(dev_index)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(backend)
    red_dev = "cuda" if backend == "nccl" else "cpu"
    barrier_kind = "none (one process, no launcher)" if dist is None else "rccl" if backend == "nccl" else backend
    import numpy as np
    from waverange_amd import api
    api.set_verbosity(0)
    api.set_threads(args.threads, args.enc_threads)
    if api.device_count() < 1:
        raise SystemExit("bench.py: no GPU visible (libwaverange_amd has no CPU fallback)")
    if args.slots:
        api.set_device_slots(dev_index, args.slots)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            t = torch.zeros(1, device=red_dev)
            dist.all_reduce(t)
            torch.cuda.synchronize()

    # How many fields in flight this rank can afford (fit_jobs): its share of the CPUs, of the host memory (coded
    # streams) and the free HBM (the quantized planes of the fields in flight live there).  --jobs is the upper bound.
    share = None
    gpus_on_node = ndev
    if backend == "nccl":
        bdfs = []
        try:
            for d in range(ndev):
                pr = torch.cuda.get_device_properties(d)
                bdfs.append("%04x:%02x:%02x.0" % (pr.pci_domain_id, pr.pci_bus_id, pr.pci_device_id))
        except (AttributeError, RuntimeError):
            bdfs = []  # (no PCI addresses: the shares are slices of all cores instead of the GPUs' own NUMA nodes)
        share = take_cpu_share(local_rank, ndev, bdfs)
    elif world > ndev:
        # rehearsal (gloo, ranks share devices): the host is split as if the node had one GPU per rank, so that the
        # ranks' CPU shares, lanes and memory shares are the ones of the real run
        gpus_on_node = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        share = take_cpu_share(local_rank, gpus_on_node, ())
    # (SizingRefused -- a rank whose share of the host memory cannot hold even the smallest configuration -- ends the rank with a
    # non-zero status and the arithmetic on stderr before it has allocated anything; under a launcher the other ranks follow)
    n, jobs, limits = fit_jobs_or_smaller(n, args.jobs, len(tols), n ** 3 * 8, torch.cuda.mem_get_info(dev_index)[0], share, host_mode,
                                          pooled=args.pool != 0, out_pool=args.out_buffers, gpus_on_node=gpus_on_node, nslots=args.slots or 3,
                                          planes_per_field=8 if min(tols) < 1e-12 else 4, fields_per_cpu=args.fields_per_cpu, hbm_per_lane=args.hbm_per_lane)
    if limits.get("fell_back_from_size") and rank == 0:
        print("bench.py: --size %d does not fit this rank's host memory, running --size %d instead: %s"
              % (limits["fell_back_from_size"], n, limits["refusal_at_that_size"]), file=sys.stderr)
    if host_mode:
        args.out_buffers = limits["out_buffers"]
    limits["cpu_affinity_share"] = share

    def cpu_ranges(cpus):
        out, cpus = [], sorted(cpus)
        for c in cpus:
            if out and c == out[-1][1] + 1:
                out[-1][1] = c
            else:
                out.append([c, c])
        return ",".join("%d" % a if a == b else "%d-%d" % (a, b) for a, b in out)

    limits["cpus"] = cpu_ranges(os.sched_getaffinity(0))
    rank_sizing = None
    if dist is not None and world > 1:  # what every rank found for itself: disjoint CPU shares, equal lanes (weak scaling)
        rank_sizing = [None] * world
        dist.all_gather_object(rank_sizing, dict(limits, rank=rank, device=dev_index))
    trim_host = bool(limits.get("host_pages_of_consumed_streams_dropped")) or args.trim_host
    pool_workers = max(1, int(limits["cpus_per_rank"])) if args.pool < 0 else args.pool
    if pool_workers:
        # (every worker pinned to a physical core of its own -- the idlest of the box, the GPU's NUMA node first -- was measured:
        # 14.87 / 14.81 against 14.93 / 14.83 GB/s, same box, K = 8: profiles/r05/h_ab_pool_workers_pinned_k8_same_box.txt.  Where the
        # scheduler puts the workers is not what makes the boxes differ.)
        api.set_coder_pool(pool_workers, args.dec_streams)

    # One lane per field of the batch (jobs x tolerance settings: independent jobs that run concurrently on
    # the one GPU; they all code this rank's synthetic field).  A lane is a two-stage pipeline -- encoder
    # context and decoder context, two coded-stream buffers in between -- so that step k+1's encode overlaps
    # step k's decode: device stages of all contexts are scheduled inside the library, host range coding overlaps.
    import threading
    ctx = api.Context(dev_index)
    shape = (n, n, n)
    nelem = n ** 3
    orig = ctx.alloc(nelem * 8)
    ctx.synth_field(orig, n, n, n, 12345 + rank)
    ctx.sync()
    _, cap = api.setup_wr(n, n, n)

    def host_field():
        try:
            return api.pinned_array(shape)
        except api.WaveRangeError:  # no pinned memory left: pageable works too (staged by the runtime)
            return np.empty(shape, dtype=np.float64)

    h_in = None
    if host_mode:
        h_in = host_field()
        api._check(api.lib().wr_dev_download(ctx.h, h_in.ctypes.data, orig.ptr, nelem * 8))
        orig.free()
        orig = None
    lanes = []
    batch = args.batch if args.batch > 0 else 6 * len(tols)
    batch = max(len(tols), batch // len(tols) * len(tols))  # every step codes the same number of fields per tolerance
    for i in range(len(tols) * jobs):
        ce = ctx if i == 0 else api.Context(dev_index)
        cd = api.Context(dev_index)
        ln = dict(enc=ce, dec=cd)
        if not host_mode:
            ln["work"], ln["rec"] = ce.alloc(nelem * 8), cd.alloc(nelem * 8)
        lanes.append(ln)

    # hand-over buffers for the coded streams (HandoverBuffers above): at most two per lane in flight at a time
    bufs = HandoverBuffers(lambda: np.empty(cap, dtype=np.uint8), 2 * len(tols) * len(lanes) + 4)
    take_buf, give_buf, all_bufs = bufs.take, bufs.give, bufs.all

    # host mode: the reconstructions land in one of a few pinned output fields, borrowed for the device half of a
    # decode only (wr_decode_begin / wr_decode_finish_host); the last reconstruction of the last lane is checked
    # against the input before its buffer goes back
    import queue
    out_pool = queue.Queue()
    if host_mode:
        for _ in range(max(1, min(args.out_buffers, len(lanes)))):
            out_pool.put(host_field())
    accuracy = {}

    def linf_vs_input(out):
        diff = amax = 0.0
        for z in range(0, n, 64):  # in slabs: no field-sized temporaries
            a, b = h_in[z:z + 64], out[z:z + 64]
            diff = max(diff, float(np.abs(a - b).max()))
            amax = max(amax, float(np.abs(a).max()))
        return diff / amax

    main_capture = {"coded": {}, "decoded": []}
    import ctypes
    libc = ctypes.CDLL(None, use_errno=True)
    libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]

    def drop_pages(buf, used):
        """Hands the pages of a consumed coded stream back to the system (MADV_DONTNEED: they read as zeros and are taken
        again when the buffer is written next).  A hand-over buffer would otherwise keep the 2 GB of the largest field it
        has ever held resident: 4 GiB per lane, two thirds of what a rank holds in host memory."""
        a = (buf.ctypes.data + 4095) & ~4095
        e = (buf.ctypes.data + int(used)) & ~4095
        if e - a >= (64 << 20):
            libc.madvise(a, e - a, 4)

    stats = {t: {} for t in tols}
    keys = ("fwd_ms", "inv_ms", "quant_ms", "dequant_ms", "minmax_ms", "enc_s", "dec_s", "enc_rc_s", "dec_rc_s", "enc_gpu_s",
            "dec_gpu_s", "enc_wait_s", "dec_wait_s", "enc_h2d_ms", "enc_d2h_ms", "dec_h2d_ms", "dec_d2h_ms", "nlay")
    acc = {k: [] for k in keys}
    lock = threading.Lock()
    errors = []

    def run_steps(nsteps, record, tols=tols, lanes=lanes, batch=batch, trim_host=trim_host, capture=None):
        """nsteps steps = nsteps x batch fields (field i at tolerance tols[i % len(tols)]), pulled from one queue by the
        lanes; returns when all of them have been encoded AND decoded.  (The secondary pass runs other tolerances on
        fewer lanes: the keyword arguments.)  capture: {"coded": {}, "decoded": []} -- the coded bytes of the last field of every
        tolerance and the last reconstruction are held back there (their buffers stay out of circulation) for the parity
        check after the run."""
        total = nsteps * batch
        nxt = [0]

        def next_field():
            with lock:
                i = nxt[0]
                if i >= total:
                    return None
                nxt[0] = i + 1
                return i

        ths = []
        for ln in lanes:
            coded = [threading.Semaphore(0), threading.Semaphore(0)]  # slot holds a coded field (or the end mark)
            free = [threading.Semaphore(1), threading.Semaphore(1)]   # slot may be overwritten
            box = [None, None]

            def encoder(ln=ln, coded=coded, free=free, box=box):
                k = 0
                try:
                    while True:
                        i = next_field()
                        free[k & 1].acquire()
                        if i is None or errors:
                            box[k & 1] = None        # end mark for this lane's decoder
                            coded[k & 1].release()
                            return
                        tol = tols[i % len(tols)]
                        buf = take_buf(tol)
                        if host_mode:
                            enc, te = ln["enc"].encode_host(h_in, tol, out=buf)
                        else:
                            ln["enc"].copy(ln["work"], orig, nelem * 8)
                            enc, te = ln["enc"].encode(ln["work"], shape, tol, out=buf)
                        box[k & 1] = (enc, te, tol, i, buf)
                        if capture is not None and i >= total - len(tols):
                            capture["coded"][tol] = (enc, buf)   # its buffer is not written again: no field follows on any lane
                        coded[k & 1].release()
                        k += 1
                except Exception as exc:
                    errors.append(exc)
                    box[0] = box[1] = None
                    for sem in coded:
                        sem.release()

            def decoder(ln=ln, coded=coded, free=free, box=box):
                k = 0
                try:
                    while True:
                        coded[k & 1].acquire()
                        item = box[k & 1]
                        if item is None or errors:
                            return
                        enc, te, tol, i, buf = item
                        # (the last field of every tolerance keeps its buffer out of circulation: parity is taken from it later)
                        held_back = capture is not None and i >= total - len(tols)
                        if host_mode:
                            ln["dec"].decode_begin(shape, enc)       # host range decoding: seconds, no field buffer
                            if trim_host and not held_back:
                                drop_pages(buf, enc["ntot_enc"])  # the coded stream is not needed any more: its pages go back
                            if not held_back:
                                give_buf(tol, buf)
                            free[k & 1].release()
                            out = out_pool.get()
                            keep = False
                            try:
                                td = ln["dec"].decode_finish_host(out)  # kernels, download: ~0.25 s
                                if capture is not None and i == total - 1:
                                    capture["decoded"].append((tol, out))  # held back: accuracy and parity are taken after the timed region
                                    keep = True
                            finally:
                                if not keep:
                                    out_pool.put(out)
                        else:
                            td = ln["dec"].decode(ln["rec"], shape, enc)
                            if record and i == total - 1:
                                diff, amax = ln["dec"].linf(orig, ln["rec"], nelem)
                                accuracy["linf_rel"] = diff / amax
                            if not held_back:
                                give_buf(tol, buf)
                            free[k & 1].release()
                        k += 1
                        if record:
                            with lock:
                                for key, val in (("fwd_ms", te["transform_ms"]), ("inv_ms", td["transform_ms"]), ("quant_ms", te["quant_ms"]),
                                                 ("dequant_ms", td["quant_ms"]), ("minmax_ms", te["minmax_ms"]), ("enc_s", te["total"]),
                                                 ("dec_s", td["total"]), ("enc_rc_s", te["rangecoder"]), ("dec_rc_s", td["rangecoder"]),
                                                 ("enc_gpu_s", te["gpu"]), ("dec_gpu_s", td["gpu"]), ("enc_wait_s", te["wait"]),
                                                 ("dec_wait_s", td["wait"]), ("enc_h2d_ms", te["h2d_ms"]), ("enc_d2h_ms", te["d2h_ms"]),
                                                 ("dec_h2d_ms", td["h2d_ms"]), ("dec_d2h_ms", td["d2h_ms"]), ("nlay", enc["nlay"])):
                                    acc[key].append(val)
                                stats[tol] = {"nlay": enc["nlay"], "ntot_enc": enc["ntot_enc"],
                                              "bits_per_symbol": [round(8.0 * v / nelem, 3) for v in enc["len_enc_vec"]]}
                except Exception as exc:
                    errors.append(exc)
                    for sem in free:
                        sem.release()

            ths += [threading.Thread(target=encoder), threading.Thread(target=decoder)]
        for th in ths:
            th.start()
        for th in ths:
            th.join()
        if errors:
            raise errors[0]

    if args.warmup:
        run_steps(args.warmup, False)
    barrier()
    def throttled_s():  # time this cgroup's threads were runnable but held back by the CPU quota (cgroup v2)
        try:
            with open("/sys/fs/cgroup/cpu.stat") as fh:
                for line in fh:
                    if line.startswith("throttled_usec"):
                        return int(line.split()[1]) * 1e-6
        except OSError:
            pass
        return None
    def worker_cpu_s():  # CPU seconds of the coder pool's workers so far (threads named wr-coder-<id>, wr_rangecoder.cpp)
        tick, tot = os.sysconf("SC_CLK_TCK"), 0.0
        try:
            for tid in os.listdir("/proc/self/task"):
                try:
                    with open("/proc/self/task/%s/stat" % tid) as fh:
                        st = fh.read()
                except OSError:
                    continue
                name, rest = st[st.index("(") + 1:st.rindex(")")], st[st.rindex(")") + 2:].split()
                if name.startswith("wr-coder-"):
                    tot += (int(rest[11]) + int(rest[12])) / tick  # utime + stime
        except OSError:
            return None
        return tot
    thr0 = throttled_s()
    wcpu0 = worker_cpu_s()
    cpu0 = sum(os.times()[:2])
    idle0 = api.stat(api.STAT_POOL_IDLE_MS)
    burn0 = api.stat(api.STAT_CLOCK_WARMUP_MS)
    gate0 = api.stat(api.STAT_DECODE_GATE_MS)
    winwait0 = api.stat(api.STAT_WINDOW_WAIT_MS)
    queue0, pwait0 = api.stat(api.STAT_POOL_QUEUE_MS), api.stat(api.STAT_PLANE_WAIT_MS)
    loops0 = api.pool_loop_stats()
    sampler = stop_sampling = None
    if args.timeline and rank == 0:
        stop_sampling = threading.Event()

        def sample():
            # workers busy per loop kind and symbols per second over every half second, fields begun so far
            with open(args.timeline, "w") as fh:
                fh.write("# t_s  idle_workers  " + "  ".join("%s(workers,Gsym/s)" % k for k in api.pool_loop_stats()) + "  fields_done\n")
                tp, ip, lp = time.perf_counter(), api.stat(api.STAT_POOL_IDLE_MS), api.pool_loop_stats()
                while not stop_sampling.wait(0.5):
                    tn, inow, ln = time.perf_counter(), api.stat(api.STAT_POOL_IDLE_MS), api.pool_loop_stats()
                    w = tn - tp
                    cols = ["%5.2f,%5.2f" % ((ln[k][0] - lp[k][0]) / w, (ln[k][1] - lp[k][1]) * 6e-5 / w) for k in ln]
                    fh.write("%7.2f  %5.2f  %s  %d\n" % (tn - t0, (inow - ip) * 1e-3 / w, "  ".join(cols), len(acc["nlay"])))
                    fh.flush()
                    tp, ip, lp = tn, inow, ln
        sampler = threading.Thread(target=sample, daemon=True)
    t0 = time.perf_counter()
    if sampler:
        sampler.start()
    run_steps(args.steps, True, capture=main_capture)
    barrier()
    dt = time.perf_counter() - t0
    if sampler:
        stop_sampling.set()
        sampler.join()
    cpu_used = (sum(os.times()[:2]) - cpu0) / dt  # this rank's average number of busy CPUs over the timed region
    planes_bytes = api.stat(api.STAT_DEVICE_PLANE_BYTES)  # the plane pool after the timed region (in use + idle: it only grows during a run)
    wcpu1 = worker_cpu_s()
    workers_cpu = None if wcpu0 is None or wcpu1 is None else (wcpu1 - wcpu0) / dt  # ... of them the coder pool's workers
    pool_idle = (api.stat(api.STAT_POOL_IDLE_MS) - idle0) * 1e-3 / dt  # workers waiting for a job, on average
    window_wait = (api.stat(api.STAT_WINDOW_WAIT_MS) - winwait0) * 1e-3 / dt  # workers blocked on a plane window's DMA copy, on average
    nfields = max(1, args.steps * batch)
    queue_wait = (api.stat(api.STAT_POOL_QUEUE_MS) - queue0) * 1e-3 / nfields   # per field: its planes' waits for a pool worker, summed
    plane_wait = (api.stat(api.STAT_PLANE_WAIT_MS) - pwait0) * 1e-3 / nfields   # per field: waits for device memory for its planes
    gate_wait = (api.stat(api.STAT_DECODE_GATE_MS) - gate0) * 1e-3 / nfields     # per field: a decode's wait for admission to the pool (no planes held yet)
    burn_ms = (api.stat(api.STAT_CLOCK_WARMUP_MS) - burn0) / float(nfields)     # per field: clock warm-up load in front of its two kernel stages (0 unless WR_CLOCK_WARMUP_MS)
    loops1 = api.pool_loop_stats()
    pool_loops = {}
    for kind in loops1:  # in-pipeline rate of every coder loop: symbols per worker-second, workers busy in it on average
        sec, blk = loops1[kind][0] - loops0[kind][0], loops1[kind][1] - loops0[kind][1]
        if sec > 0:
            pool_loops[kind] = {"Msym_per_worker_s": round(blk * 60000 / sec / 1e6, 1), "workers": round(sec / dt, 2)}
    thr1 = throttled_s()
    throttled = (thr1 - thr0) / dt if thr0 is not None and thr1 is not None else None
    if dist is not None:
        t = torch.tensor([dt], device=red_dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # accuracy of the last reconstruction (tols[-1]) against the original, and the bytes of the timed run against the
    # reference's pins -- outside the timed region
    def check_capture(cap):
        """accuracy of the held-back reconstruction and parity of the held-back bytes against this rank's pins; everything that
        was held back goes into circulation again"""
        res = {"linf_rel": None, "parity": None}
        if host_mode and cap["decoded"]:
            res["linf_rel"] = linf_vs_input(cap["decoded"][0][1])
        if host_mode:
            res["parity"] = parity_vs_pins(n, 12345 + rank, {t: e for t, (e, _) in cap["coded"].items()}, cap["decoded"][0] if cap["decoded"] else None)
        for _, out_field in cap["decoded"]:
            out_pool.put(out_field)
        for t, (_, buf) in cap["coded"].items():
            give_buf(t, buf)
        cap["coded"].clear()
        cap["decoded"][:] = []
        return res

    # EVERY rank checks what it coded in the timed region against the reference's pins for its own field (seed 12345 + rank)
    main_check = check_capture(main_capture)
    parity = main_check["parity"]
    if main_check["linf_rel"] is not None:
        accuracy["linf_rel"] = main_check["linf_rel"]
    linf_rel = accuracy.get("linf_rel")
    if dist is not None and world > 1:
        every = [None] * world
        dist.all_gather_object(every, {"rank": rank, "parity": parity, "linf_rel": linf_rel})
        if rank == 0 and parity is not None:
            def ok(p):  # everything that could be checked was right, and something was checked
                return bool(p and p["checked"] and p["coded_sha_ok"] is not False and p["decoded_sha_ok"] is not False
                            and (p["coded_sha_ok"] or p["decoded_sha_ok"]))
            parity = dict(parity, ranks_ok=sum(1 for e in every if ok(e["parity"])),
                          ranks_without_pins=[e["rank"] for e in every if not (e["parity"] and e["parity"]["checked"])],
                          ranks_failed=[e["rank"] for e in every if e["parity"] and (e["parity"]["coded_sha_ok"] is False or e["parity"]["decoded_sha_ok"] is False)],
                          linf_rel_max=max([e["linf_rel"] for e in every if e["linf_rel"] is not None] or [None]))
    elif parity is not None:
        parity = dict(parity, ranks_ok=1 if (parity["checked"] and parity["coded_sha_ok"] is not False and parity["decoded_sha_ok"] is not False) else 0)

    mean = lambda v: float(sum(v) / max(1, len(v)))  # noqa: E731
    if rank == 0:
        field_mb = nelem * 8 / 1e6
        total_mb = world * args.steps * batch * field_mb
        alg_bytes = 16.0 * box_elems(n)             # per direction (18.28 B/elem at 2^k sizes)
        fwd_ms, inv_ms = mean(acc["fwd_ms"]), mean(acc["inv_ms"])
        t_ms = 0.5 * (fwd_ms + inv_ms)
        achieved = alg_bytes / (t_ms * 1e-3) / 1e9
        traffic, traffic_src, traffic_stale = measured_traffic(n)

        def group(alg, ms):  # roofline of a kernel group from its algorithmic bytes and its HIP-event time
            gbs = alg / (ms * 1e-3) / 1e9 if ms > 0 else 0.0
            return {"algorithmic_bytes": alg, "ms": round(ms, 3), "achieved": round(gbs, 1), "frac": round(gbs / HBM_PEAK_GBS, 4)}

        L = mean(acc["nlay"])  # planes per field, mean over the batch
        nbytes_field = nelem * 8.0
        out = {
            "metric": "encode+decode MB/s on %d^3 fp64, %s (whole job = n_gpus x per-GPU rate; host range coder included)"
                      % (n, "host buffer to host buffer" if host_mode else "field resident in HBM"),
            "value": round(total_mb / dt, 2), "unit": "MB/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 2), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "per_gpu_MBps": round(total_mb / dt / world, 2),
            "config": {"workload": "%s, tol=%s (%s)"
                                   % ("single %d^3 fp64 field per GPU" % n if world == 1 else "%d independent %d^3 fp64 fields, one per GPU" % (world, n),
                                      " and ".join("%g" % t for t in tols),
                                      "BASELINE configs[2]" if n == 1024 else
                                      ("BASELINE configs[3]: independent 512^3 fields sharded one per GPU" if world > 1 else "BASELINE configs[1]") if n == 512
                                      else "parity-size run"),
                       "boundary": "host buffers (pinned), wr_encode_host / wr_decode_begin + wr_decode_finish_host (%d output fields shared by the lanes)" % out_pool.qsize() if host_mode else "device buffers, wr_encode_device / wr_decode_device",
                       "field_shards": world,
                       "barrier": barrier_kind,
                       "multi_gpu": ("weak scaling: every rank round-trips its own stream of %d^3 fields (seed 12345 + rank) on its own GPU and its share of the "
                                     "host cores, no data-path collective%s" % (n, "" if n == 512 else "; BASELINE configs[3] (NF = 8 x 512^3, one per GPU) is this with --size 512"))
                       if world > 1 else None,
                       "range_coder": ({"pool_workers": pool_workers, "decoder_streams_per_loop": args.dec_streams, "encoder_streams_per_loop": 3} if pool_workers
                                       else {"threads_per_call": {"encode": args.enc_threads or args.threads, "decode": args.threads}}),
                       "concurrent_jobs_per_gpu": len(lanes), "fields_per_step_per_gpu": batch, "sizing": limits, "sizing_of_every_rank": rank_sizing,
                       "pipeline": "a step is a batch of %d fields (%d per tolerance); the %d lanes (encoder + decoder context each) pull fields from the run's queue, "
                                   "so steps overlap: a lane encodes its next field while it decodes the previous one" % (batch, batch // len(tols), len(lanes)),
                       "planes": {("%g" % t): stats[t] for t in tols}},
            # the figure of the library AS SHIPPED: inside the pipeline, every kernel stage starting on a GPU that has been idle
            # (no clock warm-up: WR_CLOCK_WARMUP_MS is a measurement hook that is off by default; if this run set it, `condition`
            # says so).  Beside it: roofline_warm, the same kernels back to back with the shader clock up.
            "roofline": {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src, "traffic_stale": traffic_stale,
                         "kernel": "3-D CDF-9/7 transform, 4 levels (mean of forward and inverse), all launches",
                         "algorithmic_bytes": alg_bytes, "fwd_ms": round(fwd_ms, 3), "inv_ms": round(inv_ms, 3),
                         "condition": ("in the pipeline, shipped default: no clock warm-up, a kernel stage starts on an idle GPU" if burn_ms == 0 else
                                       "in the pipeline behind WR_CLOCK_WARMUP_MS=%s of load (NOT the shipped default)" % os.environ.get("WR_CLOCK_WARMUP_MS")),
                         "rocprof_kernel_stats": committed_kernel_stats()},
            # the other kernel groups: algorithmic bytes per element L + 8 (dequantise-accumulate), 8 per min/max
            # pass (two passes per encode); times from HIP events in the run.  Quantizer: SURVEY.md 8d counts the reference's
            # formulation, 17 L - 8 (the residual array read and rewritten by every plane but the last); the kernels here cut
            # every plane from a residual recomputed from the coefficient array (k_quant_blk), so what a plane MUST move is 8
            # bytes read + 1 written: 9 L, and that is what the fraction is priced on (block histograms ride along)
            "roofline_groups": {"forward_transform": group(alg_bytes, fwd_ms), "inverse_transform": group(alg_bytes, inv_ms),
                                "quantizer": dict(group(9.0 * L * nelem, mean(acc["quant_ms"])),
                                                  in_place_formulation_bytes=(17.0 * L - 8.0) * nelem,
                                                  note="planes cut from residuals recomputed from the coefficients: 9 B per element and plane instead of the "
                                                       "in-place formulation's 17 (SURVEY.md 8d); block histograms included"),
                                "dequantizer": group((L + 8.0) * nelem, mean(acc["dequant_ms"])),
                                "minmax_x2": group(2 * nbytes_field, mean(acc["minmax_ms"])) if mean(acc["minmax_ms"]) > 1e-3
                                else {"fused_into": "forward_transform", "note": "both reductions ride on the forward kernels' loads and stores; their 16 B/elem are not counted in its algorithmic bytes"},
                                "mean_planes": round(L, 2)},
            "stages": {"encode_s": round(mean(acc["enc_s"]), 3), "decode_s": round(mean(acc["dec_s"]), 3),
                       "encode_gpu_s": round(mean(acc["enc_gpu_s"]), 4), "decode_gpu_s": round(mean(acc["dec_gpu_s"]), 4),
                       "encode_slot_wait_s": round(mean(acc["enc_wait_s"]), 4), "decode_slot_wait_s": round(mean(acc["dec_wait_s"]), 4),
                       "encode_rangecoder_s": round(mean(acc["enc_rc_s"]), 3),
                       "decode_rangecoder_s": round(mean(acc["dec_rc_s"]), 3),
                       "quant_ms": round(mean(acc["quant_ms"]), 3), "dequant_ms": round(mean(acc["dequant_ms"]), 3),
                       "minmax_ms": round(mean(acc["minmax_ms"]), 3),
                       "pcie": {"encode_field_h2d_ms": round(mean(acc["enc_h2d_ms"]), 2), "encode_planes_d2h_span_ms": round(mean(acc["enc_d2h_ms"]), 2),
                                "decode_planes_h2d_ms": round(mean(acc["dec_h2d_ms"]), 2), "decode_field_d2h_ms": round(mean(acc["dec_d2h_ms"]), 2),
                                "field_h2d_GBps": round(nbytes_field / 1e6 / max(1e-9, mean(acc["enc_h2d_ms"])), 1) if host_mode else None,
                                "field_d2h_GBps": round(nbytes_field / 1e6 / max(1e-9, mean(acc["dec_d2h_ms"])), 1) if host_mode else None},
                       "burn_ms_per_field": round(burn_ms, 2),
                       "device_only_MBps": round(2 * field_mb / max(1e-9, mean(acc["enc_gpu_s"]) + mean(acc["dec_gpu_s"])), 1)},
            "accuracy": {"tol": tols[-1], "linf_rel": linf_rel},
            "parity": parity,
        }
        # the same transform kernels back to back on a busy GPU (outside the timed region): what they do with the
        # shader clock up -- inside the pipeline every kernel stage starts on a GPU that has been idle
        try:
            if args.no_warm_transforms:
                raise RuntimeError("skipped (--no-warm-transforms)")
            ctx.trim()  # the plane pool holds the timed run's plane memory idle (most of the HBM): this leg needs a field's worth of it
            wbuf = ctx.alloc(nelem * 8)
            ctx.synth_field(wbuf, n, n, n, 12345 + rank)
            ctx.sync()
            for lvl in (4, -4):
                ctx.bench_transform(wbuf, shape, lvl, 6)   # brings the clocks up
            wf, wi = ctx.bench_transform(wbuf, shape, 4, 8), ctx.bench_transform(wbuf, shape, -4, 8)
            wbuf.free()
            wa = alg_bytes / (0.5 * (wf + wi) * 1e-3) / 1e9
            out["roofline_warm"] = {"achieved": round(wa, 1), "frac": round(wa / HBM_PEAK_GBS, 4), "fwd_ms": round(wf, 3), "inv_ms": round(wi, 3),
                                    "note": "plain forward kernels (no min/max riding along), 8 transforms back to back after 12 to warm up: the shader clock is up "
                                            "(what WR_CLOCK_WARMUP_MS buys inside the pipeline; it does not move `value`)"}
        except Exception as exc:  # noqa: BLE001
            out["roofline_warm"] = {"error": str(exc)}
        other, other_src = committed_extra("bench%d_resident.json" % n if host_mode else "bench%d_host.json" % n)
        if other:
            out["resident_fields" if host_mode else "host_buffers"] = {"value": other.get("value"), "unit": "MB/s", "source": other_src}
        try:  # peak resident host memory of this rank (pinned staging included)
            with open("/proc/self/status") as fh:
                hwm = [l for l in fh if l.startswith("VmHWM")][0].split()
            out["host_peak_rss_gib"] = round(int(hwm[1]) / 2 ** 20, 2)
        except Exception:
            pass
        out["hbm_planes_gib"] = round(planes_bytes / 2 ** 30, 1)  # device buffers of quantized planes at the end of the timed region (in use + idle)
        out["host_cpus_busy"] = round(cpu_used, 2)  # process CPU time / wall time of the timed region (this rank)
        if workers_cpu is not None:
            # ... split into the coder pool's workers and everything else (the lanes' threads, the HIP / HSA runtime's threads, Python)
            out["host_cpus_busy_by"] = {"pool_workers": round(workers_cpu, 2), "other_threads": round(cpu_used - workers_cpu, 2)}
        out["pool_workers_waiting_for_window_dma"] = round(window_wait, 2)
        out["pool_loops"] = pool_loops
        out["waits_per_field_s"] = {"planes_in_the_pool_queue_summed": round(queue_wait, 2), "device_memory_for_planes": round(plane_wait, 2),
                                    "decode_admission_gate": round(gate_wait, 2)}
        out["pool_workers_idle"] = round(pool_idle, 2)  # of the pool's workers, how many were waiting for a job on average
        if throttled is not None:
            out["cpu_quota_throttled"] = round(throttled, 3)  # cgroup cpu.stat throttled time / wall time of the timed region
        if world == 1 and args.secondary_steps > 0 and host_mode and pool_workers:
            # BASELINE configs[4]'s tolerance on the same pipeline, outside the timed region: a bounded sample (few lanes, few
            # steps: mostly fill and drain) so that the command stays within minutes; `bench.py --tols 1e-16` is the full run
            try:
                for buf in all_bufs:   # the pages of the main run's coded streams go back first: a 1e-16 stream is 6 GB at 1024^3
                    drop_pages(buf, cap)
                bufs.reset(args.secondary_tol)  # ... and every buffer is in circulation again (the parity check is done)
                avail = None
                with open("/proc/meminfo") as fh:
                    avail = [int(l.split()[1]) * 1024 for l in fh if l.startswith("MemAvailable")][0]
                for path in ("/sys/fs/cgroup/memory.max",):
                    m, cur = _cgroup_number(path), _cgroup_number("/sys/fs/cgroup/memory.current")
                    if m and cur and m[0] != "max":
                        avail = min(avail, int(m[0]) - int(cur[0]))
                nl = max(1, min(args.secondary_lanes, len(lanes), int(0.5 * avail // (2 * 0.75 * nbytes_field))))
                sb = nl  # one field per lane and step
                t2 = time.perf_counter()
                stats[args.secondary_tol] = {}
                sec_capture = {"coded": {}, "decoded": []}
                run_steps(args.secondary_steps, False, tols=[args.secondary_tol], lanes=lanes[:nl], batch=sb, trim_host=True, capture=sec_capture)
                torch.cuda.synchronize()
                dt2 = time.perf_counter() - t2
                sec_planes = {"nlay": sec_capture["coded"][args.secondary_tol][0]["nlay"], "ntot_enc": int(sec_capture["coded"][args.secondary_tol][0]["ntot_enc"])} \
                    if args.secondary_tol in sec_capture["coded"] else None
                sec_check = check_capture(sec_capture)  # the last field's bytes and reconstruction against the reference's pin at this tolerance
                out["secondary"] = {"tol_%g" % args.secondary_tol: {
                    "value": round(args.secondary_steps * sb * field_mb / dt2, 2), "unit": "MB/s", "lanes": nl, "steps": args.secondary_steps,
                    "fields": args.secondary_steps * sb, "seconds": round(dt2, 1),
                    "planes": sec_planes, "accuracy": {"tol": args.secondary_tol, "linf_rel": sec_check["linf_rel"]}, "parity": sec_check["parity"],
                    "note": "BASELINE configs[4]'s tolerance (8 planes per field, six of them noise) through the same pipeline after the timed region; "
                            "a bounded sample -- %d fields on %d lanes, fill and drain included -- not the steady-state rate (`bench.py --tols %g`)"
                            % (args.secondary_steps * sb, nl, args.secondary_tol)}}
            except Exception as exc:  # noqa: BLE001
                out["secondary"] = {"error": str(exc)}
        if world == 1 and not args.no_cpu_baseline:
            # one field alone on the idle machine, a coder thread per plane: the latency a single
            # encoding_wrap / decoding_wrap caller sees (outside the timed region)
            api.set_coder_pool(0)
            api.set_threads(8)
            try:  # (a leg beside the measurement: whatever goes wrong in it must not cost the line)
                ln = lanes[-1]
                if host_mode:
                    out1 = out_pool.get()
                    enc1, te1 = ln["enc"].encode_host(h_in, tols[-1], out=take_buf(tols[-1]))
                    td1 = ln["dec"].decode_host(out1, enc1)
                else:
                    ln["enc"].copy(ln["work"], orig, nelem * 8)
                    enc1, te1 = ln["enc"].encode(ln["work"], shape, tols[-1], out=take_buf(tols[-1]))
                    td1 = ln["dec"].decode(ln["rec"], shape, enc1)
                L1 = enc1["nlay"]
                out["single_field"] = {"tol": tols[-1], "coder_threads": "one per plane", "encode_s": round(te1["total"], 3),
                                       "decode_s": round(td1["total"], 3),
                                       "MBps": round(field_mb / (te1["total"] + td1["total"]), 1),
                                       # one encoding_wrap / decoding_wrap call is as long as its slowest plane stream: a serial recurrence per
                                       # plane (rangecod.c:217-229, 309-351) that no number of cores shortens; throughput comes from concurrent calls
                                       "plane_streams": [{"plane": l, "bits_per_symbol": round(8.0 * enc1["len_enc_vec"][l] / nelem, 3),
                                                          "encode_Msym_per_s": round(nelem / max(1e-9, te1["plane_coder_s"][l]) / 1e6, 1),
                                                          "decode_Msym_per_s": round(nelem / max(1e-9, td1["plane_coder_s"][l]) / 1e6, 1)} for l in range(L1)],
                                       "bound_by": "the slowest plane stream: encode %.2f s, decode %.2f s of the call's %.2f / %.2f s"
                                                   % (te1["rangecoder"], td1["rangecoder"], te1["total"], td1["total"])}
            except Exception as exc:  # noqa: BLE001
                out["single_field"] = {"error": str(exc)}
            api.set_threads(args.threads, args.enc_threads)
            ncores = max(1, int(limits["cpus_per_rank"]))
            try:
                out["cpu_baseline"] = cpu_baseline(args.cpu_size, tols, ncores)
                # GPU path over the CPU reference on this box (vs_baseline stays null: BASELINE.md has no published number)
                out["vs_cpu_baseline"] = {"one_core": round(out["value"] / out["cpu_baseline"]["value"], 1),
                                          "all_cores": round(out["value"] / out["cpu_baseline"]["all_cores"]["value"], 2)}
            except Exception as exc:  # noqa: BLE001  (the line is the measurement's; this leg only stands beside it)
                out["cpu_baseline"] = {"error": str(exc)}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    for ln in lanes:
        ln["dec"].close()
        ln["enc"].close()


if __name__ == "__main__":
    main()
