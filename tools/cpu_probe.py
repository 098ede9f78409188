#!/usr/bin/env python3
"""What the host side of a rank looks like: CPUs the process may run on, the cgroup's CPU quota, cores / SMT siblings / NUMA
nodes, how busy every CPU is over half a second (other tenants of the box), the NUMA node of every GPU.  CPU only."""
import glob
import os
import time


def read(path):
    try:
        with open(path) as fh:
            return fh.read().strip()
    except OSError:
        return None


def stat():
    out = {}
    with open("/proc/stat") as fh:
        for line in fh:
            p = line.split()
            if p[0].startswith("cpu") and p[0] != "cpu":
                v = [int(x) for x in p[1:]]
                out[int(p[0][3:])] = (sum(v), v[3] + v[4])
    return out


allowed = sorted(os.sched_getaffinity(0))
print("allowed cpus:", len(allowed), allowed[:4], "...", allowed[-4:])
print("cpu.max:", read("/sys/fs/cgroup/cpu.max"), "| v1 quota:", read("/sys/fs/cgroup/cpu/cpu.cfs_quota_us"), read("/sys/fs/cgroup/cpu/cpu.cfs_period_us"))
print("cpuset.cpus.effective:", read("/sys/fs/cgroup/cpuset.cpus.effective"))
nodes = {}
for d in sorted(glob.glob("/sys/devices/system/node/node[0-9]*")):
    nodes[int(d.rsplit("node", 1)[1])] = read(d + "/cpulist")
print("numa nodes:", nodes)
sib = {}
for c in allowed:
    sib[c] = read("/sys/devices/system/cpu/cpu%d/topology/thread_siblings_list" % c)
print("siblings of the first cpus:", {c: sib[c] for c in allowed[:4]})
a = stat(); time.sleep(0.5); b = stat()
busy = {c: 1.0 - (b[c][1] - a[c][1]) / max(1, b[c][0] - a[c][0]) for c in b}
line = " ".join("%d:%.0f" % (c, 100 * busy[c]) for c in sorted(busy) if busy[c] > 0.2)
print("cpus busier than 20 %% over 0.5 s (%d of %d): %s" % (sum(1 for c in busy if busy[c] > 0.2), len(busy), line))
for d in sorted(glob.glob("/sys/class/drm/card[0-9]*/device")):
    if read(d + "/vendor") == "0x1002":
        print("gpu", os.path.basename(os.path.dirname(d)), os.path.basename(os.path.realpath(d)), "numa_node", read(d + "/numa_node"), "local_cpulist", read(d + "/local_cpulist"))
print("loadavg:", read("/proc/loadavg"))
