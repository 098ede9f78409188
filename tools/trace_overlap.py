#!/usr/bin/env python3
"""Timeline of a rocprofv3 --kernel-trace --memory-copy-trace run of one of the command-line tools: which engine was
busy when (host -> device copies, kernels, device -> host copies), in bins, and the bursts of each kind -- shows that a
later dataset's upload and kernels run under an earlier dataset's plane traffic to the host coder (BASELINE configs[4]:
"async H2D/D2H overlap").

    python tools/trace_overlap.py DIR_WITH_CSVS [bin_ms] > profiles/rNN/<name>.txt"""
import csv
import glob
import os
import sys


def load(d, pattern, kind_of):
    ev = []
    for f in glob.glob(os.path.join(d, "**", pattern), recursive=True):
        with open(f) as fh:
            for r in csv.DictReader(fh):
                ev.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), kind_of(r)))
    return ev


def bursts(ev, gap_ns=5e6):
    out = []
    for s, e, _ in sorted(ev):
        if out and s - out[-1][1] < gap_ns:
            out[-1][1] = max(out[-1][1], e); out[-1][2] += 1; out[-1][3] += e - s
        else:
            out.append([s, e, 1, e - s])
    return out


def main():
    d = sys.argv[1]
    bin_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 50.0
    kern = load(d, "*kernel_trace.csv", lambda r: r["Kernel_Name"].split("(")[0].replace("void ", ""))
    cp = load(d, "*memory_copy_trace.csv", lambda r: "H2D" if "HOST_TO_DEVICE" in r["Direction"] else "D2H")
    h2d = [e for e in cp if e[2] == "H2D"]
    d2h = [e for e in cp if e[2] == "D2H"]
    t0 = min(e[0] for e in kern + cp)
    t1 = max(e[1] for e in kern + cp)
    print("events: %d kernels, %d host->device copies, %d device->host copies over %.3f s" % (len(kern), len(h2d), len(d2h), (t1 - t0) * 1e-9))
    for name, ev, gap in (("host->device copies", h2d, 20e6), ("kernels", kern, 20e6)):
        print("\n%s: bursts (gap > 20 ms starts a new one)" % name)
        for s, e, n, busy in bursts(ev, gap):
            print("  %8.3f - %8.3f s   %4d events, engine busy %7.2f ms" % ((s - t0) * 1e-9, (e - t0) * 1e-9, n, busy * 1e-6))
    for name, ev in (("host->device copies", h2d), ("device->host copies", d2h)):
        if ev:
            print("\n%s: first %.3f s, last %.3f s, %d events, engine busy %.1f ms in all, largest gap %.0f ms"
                  % (name, (min(e[0] for e in ev) - t0) * 1e-9, (max(e[1] for e in ev) - t0) * 1e-9, len(ev), sum(e[1] - e[0] for e in ev) * 1e-6,
                     max([b[0] - a[1] for a, b in zip(sorted(ev), sorted(ev)[1:])] + [0]) * 1e-6))
    nb = int((t1 - t0) / (bin_ms * 1e6)) + 1
    rows = {"H2D": [0.0] * nb, "kernels": [0.0] * nb, "D2H": [0.0] * nb}
    for s, e, k in h2d + d2h + [(a, b, "kernels") for a, b, _ in kern]:
        b0, b1 = int((s - t0) / (bin_ms * 1e6)), int((e - t0) / (bin_ms * 1e6))
        for b in range(b0, b1 + 1):
            lo, hi = max(s, t0 + b * bin_ms * 1e6), min(e, t0 + (b + 1) * bin_ms * 1e6)
            rows[k][b] += max(0.0, hi - lo)
    print("\nbusy fraction of the engine per %.0f ms bin ('.' idle, 1-9 tenths, '#' > 95 %%; kernels: x10, they are milliseconds long):" % bin_ms)
    for k in ("H2D", "kernels", "D2H"):
        line = ""
        for v in rows[k]:
            f = v / (bin_ms * 1e6) * (10.0 if k == "kernels" else 1.0)
            line += "." if f < 0.005 else "#" if f > 0.95 else str(max(1, min(9, int(f * 10))))
        print("  %-8s %s" % (k, line))
    # encode: a later dataset's field upload / kernels inside the span in which an earlier dataset's planes still travel to
    # the host coder (window by window, as the coder asks for them); decode: a later dataset's plane uploads (decoded
    # windows) before an earlier dataset's field download
    ub, kb = bursts(h2d, 20e6), bursts(kern, 20e6)
    if d2h and len(ub) > 1 and len(ub) <= 8:
        first_d2h, last_d2h = min(e[0] for e in d2h), max(e[1] for e in d2h)
        n_up = sum(1 for u in ub[1:] if first_d2h < u[0] < last_d2h)
        n_k = sum(1 for k in kb[1:] if first_d2h < k[0] < last_d2h)
        print("\nfield uploads of later datasets that start while device->host plane traffic is flowing: %d of %d; kernel stages: %d of %d"
              % (n_up, len(ub) - 1, n_k, len(kb) - 1))
    db = bursts(d2h, 20e6)
    if h2d and len(db) >= 1 and len(ub) > 8:
        last_h2d = max(e[1] for e in h2d)
        print("\nfield downloads (device->host bursts) that finish while later datasets' decoded windows are still being uploaded: %d of %d"
              % (sum(1 for b in db if b[1] < last_h2d), len(db)))


if __name__ == "__main__":
    main()
