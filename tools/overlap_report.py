#!/usr/bin/env python
"""Reads a `rocprofv3 --kernel-trace --output-format csv` kernel trace of bench.py and reports how the kernels of the side queue (deferred
weight gradients, model.hip) lie against the kernels of the main queue, for the LAST complete training step of the trace (steps are
split at the preprocessing kernel that opens every forward pass):

  * step duration, busy time of each queue, time with both queues busy;
  * the timeline of that step from the first side-queue kernel on: every kernel of both queues with start / duration / queue;
  * per kernel symbol: launches and summed duration inside the co-run window.

    python tools/overlap_report.py <kernel_trace.csv> [--timeline out.txt]
"""
import csv
import sys
from collections import defaultdict


def short(name, n=64):
    name = name.replace("fcn8s::", "").replace("void ", "")
    return name if len(name) <= n else name[:n - 3] + "..."


def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for a, b in iv:
        if cs is None:
            cs, ce = a, b
        elif a <= ce:
            ce = max(ce, b)
        else:
            tot += ce - cs; cs, ce = a, b
    if cs is not None:
        tot += ce - cs
    return tot


def inter(iv1, iv2):
    """total time covered by both interval sets (each first merged)"""
    def merge(iv):
        out = []
        for a, b in sorted(iv):
            if out and a <= out[-1][1]:
                out[-1][1] = max(out[-1][1], b)
            else:
                out.append([a, b])
        return out
    a, b = merge(iv1), merge(iv2)
    i = j = tot = 0
    while i < len(a) and j < len(b):
        lo, hi = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if hi > lo:
            tot += hi - lo
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    path = sys.argv[1]
    tl_path = sys.argv[sys.argv.index("--timeline") + 1] if "--timeline" in sys.argv else None
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "?")))
    rows.sort()
    nq = defaultdict(int)
    for r in rows:
        nq[r[3]] += 1
    main_q = max(nq, key=nq.get)
    print("queues (id: kernels):", dict(nq), "-> main queue", main_q)
    starts = [i for i, r in enumerate(rows) if "preprocess" in r[2] and r[3] == main_q]
    if len(starts) < 2:
        print("fewer than two forward passes in the trace"); return
    # last complete TRAINING step: the last [preprocess, next preprocess) span that contains an optimizer kernel
    step = None
    for a, b in zip(starts[:-1], starts[1:]):
        if any(("adam" in r[2] or "sgd" in r[2]) for r in rows[a:b]):
            step = (a, b)
    if step is None:
        print("no training step found"); return
    srows = rows[step[0]:step[1]]
    t0 = srows[0][0]
    tend = max(r[1] for r in srows)
    mainiv = [(r[0], r[1]) for r in srows if r[3] == main_q]
    sideiv = [(r[0], r[1]) for r in srows if r[3] != main_q]
    print("last complete training step: %.3f ms from its first kernel to its last; %d kernels on the main queue, %d on the side queue"
          % ((tend - t0) / 1e6, len(mainiv), len(sideiv)))
    print("  main queue busy %.3f ms, side queue busy %.3f ms, both busy at once %.3f ms, either busy %.3f ms"
          % (union(mainiv) / 1e6, union(sideiv) / 1e6, inter(mainiv, sideiv) / 1e6, union(mainiv + sideiv) / 1e6))
    if not sideiv:
        print("nothing ran on a side queue in that step"); return
    w0, w1 = min(a for a, _ in sideiv), max(b for _, b in sideiv)
    print("  side-queue window: starts %.3f ms into the step, lasts %.3f ms" % ((w0 - t0) / 1e6, (w1 - w0) / 1e6))
    agg = defaultdict(lambda: [0, 0.0])
    for r in srows:
        if r[1] > w0 and r[0] < w1:
            k = ("side " if r[3] != main_q else "main ") + short(r[2], 70)
            agg[k][0] += 1; agg[k][1] += (r[1] - r[0]) / 1e6
    print("\nkernels inside the side-queue window, summed per symbol:")
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        if ms >= 0.02:
            print("   %-76s x%-3d %8.3f ms" % (k, n, ms))
    lines = ["# queue  start_ms  dur_ms  kernel   (last complete training step, from its first kernel; side queue = deferred weight gradients)"]
    for r in srows:
        lines.append("%s %9.3f %8.3f  %s" % ("side" if r[3] != main_q else "main", (r[0] - t0) / 1e6, (r[1] - r[0]) / 1e6, short(r[2], 90)))
    if tl_path:
        with open(tl_path, "w") as f:
            f.write("\n".join(lines) + "\n")
        print("\ntimeline of the step written to", tl_path)


if __name__ == "__main__":
    main()
