#!/usr/bin/env python
"""How much of a pass is launch gap?  Reads a `rocprofv3 --kernel-trace` CSV of `bench.py --mode infer --batch 1` (or any run whose passes
start with preprocess_kernel), splits it into passes and prints, per pass (median over the steady-state passes): wall span, time some
kernel is running, idle time between consecutive kernels of the pass, idle time before the next pass, launches.  The idle time inside a
pass is the most a captured graph (hipGraph) of the pass could remove.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -o t -- python bench.py --mode infer --batch 1 --steps 40 --warmup 5 --no-cpu-baseline
    python tools/launch_gap_report.py gpurun_out/trace/**/t_kernel_trace.csv
"""
import csv
import statistics
import sys


def main(path, marker="preprocess_kernel"):
    rows = list(csv.DictReader(open(path)))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    idx = [i for i, e in enumerate(ev) if marker in e[2]]
    if len(idx) < 12:
        raise SystemExit("fewer than 12 passes in the trace")
    lo = len(idx) // 3                      # skip warm-up and the profiled extra passes at the start
    stats = []
    for a, b in zip(idx[lo:-1], idx[lo + 1:]):
        seg = ev[a:b]
        busy = sum(e[1] - e[0] for e in seg)
        gaps = [seg[i + 1][0] - seg[i][1] for i in range(len(seg) - 1)]
        stats.append((ev[b][0] - seg[0][0], busy, sum(g for g in gaps if g > 0), max(0, ev[b][0] - seg[-1][1]), len(seg),
                      max(gaps) if gaps else 0))
    med = lambda i: statistics.median(s[i] for s in stats)
    print("passes analysed: %d (of %d)" % (len(stats), len(idx)))
    print("wall span per pass        %8.1f us" % (med(0) / 1e3))
    print("a kernel is running       %8.1f us" % (med(1) / 1e3))
    print("idle between its kernels  %8.1f us  (%.1f %% of the span; %d launches, %.2f us per gap, largest %.1f us)"
          % (med(2) / 1e3, 100.0 * med(2) / med(0), med(4), med(2) / 1e3 / max(1, med(4) - 1), med(5) / 1e3))
    print("idle before the next pass %8.1f us" % (med(3) / 1e3))


if __name__ == "__main__":
    main(sys.argv[1])
