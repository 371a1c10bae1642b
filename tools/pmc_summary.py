#!/usr/bin/env python3
"""Summarises two rocprofv3 PMC passes (one with --pmc FETCH_SIZE, one with --pmc WRITE_SIZE, each with
--kernel-trace only) of the same bench.py command into HBM bytes per launch per kernel symbol.

Units / corrections follow /opt/skills/guides (MI355X_MICROARCH.md, HBM section):
  * FETCH_SIZE / WRITE_SIZE are reported in KiB (bytes = value * 1024);
  * on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads -> the fetch side is DOUBLED.
    (The doubling is exact for streaming kernels -- it reproduces the optimizer's 4 x 538 MB -- but an UPPER BOUND for kernels
    whose loads are 64-byte row segments, such as the A operand of the GEMM kernels (16 floats per row and K-tile); the
    uncorrected figure is kept next to it as the lower bound.)

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [source note]
"""
import csv, json, sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        a = acc[row["Kernel_Name"]]
        a[0] += float(row["Counter_Value"]); a[1] += 1
    return acc


def main():
    fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    out = {"source": sys.argv[4] if len(sys.argv) > 4 else None,
           "corrections": "bytes = KiB*1024; FETCH_SIZE doubled (gfx950 128-B requests tallied as 64 B)", "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        f, nf = fetch.get(k, [0.0, 0]); w, nw = write.get(k, [0.0, 0])
        fm = 2.0 * f * 1024 / max(nf, 1) / 1e6; wm = w * 1024 / max(nw, 1) / 1e6
        out["kernels"][k] = {"launches": max(nf, nw), "fetch_mb_per_launch": round(fm, 3), "write_mb_per_launch": round(wm, 3),
                             "hbm_mb_per_launch": round(fm + wm, 3), "fetch_mb_per_launch_uncorrected": round(fm / 2, 3)}
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in sorted(out["kernels"].items(), key=lambda kv: -kv[1]["hbm_mb_per_launch"] * kv[1]["launches"])[:14]:
        print("%-100s n=%4d fetch %9.1f MB write %9.1f MB" % (k[:100], v["launches"], v["fetch_mb_per_launch"], v["write_mb_per_launch"]))


if __name__ == "__main__":
    main()
