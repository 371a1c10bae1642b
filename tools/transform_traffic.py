#!/usr/bin/env python3
"""Fabric-side traffic RATE of the HBM-bound kernels of a training step, from the two rocprofv3 PMC passes of bench.py (FETCH_SIZE, WRITE_SIZE;
--kernel-trace only) that tools/collect_profiles.sh writes: per kernel symbol and launch size (grid), bytes fetched / written per launch and
bytes / duration, with the fetch side raw and x2-corrected (MI355X_MICROARCH.md: on gfx950 FETCH_SIZE tallies 128-byte requests at 64 bytes; the
correction is exact for fully coalesced streams -- checked here on the optimizer kernel, whose streams are known: 3 reads + 2 writes of the
parameter buffer).  What it shows: counted at the fabric, the Winograd transforms move 5.5-6.6 TB/s -- a plain stream's rate;
their 4.6-5.0 TB/s of ALGORITHMIC bytes is that rate divided by the halo over-fetch (each XCD's L2 fetches its own copy of the two-pixel patch overlap).

usage: transform_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv>"""
import csv
import sys
from collections import defaultdict


def load(path, counter):
    acc = defaultdict(lambda: [0.0, 0, 0.0])
    for row in csv.DictReader(open(path)):
        if row["Counter_Name"] != counter:
            continue
        a = acc[(row["Kernel_Name"].split("(")[0].replace("void fcn8s::", "").replace("fcn8s::", ""), int(row["Grid_Size"]))]
        a[0] += float(row["Counter_Value"]) * 1024
        a[1] += 1
        a[2] += int(row["End_Timestamp"]) - int(row["Start_Timestamp"])
    return acc


def main():
    f, w = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
    want = ("wino_", "sgd_momentum", "tf_adam", "softmax_xent", "conv1_tile", "conv1_wgrad_mfma", "maxpool")
    rows = []
    for k in f:
        if not any(x in k[0] for x in want) or k not in w:
            continue
        n = f[k][1]
        t = 0.5 * (f[k][2] / n + w[k][2] / max(w[k][1], 1))          # ns per launch (the two passes agree to a percent)
        fb, wb = f[k][0] / n, w[k][0] / max(w[k][1], 1)
        rows.append((t * n, k, n, t, fb, wb))
    print("%-46s %10s %5s %9s %11s %11s %13s %13s" % ("kernel", "grid", "n", "us", "fetch MB", "write MB", "TB/s raw", "TB/s fetch x2"))
    for _, k, n, t, fb, wb in sorted(rows, reverse=True)[:40]:
        print("%-46s %10d %5d %9.1f %11.1f %11.1f %13.2f %13.2f" % (k[0][:46], k[1], n, t / 1e3, fb / 1e6, wb / 1e6, (fb + wb) / t / 1e3, (2 * fb + wb) / t / 1e3))


if __name__ == "__main__":
    main()
