#!/usr/bin/env python3
"""Effective shader clock and matrix-core occupancy per kernel symbol from one rocprofv3 PMC pass
(--pmc GRBM_GUI_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES, --kernel-trace only) of bench.py.

rocprofv3 reports GRBM_GUI_ACTIVE summed over the 8 XCDs and SQ_VALU_MFMA_BUSY_CYCLES summed over all 1024 SIMDs (checked:
busy cycles = 64 x the number of v_mfma_f32_32x32x2_f32 the kernel's flop count implies, to 2 %).
  effective clock   = GRBM_GUI_ACTIVE / 8 / kernel duration (the chip clocks to its power budget: MI355X_MICROARCH.md, "DVFS give-back")
  MFMA pipe busy    = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024 SIMDs)     (256 CUs x 4 SIMDs)
The f32 MFMA peak of 157.3 TFLOP/s is quoted at 2.4 GHz; at the clock a kernel actually sustains the ceiling is
157.3 * clock / 2.4.

usage: pmc_clock_summary.py <counter_collection.csv> <out.json>
"""
import csv, json, sys
from collections import defaultdict


def main():
    per = defaultdict(lambda: defaultdict(float))
    dur = defaultdict(float); cnt = defaultdict(int)
    seen = set()
    for row in csv.DictReader(open(sys.argv[1])):
        k = row["Kernel_Name"]
        per[k][row["Counter_Name"]] += float(row["Counter_Value"])
        key = row["Dispatch_Id"]
        if key not in seen:
            seen.add(key)
            dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); cnt[k] += 1
    out = {}
    for k in per:
        gui = per[k].get("GRBM_GUI_ACTIVE", 0.0)
        if not gui or not dur[k]:
            continue
        gui /= 8.0                                             # per-XCD cycles
        clock = gui / dur[k]                                   # cycles per ns = GHz
        out[k] = {"launches": cnt[k], "avg_ms": round(dur[k] / cnt[k] / 1e6, 4), "effective_clock_ghz": round(clock, 3),
                  "f32_mfma_ceiling_tflops_at_clock": round(157.3 * clock / 2.4, 1),
                  "mfma_pipe_busy": round(per[k].get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui * 1024), 4)}
    json.dump(out, open(sys.argv[2], "w"), indent=1)
    for k, v in sorted(out.items(), key=lambda kv: -kv[1]["avg_ms"] * kv[1]["launches"])[:12]:
        print("%-90s n=%4d %8.3f ms  %.2f GHz  mfma busy %.3f" % (k[:90], v["launches"], v["avg_ms"], v["effective_clock_ghz"], v["mfma_pipe_busy"]))


if __name__ == "__main__":
    main()
