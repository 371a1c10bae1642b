#!/usr/bin/env python3
"""Where the waves of each kernel spend their cycles, from rocprofv3 PMC passes (--kernel-trace only) over the same command.
One row per (kernel symbol, grid size) -- the grid tells the layers of a shape apart.

pass A: --pmc GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES
pass B: --pmc GRBM_GUI_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_VMEM

SQ_WAIT_ANY = wave parked (s_waitcnt / barrier), SQ_WAIT_INST_ANY = issue stall, SQ_ACTIVE_INST_ANY = issuing; the three add up to SQ_WAVE_CYCLES
(MI355X_MICROARCH.md, counter table).

usage: pmc_wave_summary.py <name filter> <counter_collection.csv> [<counter_collection.csv> ...]
"""
import csv, sys
from collections import defaultdict


def main():
    flt = sys.argv[1]
    per = defaultdict(lambda: defaultdict(float)); dur = defaultdict(float); cnt = defaultdict(int)
    for path in sys.argv[2:]:
        seen = set()
        for row in csv.DictReader(open(path)):
            if flt not in row["Kernel_Name"]:
                continue
            k = (row["Kernel_Name"].split("(")[0][-40:], int(row["Grid_Size"]) // max(1, int(row["Workgroup_Size"])))
            per[k][path + ":" + row["Counter_Name"]] += float(row["Counter_Value"])
            per[k][row["Counter_Name"]] = per[k][path + ":" + row["Counter_Name"]]
            if row["Dispatch_Id"] not in seen and path == sys.argv[2]:
                seen.add(row["Dispatch_Id"])
                dur[k] += float(row["End_Timestamp"]) - float(row["Start_Timestamp"]); cnt[k] += 1
    print("%-42s %8s %3s %8s | %6s %6s %6s %6s | %6s | %6s %6s %6s" % ("kernel", "blocks", "n", "avg ms", "parked", "stall", "issue", "st.lds", "mfma", "ldsact", "confl", "vmem"))
    for k, c in sorted(per.items(), key=lambda kv: -dur[kv[0]]):
        wc = c.get("SQ_WAVE_CYCLES", 0.0) or 1.0
        gui = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0 or 1.0
        lds = c.get("SQ_LDS_IDX_ACTIVE", 0.0)
        print("%-42s %8d %3d %8.3f | %6.3f %6.3f %6.3f %6.3f | %6.3f | %6.3f %6.3f %6.3f" % (
            k[0], k[1], cnt[k], dur[k] / max(1, cnt[k]) / 1e6,
            c.get("SQ_WAIT_ANY", 0) / wc, c.get("SQ_WAIT_INST_ANY", 0) / wc, c.get("SQ_ACTIVE_INST_ANY", 0) / wc, c.get("SQ_WAIT_INST_LDS", 0) / wc,
            c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024),
            lds / (gui * 256) if lds else 0.0, c.get("SQ_LDS_BANK_CONFLICT", 0) / lds if lds else 0.0, c.get("SQ_ACTIVE_INST_VMEM", 0) / (gui * 1024)))


if __name__ == "__main__":
    main()
