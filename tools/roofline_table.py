#!/usr/bin/env python3
"""Per-kernel-group roofline table of one bench line (the JSON line bench.py prints; its `kernel_groups_*` blocks come from the library's
HIP events on the launch stream, one pair per launch, recorded in a pass of their own after the timed regions).

    python tools/roofline_table.py profiles/r04_bench_train_bs16.json > profiles/r04_roofline_table.md

Every group is held against the roof that bounds it (max(algorithmic bytes / t / 8 TB/s, algorithmic flops / t / matrix peak), the line's
`kernel_groups_roof`; lines of rounds 1-5 lack that block: there a group that reports flops counts as an MFMA group): an MFMA group against the dense matrix peak of the arithmetic it runs
in -- 157.3 TFLOP/s for v_mfma_f32_32x32x2_f32, 2500 / 3 and 2500 / 6 "fp32-equivalent" TFLOP/s for the two- and three-piece bf16 modes, 2500
for the bf16 direct convolutions -- and an HBM group (it reports algorithmic bytes, no flops) against 8 TB/s.  The flops are the ALGORITHM's
(Winograd-domain multiplies for the Winograd GEMMs, not the direct-convolution count), the bytes are each tensor once per producing /
consuming kernel."""
import json
import sys

PEAK_F32, PEAK_BF16, PEAK_HBM = 157.3, 2500.0, 8000.0


def peak_for(group, dtype):
    if group.endswith("_bf16"):
        return PEAK_BF16, "bf16 MFMA"
    if "f32x3" in dtype and group.startswith(("wino_gemm", "fc7_", "tconv_")):
        return PEAK_BF16 / 6.0, "bf16 MFMA / 6 products"
    if ("f32x2" in dtype or "two bf16 pieces" in dtype) and group.startswith(("wino_gemm", "fc7_", "tconv_")):
        return PEAK_BF16 / 3.0, "bf16 MFMA / 3 products"
    return PEAK_F32, "f32 MFMA"


def main(path):
    line = [l for l in open(path) if l.startswith("{")][-1]
    d = json.loads(line)
    ms = d["kernel_groups_ms_per_step"]
    tf = d.get("kernel_groups_tflops", {})
    gb = d.get("kernel_groups_gbs", {})
    step = d["ms_per_step"]
    total = sum(ms.values())
    print("# Per-kernel-group roofline: %s" % d["metric"])
    print()
    print("`%s` -- %s images/s, %.3f ms per step (timed regions, no event recording); the groups below sum to %.2f ms (the pass with the "
          "library's HIP events on, %s ms per step)." % (path, d["value"], step, total, d.get("profiled_pass", {}).get("ms_per_step")))
    print("dtype: %s.  workload: %s." % (d["dtype"], d["config"]["workload"]))
    print()
    print("| kernel group | ms / step | share | bound by | achieved | peak | fraction |")
    print("|---|---:|---:|---|---:|---:|---:|")
    mfma_ms = hbm_ms = 0.0
    mfma_w = hbm_w = 0.0
    roof = d.get("kernel_groups_roof") or {}
    for g, t in sorted(ms.items(), key=lambda kv: -kv[1]):
        r = roof.get(g)
        # a group is held against the roof it sits closer to, max(bytes / t / 8 TB/s, flops / t / matrix peak); lines without that block (rounds 1-5)
        # fall back to "reports flops = MFMA group"
        if r and r["bound"] == "hbm" and g in gb:
            print("| %s | %.3f | %.1f %% | HBM%s | %.0f GB/s | %.0f | %.2f |" % (g, t, 100 * t / total, (" (MFMA %.2f)" % r["mfma_frac"]) if r["mfma_frac"] else "", gb[g], PEAK_HBM, r["frac"]))
            hbm_ms += t; hbm_w += t * r["frac"]
        elif g in tf:
            peak, what = peak_for(g, d["dtype"])
            frac = tf[g] / peak
            print("| %s | %.3f | %.1f %% | %s | %.1f TFLOP/s | %.1f | %.2f |" % (g, t, 100 * t / total, what, tf[g], peak, frac))
            mfma_ms += t; mfma_w += t * frac
        elif g in gb:
            frac = gb[g] / PEAK_HBM
            print("| %s | %.3f | %.1f %% | HBM | %.0f GB/s | %.0f | %.2f |" % (g, t, 100 * t / total, gb[g], PEAK_HBM, frac))
            hbm_ms += t; hbm_w += t * frac
        else:
            print("| %s | %.3f | %.1f %% | - | - | - | - |" % (g, t, 100 * t / total))
    print()
    if mfma_ms:
        print("MFMA-bound groups: %.2f ms (%.0f %% of the step) at a time-weighted %.2f of their peaks.  " % (mfma_ms, 100 * mfma_ms / total, mfma_w / mfma_ms), end="")
    if hbm_ms:
        print("HBM-bound groups: %.2f ms (%.0f %%) at a time-weighted %.2f of 8 TB/s." % (hbm_ms, 100 * hbm_ms / total, hbm_w / hbm_ms))
    h = d.get("roofline_hbm")
    if h and h.get("traffic"):
        print()
        print("Largest HBM-bound group `%s`: %.0f MB algorithmic per step; HBM-side traffic of all its kernels (PMC, FETCH_SIZE x2-corrected + WRITE_SIZE) %.0f MB per step = "
              "%.2fx (uncorrected fetch: %.2fx)." % (h["kernel_group"], h["algorithmic_mb_per_step"], h["traffic"] / 1e6, h["traffic_over_algorithmic"], h["traffic_over_algorithmic_lower_bound"]))
    r = d.get("roofline")
    if r:
        print()
        print("Dominant symbol `%s`: %.1f %s = %.3f of %.1f over %d launches of %.3f ms; HBM-side traffic per launch %s (PMC) against %.0f MB algorithmic."
              % (r["kernel"], r["achieved"], r["unit"], r["frac"], r["peak"], r["launches"], r["avg_launch_ms"],
                 ("%.0f MB" % (r["traffic"] / 1e6)) if r.get("traffic") else "n/a", r["algorithmic_mb_per_launch"]))
    print()
    print("(The K = 64 / 128 position GEMMs inside `wino_gemm_*` are HBM-bound streaming kernels, DESIGN.md section 4: held against the MFMA peak "
          "here they pull those groups' fractions down; `profiles/*_layer_bench.txt` lists them per layer.)")


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "profiles/r04_bench_train_bs16.json")
