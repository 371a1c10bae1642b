#!/usr/bin/env python
"""Parity report: the HIP path (through the C ABI) against the CPU oracle at BASELINE.json's sizes, as numbers in a tracked file
instead of pytest dots.  Runs on the GPU box:

    python tools/parity_report.py --out gpurun_out/parity_r04.json          (then copy to profiles/)

For each algorithm variant of the 3x3 stack -- "f6" (the default: Winograd F(6x6,3x3) + F(4x4,4x4) for fc6), "f4" (largest tile
4x4), "direct" (no Winograd anywhere: summation order is then the only difference from the oracle) -- it records

  c1 (256x256, one image, seed 7) and c2 (1024x512, one image):
      max |logit - oracle logit| for two decoders (logits O(100) and O(1-10)), and the number of pixels whose argmax differs from the
      oracle's, split by the oracle's top-2 logit margin: above 2e-3 (x logit scale) -- must be 0 -- and at or below it.
  c3 (1024x512, one image, full width): the error of each of the 42 gradient tensors, as max |g - g_ref| / max |g_ref| and as
      |g - g_ref|_2 / |g_ref|_2, against the fp32 oracle AND against the same oracle run in float64 -- next to the fp32 oracle's own
      distance from float64, which is what "as accurate as the reference's fp32 CPU path" has to be measured against.  Each comparison
      is made twice: against the oracle's own ReLU / max-pool decisions ("raw") and against the oracle differentiating along the decisions
      the device took ("aligned", Engine.relu_branches / pool_routes -> oracle `branches=` / `routes=`), with the number of units / windows
      that differ and how close to a tie they sit.

Imports only the package and oracle/ (checker)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402


VARIANTS = {
    "f6": {},
    "f4": {"winograd_tile": 4},
    "direct": {"winograd_min_cin": 0, "winograd_fc6": 0},
    "f6_b1f4": {"winograd_tile_hires": 4, "winograd_hires_pixels": 512 * 1024},            # block 1 (conv1_2) on F(4x4), the rest F(6x6)
    "f6_b12f4": {"winograd_tile_hires": 4, "winograd_hires_pixels": 256 * 512},            # blocks 1 and 2 on F(4x4)
}


# what the oracle has to round for a precision mode to be "the same arithmetic" (oracle.forward / loss_and_grads keywords)
ORACLE_KW = {"bf16_fc": {"bf16_fc": True}, "bf16_fwd": {"bf16_fc": True, "bf16_convs": True}, "bf16_fwd_x2": {"bf16_fc": True, "bf16_convs": True}}


def make_engine(options, precision="fp32"):
    from fcn8s_tensorflow_amd.engine import Engine
    return Engine(20, options=options, precision=precision)


def logits_block(e, P, img, ref, ref_arg):
    e.set_params(P)
    pred = e.predict(img, argmax=True)
    n, h, w = img.shape[:3]
    logits = e.activation("logits", (n, h, w, 20))
    scale = max(1.0, float(np.abs(ref).max()))
    srt = np.sort(ref, -1)
    margin = srt[..., -1] - srt[..., -2]
    wrong = pred != ref_arg
    thr = 2e-3 * scale
    return {
        "max_abs_logit": float(np.abs(ref).max()),
        "max_abs_logit_error": float(np.abs(logits - ref).max()),
        "max_logit_error_over_scale": float(np.abs(logits - ref).max() / scale),
        "pixels": int(pred.size),
        "argmax_mismatch_total": int(wrong.sum()),
        "argmax_mismatch_margin_gt_2e-3": int((wrong & (margin > thr)).sum()),
        "argmax_mismatch_margin_le_2e-3": int((wrong & (margin <= thr)).sum()),
        "pixels_with_margin_le_2e-3": int((margin <= thr).sum()),
        "largest_margin_among_mismatches": float(margin[wrong].max()) if wrong.any() else 0.0,
    }


def grad_errors(g, ref):
    out = {}
    for k in ref:
        a = np.asarray(g[k], np.float64); b = np.asarray(ref[k], np.float64)
        out[k] = {"max_over_max": float(np.abs(a - b).max() / (np.abs(b).max() + 1e-300)),
                  "l2_rel": float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-300))}
    return out


def summarize(errs):
    v = [x["max_over_max"] for x in errs.values()]
    worst = sorted(errs.items(), key=lambda kv: -kv[1]["max_over_max"])[:5]
    return {"worst_max_over_max": float(max(v)), "median_max_over_max": float(np.median(v)),
            "worst_five": [[k, x["max_over_max"]] for k, x in worst]}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "parity_r05.json"))
    ap.add_argument("--variants", default="f6,f4,direct")
    ap.add_argument("--precisions", default="fp32", help="comma list out of fp32,f32x3,f32x2,bf16_fc,bf16_fwd,bf16_fwd_x2 (run for the f6 variant only; the bf16 modes "
                    "are compared with the oracle applying the same operand rounding to the same layers)")
    ap.add_argument("--threads", type=int, default=32)
    ap.add_argument("--skip-f64", action="store_true")
    ap.add_argument("--small", action="store_true", help="quarter-size c2 / c3 (plumbing check)")
    args = ap.parse_args()
    import torch
    torch.set_num_threads(args.threads)
    from oracle import fcn8s_oracle as orc

    H2, W2 = (256, 512) if args.small else (512, 1024)
    rep = {"tool": "tools/parity_report.py", "device": torch.cuda.get_device_name(0), "sizes": {"c1": [1, 256, 256], "c2": [1, H2, W2], "c3": [1, H2, W2]},
           "margin_threshold": "2e-3 x max(1, max |logit|)", "variants": {}}
    t0 = time.time()
    # ---- oracle side, once -----------------------------------------------------------------------------------------------------------
    rng = np.random.default_rng(7)
    img1 = rng.integers(0, 256, (1, 256, 256, 3), dtype=np.uint8)                       # c1 = one 256x256 image, seed 7
    img2, lab2 = orc.synthetic_batch(1, H2, W2)
    P_hi = orc.init_params(20, seed=0, decoder_std_scale=30.0, bias_std=0.05)            # logits O(100): real margins
    P_lo = orc.init_params(20, seed=1, decoder_std_scale=5.0, bias_std=0.05)             # logits O(1-10): north_star's absolute 1e-3
    P_g = orc.init_params(20, seed=4, decoder_std_scale=30.0, bias_std=0.05)             # the gradient case of tests/test_fullsize_gpu.py
    onehot = orc.one_hot(lab2, 20).astype(np.float32)

    def oracle_side(kw, want_f64):
        """Everything the comparisons need from the oracle for one arithmetic (kw = the operand rounding it applies)."""
        fwd = {}
        for cname, img in (("c1", img1), ("c2", img2)):
            for pname, P in (("decoder_x30", P_hi), ("decoder_x5", P_lo)):
                ref = orc.forward(P, img, **kw)
                fwd[(cname, pname)] = (P, img, ref, np.argmax(orc.softmax(ref), -1))
        loss32, g32, _ = orc.loss_and_grads(P_g, img2, onehot, l2_rate=1e-3, **kw)
        _, acts_g = orc.forward(P_g, img2, keep=True, **kw)
        own_routes, route_gaps = orc.pool_routes(acts_g)
        acts_g = {k: acts_g[k] for k in orc.branch_layers()}
        g64 = loss64 = None
        truth = None
        if want_f64:
            loss64, g64, _ = orc.loss_and_grads(P_g, img2, onehot, l2_rate=1e-3, dtype=torch.float64, **kw)
            # the float64 run of config 2 (O(1-10) decoder): the truth both the device and the fp32 oracle are measured against
            l64 = orc.forward(P_lo, img2, dtype=torch.float64, **kw)
            l32 = fwd[("c2", "decoder_x5")][2].astype(np.float64)
            truth = {"l64": l64, "a64": np.argmax(orc.softmax(l64), -1), "oracle_fp32_max_abs_logit_error_vs_f64": float(np.abs(l32 - l64).max()),
                     "oracle_fp32_argmax_mismatch_vs_f64": int((fwd[("c2", "decoder_x5")][3] != np.argmax(orc.softmax(l64), -1)).sum())}
        return {"truth": truth, "fwd": fwd, "loss32": loss32, "g32": g32, "acts_g": acts_g, "own_routes": own_routes, "route_gaps": route_gaps, "g64": g64, "loss64": loss64}

    sides = {"": oracle_side({}, not args.skip_f64)}
    rep["oracle_seconds_fp32_and_fp64"] = time.time() - t0
    if sides[""]["g64"] is not None:
        e = grad_errors(sides[""]["g32"], sides[""]["g64"])
        rep["oracle_fp32_vs_fp64"] = {"loss_fp32": sides[""]["loss32"], "loss_fp64": sides[""]["loss64"], "summary": summarize(e), "per_tensor": e}
    # ---- GPU side ---------------------------------------------------------------------------------------------------------------------
    runs = [(v, "fp32") for v in args.variants.split(",") if v]
    runs += [("f6", p) for p in args.precisions.split(",") if p and p != "fp32"]
    for vname, prec in runs:
        key = vname if prec == "fp32" else vname + "/" + prec
        kw = ORACLE_KW.get(prec, {})
        skey = json.dumps(kw, sort_keys=True) if kw else ""
        if skey not in sides:
            sides[skey] = oracle_side(kw, False)           # (the float64 run of a rounded graph says nothing new)
        S = sides[skey]
        fwd, loss32, g32, acts_g, own_routes, route_gaps, g64 = S["fwd"], S["loss32"], S["g32"], S["acts_g"], S["own_routes"], S["route_gaps"], S["g64"]
        e = make_engine(VARIANTS[vname], prec)
        block = {"options": VARIANTS[vname], "precision": prec, "oracle_rounding": kw}
        for (cname, pname), (P, img, ref, ref_arg) in fwd.items():
            block.setdefault(cname, {})[pname] = logits_block(e, P, img, ref, ref_arg)
        if S.get("truth"):
            T = S["truth"]
            e.set_params(P_lo)
            pred = e.predict(img2, argmax=True)
            lg = e.activation("logits", (1, H2, W2, 20)).astype(np.float64)
            srt = np.sort(T["l64"], -1)
            safe = (srt[..., -1] - srt[..., -2]) > 2e-3
            dev_err = float(np.abs(lg - T["l64"]).max())
            block["float64_truth_c2_decoder_x5"] = {
                "what": "config 2 (1024x512, one image, O(1-10) logits) against the oracle run in float64; nothing aligned",
                "device_max_abs_logit_error_vs_f64": dev_err, "oracle_fp32_max_abs_logit_error_vs_f64": T["oracle_fp32_max_abs_logit_error_vs_f64"],
                "ratio": dev_err / T["oracle_fp32_max_abs_logit_error_vs_f64"],
                "device_argmax_mismatch_vs_f64": int((pred != T["a64"]).sum()), "oracle_fp32_argmax_mismatch_vs_f64": T["oracle_fp32_argmax_mismatch_vs_f64"],
                "device_argmax_mismatch_vs_f64_above_margin_2e-3": int((pred != T["a64"])[safe].sum()), "pixels": int(pred.size)}
        e.set_params(P_g)
        loss = e.forward_backward(img2, lab2, keep_prob=1.0, l2_rate=1e-3)
        g = e.get_grads()
        e32 = grad_errors(g, g32)
        block["c3"] = {"loss": loss, "loss_oracle_fp32": loss32, "vs_oracle_fp32": {"summary": summarize(e32), "per_tensor": e32}}
        if g64 is not None:
            e64 = grad_errors(g, g64)
            block["c3"]["vs_oracle_fp64"] = {"summary": summarize(e64), "per_tensor": e64}
            o = rep["oracle_fp32_vs_fp64"]["summary"]
            block["c3"]["float64_truth_unaligned"] = {"device_worst": summarize(e64)["worst_max_over_max"], "oracle_fp32_worst": o["worst_max_over_max"],
                                                      "ratio_worst": summarize(e64)["worst_max_over_max"] / o["worst_max_over_max"],
                                                      "device_median": summarize(e64)["median_max_over_max"], "oracle_fp32_median": o["median_max_over_max"],
                                                      "ratio_median": summarize(e64)["median_max_over_max"] / o["median_max_over_max"]}
        # the same comparison along the branches the device took
        br = e.relu_branches((1, H2, W2))
        n_units = int(sum(v.size for v in br.values()))
        n_diff, worst = 0, 0.0
        for k, on in br.items():
            d = on != (acts_g[k] > 0)
            if d.any():
                n_diff += int(d.sum())
                gpu = e.activation(k, acts_g[k].shape, missing_ok=True)
                big = np.abs(acts_g[k][d]) if gpu is None else np.maximum(np.abs(gpu[d]), np.abs(acts_g[k][d]))
                worst = max(worst, float(big.max() / (np.abs(acts_g[k]).max() + 1e-30)))
        rt = e.pool_routes((1, H2, W2))
        n_win, n_rdiff, worst_gap = 0, 0, 0.0
        for k in rt:
            d = rt[k] != own_routes[k]
            n_win += rt[k].size
            if d.any():
                n_rdiff += int(d.sum())
                tie = d & ((rt[k] == 4) == (own_routes[k] == 4))
                if tie.any():
                    worst_gap = max(worst_gap, float(route_gaps[k][tie].max()))
        block["c3"]["pool_windows"] = n_win
        block["c3"]["pool_routes_differing_from_oracle"] = n_rdiff
        block["c3"]["largest_relative_gap_between_the_two_window_maxima_at_a_differing_route"] = worst_gap
        block["c3"]["relu_units"] = n_units
        block["c3"]["relu_units_differing_from_oracle"] = n_diff
        block["c3"]["largest_activation_at_a_differing_unit_over_layer_max"] = worst
        e.close()
        _, ga32, _ = orc.loss_and_grads(P_g, img2, onehot, l2_rate=1e-3, branches=br, routes=rt, **kw)
        ea32 = grad_errors(g, ga32)
        block["c3"]["aligned_vs_oracle_fp32"] = {"summary": summarize(ea32), "per_tensor": ea32}
        if g64 is not None:
            _, ga64, _ = orc.loss_and_grads(P_g, img2, onehot, l2_rate=1e-3, dtype=torch.float64, branches=br, routes=rt, **kw)
            ea64 = grad_errors(g, ga64)
            block["c3"]["aligned_vs_oracle_fp64"] = {"summary": summarize(ea64), "per_tensor": ea64}
            eo = grad_errors(ga32, ga64)
            block["c3"]["oracle_fp32_vs_fp64_on_these_branches"] = {"summary": summarize(eo)}
        del br
        rep["variants"][key] = block
        print(key, "c2 x30:", block["c2"]["decoder_x30"], "\n   c3 vs fp32 oracle:", block["c3"]["vs_oracle_fp32"]["summary"],
              "\n   c3 vs fp64 oracle:", block["c3"].get("vs_oracle_fp64", {}).get("summary"),
              "\n   c3 relu units differing:", n_diff, "of", n_units, "worst", worst, "; pool routes differing:", n_rdiff, "of", n_win, "worst gap", worst_gap,
              "\n   c3 aligned vs fp32 oracle:", block["c3"]["aligned_vs_oracle_fp32"]["summary"],
              "\n   c3 aligned vs fp64 oracle:", block["c3"].get("aligned_vs_oracle_fp64", {}).get("summary"), flush=True)
    if "oracle_fp32_vs_fp64" in rep:
        print("oracle fp32 vs fp64:", rep["oracle_fp32_vs_fp64"]["summary"])
    rep["seconds"] = time.time() - t0
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(rep, f, indent=1)
    print("wrote", args.out)


if __name__ == "__main__":
    main()
