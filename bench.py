#!/usr/bin/env python3
"""FCN-8s training throughput on MI355X (BASELINE.json metric).

One "step" = one full training step of the hot path (forward, softmax-CE loss,
backward through every VGG + decoder variable, TF-Adam update) on one synthetic
1024x512 batch of 16 images per GPU, inputs already resident in HBM.
`python bench.py --gpus N --steps K --warmup W` prints ONE JSON line on rank 0.
For N > 1 launch with `python -m torch.distributed.run --nproc-per-node N ...`:
one process per GPU, gradients all-reduced over RCCL/xGMI in three buckets that
overlap with the backward pass (weak scaling: 16 images per GPU).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FWD_GFLOP_PER_IMG_512x1024 = 445.04
TRAIN_GFLOP_PER_IMG_512x1024 = 1333.3
PEAK_F32_MFMA_TFLOPS = 157.3             # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
PEAK_HBM_GBS = 8000.0


def pmc_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes of this same command
    (profiles/pmc_traffic.json, written by tools/pmc_summary.py); None if not collected."""
    try:
        d = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
        for name, v in d.get("kernels", {}).items():
            if name.replace(" ", "").startswith("voidfcn8s::" + kernel.replace(" ", "")) or name.replace(" ", "").startswith("fcn8s::" + kernel.replace(" ", "")):
                out = {"hbm_mb_per_launch": v["hbm_mb_per_launch"], "fetch_mb": v["fetch_mb_per_launch"],
                       "write_mb": v["write_mb_per_launch"], "source": d.get("source")}
                if "fetch_mb_per_launch_uncorrected" in v:      # the x2 FETCH_SIZE correction is an upper bound for 64-byte row-segment loads
                    out["hbm_mb_per_launch_lower_bound"] = round(v["fetch_mb_per_launch_uncorrected"] + v["write_mb_per_launch"], 3)
                return out
    except Exception:
        pass
    return None


def cpu_baseline(h, w, seconds_budget=30.0, optimizer="sgd"):
    """CPU restatement of the reference graph (oracle, kind 'port'), timed on this
    host's cores on a bounded sample: bs1 training steps (fwd + bwd + the same optimizer
    as the GPU run) at the bench resolution.  TF1 itself is not installable here."""
    import torch
    from oracle import fcn8s_oracle as orc
    cores = os.cpu_count() or 1
    torch.set_num_threads(cores)
    P = orc.init_params(20, seed=0)
    img, lab = orc.synthetic_batch(1, h, w)
    onehot = orc.one_hot(lab, 20).astype(np.float32)
    m = {k: np.zeros_like(v) for k, v in P.items()}
    v_ = {k: np.zeros_like(v) for k, v in P.items()}
    n, t0, t = 0, time.perf_counter(), 0
    while True:
        _, g, _ = orc.loss_and_grads(P, img, onehot)
        t += 1
        for k in P:
            if optimizer == "adam":
                P[k], m[k], v_[k] = orc.tf_adam_step(P[k], g[k], m[k], v_[k], t, 1e-4)
            else:
                P[k], m[k] = orc.sgd_momentum_step(P[k], g[k], m[k], 1e-4)
        n += 1
        el = time.perf_counter() - t0
        if el > seconds_budget * 0.5 or n >= 3:
            break
    return {"value": round(n / el, 4), "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "%d training step(s) (fwd+bwd+%s) of 1 image %dx%d on torch-CPU fp32, %.1f s"
                      % (n, "TF-Adam" if optimizer == "adam" else "SGD+momentum", w, h, el)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=16, help="images per GPU")
    ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024)
    ap.add_argument("--optimizer", default="sgd", choices=["adam", "sgd"],
                    help="sgd = SGD+momentum as BASELINE.json config 3 names; adam = TF-Adam, the reference's own optimizer "
                         "(fcn8s_tensorflow.py:256) -- same step time to within 0.1 percent")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16_fc"],
                    help="fp32 = the reference's arithmetic (the headline number); bf16_fc = BASELINE config 5's mode (forward "
                         "fc6/fc7 with bf16 operands on the bf16 MFMA, fp32 accumulate) -- reported as dtype 'bf16_fc+f32'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--mode", default="train", choices=["train", "infer"])
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL); 'gloo' lets the "
                    "multi-rank path be exercised on a single-GPU box together with --device")
    ap.add_argument("--device", type=int, default=None, help="HIP device ordinal (default: LOCAL_RANK)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from fcn8s_tensorflow_amd import _lib as L
    from fcn8s_tensorflow_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = local_rank if args.device is None else args.device
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev))
        else:
            dist.init_process_group(args.backend)
    if args.gpus != world and rank == 0 and world == 1 and args.gpus > 1:
        print("bench.py: --gpus %d needs `python -m torch.distributed.run --nproc-per-node %d bench.py ...`" % (args.gpus, args.gpus), file=sys.stderr)
        sys.exit(2)

    trace = os.environ.get("FCN8S_BENCH_TRACE")
    def mark(msg):
        if trace:
            print("[bench rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)
    N, H, W = args.batch, args.height, args.width
    mark("process group up; creating engine")
    eng = Engine(20, device_id=dev, seed=1234 + rank, precision=args.precision)
    eng.init_params(seed=0)                       # He-normal VGG, reference decoder init (same on every rank)
    mark("broadcast params")
    eng.broadcast_params(0)
    mark("params broadcast")
    rng = np.random.default_rng(1234 + rank)      # SURVEY 8d synthetic inputs
    images = torch.from_numpy(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).cuda()
    labels = torch.from_numpy(rng.integers(0, 20, (N, H, W), dtype=np.uint8)).cuda()
    opt = L.OPT_TF_ADAM if args.optimizer == "adam" else L.OPT_SGD_MOMENTUM

    if args.mode == "infer":
        eng.freeze(True)                          # a serving loop: constant weights, transformed filters built once (fcn8s_freeze_params)

    def step():
        if args.mode == "train":
            eng.train_step(images, labels, 1e-4, keep_prob=0.5, l2_rate=0.0, optimizer=opt, fetch_loss=False)
        else:
            eng.predict(images, argmax=True)

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        mark("warmup step %d enqueued" % i)
    fence()
    mark("warmup done")
    eng.profile(True)
    eng.profile_reset()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    dt = time.perf_counter() - t0
    mark("timed region done")
    prof = eng.profile_results()
    eng.profile(False)
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    loss = eng.forward_backward(images, labels, keep_prob=1.0) if args.mode == "train" else None

    if rank == 0:
        ms = dt / args.steps * 1e3
        value = N * world * args.steps / dt
        scale = (H * W) / (512.0 * 1024.0)
        gflop_img = (TRAIN_GFLOP_PER_IMG_512x1024 if args.mode == "train" else FWD_GFLOP_PER_IMG_512x1024) * scale
        # dominant kernel family (by time) among the MFMA convolution groups
        # (the library keeps a second view of its HIP-event timings keyed by kernel symbol: "kernel:<name>")
        kern = {k[7:]: v for k, v in prof.items() if k.startswith("kernel:") and v["flops"] > 0 and v["launches"] > 0}
        prof = {k: v for k, v in prof.items() if not k.startswith("kernel:")}
        dom = max(kern, key=lambda k: kern[k]["ms"]) if kern else None
        roof = None
        if dom:
            g = kern[dom]
            ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
            tr = pmc_traffic(dom)
            roof = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                    # HBM-side bytes per launch from the PMC passes (FETCH_SIZE x2-corrected + WRITE_SIZE); compare with
                    # algorithmic_mb_per_launch * 1e6.  Details (and the uncorrected lower bound) in traffic_detail.
                    "traffic": (round(tr["hbm_mb_per_launch"] * 1e6) if tr else None), "traffic_unit": "bytes/launch", "traffic_detail": tr,
                    "launches": g["launches"], "avg_launch_ms": round(g["ms"] / g["launches"], 4),
                    "algorithmic_gflop_per_launch": round(g["flops"] / g["launches"] / 1e9, 3),
                    "algorithmic_mb_per_launch": round(g["bytes"] / g["launches"] / 1e6, 2),
                    "share_of_step_time": round(g["ms"] / (dt * 1e3), 3)}
        # largest HBM-bound kernel family (Winograd transforms, pools, ...) against the HBM roof
        hb = {k: v for k, v in prof.items() if v["flops"] == 0 and v["bytes"] > 0 and v["ms"] > 0}
        hdom = max(hb, key=lambda k: hb[k]["ms"]) if hb else None
        hbm_roof = None
        if hdom:
            g = hb[hdom]
            gbs = g["bytes"] / (g["ms"] * 1e-3) / 1e9
            hbm_roof = {"kernel_group": hdom, "bound": "hbm", "achieved": round(gbs, 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                        "frac": round(gbs / PEAK_HBM_GBS, 4), "launches": g["launches"],
                        "algorithmic_mb_per_launch": round(g["bytes"] / g["launches"] / 1e6, 2),
                        "share_of_step_time": round(g["ms"] / (dt * 1e3), 3)}
        out = {
            "metric": "training images/sec at 1024x512 bs16" if args.mode == "train" else "inference images/sec at 1024x512",
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32" if args.precision == "fp32" else "bf16_fc+f32", "data": "synthetic",
            "config": {"workload": "FCN-8s (VGG-16, fc6 7x7, 20 classes) %s step, %dx%d, %d images/GPU, %s, keep_prob 0.5"
                                   % (args.mode, W, H, N, "TF-Adam" if args.optimizer == "adam" else "SGD+momentum"),
                       "global_batch": N * world, "parallelism": "dp%d" % world},
            # direct-convolution flop count per second; the Winograd layers execute 2.25-4x fewer
            # multiplies than that count, so this "effective" rate may exceed the matrix-core peak
            "effective_tflops_direct_conv_count": round(gflop_img * N * world * args.steps / dt / 1e3, 2),
            "fp32_direct_conv_ceiling_images_per_sec_per_gpu": round(PEAK_F32_MFMA_TFLOPS * 1e3 / gflop_img, 1),
            "roofline": roof,
            "roofline_hbm": hbm_roof,
            "kernel_groups_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
            "kernel_groups_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in prof.items() if v["flops"] > 0 and v["ms"] > 0},
            "kernel_groups_gbs": {k: round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) for k, v in prof.items() if v["flops"] == 0 and v["ms"] > 0},
            "final_loss": loss,
        }
        if world == 1 and not args.no_cpu_baseline:
            try:
                out["cpu_baseline"] = cpu_baseline(H, W, optimizer=args.optimizer)
            except Exception as ex:  # the oracle is only a reported baseline
                out["cpu_baseline"] = {"error": repr(ex)}
        print(json.dumps(out), flush=True)
    eng.close()
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
