#!/usr/bin/env python3
"""FCN-8s training throughput on MI355X (BASELINE.json metric).

One "step" = one full training step of the hot path (forward, softmax-CE loss,
backward through every VGG + decoder variable, optimizer update: SGD+momentum by
default as BASELINE.json config 3 names, `--optimizer adam` = the reference's
TF-Adam) on one synthetic 1024x512 batch of 16 images per GPU, inputs already
resident in HBM.  `python bench.py --gpus N --steps K --warmup W` prints ONE JSON
line on rank 0.  For N > 1 it is either launched under torchrun (RANK / WORLD_SIZE
in the environment) or, called plainly, re-launches itself as N ranks through
`python -m torch.distributed.run`: one process per GPU, gradients all-reduced over
RCCL/xGMI in four buckets that overlap with the backward pass (weak scaling:
16 images per GPU).  `--mode e2e` drives FCN8s.train() from BatchGenerator over
generated PNG files instead (host feeder + H2D included), `--mode infer` the
serving loop.
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
PEAK_BF16_MFMA_TFLOPS = 2500.0           # dense bf16 MFMA peak (same guide); the f32x3 mode's roof is a sixth of it


DTYPE_LABEL = {"fp32": "f32", "bf16_fc": "bf16_fc+f32", "f32x3": "f32x3 (fp32 operands as three bf16 pieces on the bf16 MFMA, fp32 accumulate)",
               "bf16_fwd": "bf16 forward (conv3_1..conv5_3, fc6, fc7: bf16-rounded operands, fp32 accumulate) + f32x3 for every other GEMM",
               "f32x2": "f32x2 (fp32 operands as two bf16 pieces = 16 significand bits on the bf16 MFMA, fp32 accumulate; reduced precision)",
               "bf16_fwd_x2": "bf16 forward (conv3_1..conv5_3, fc6, fc7: bf16-rounded operands, fp32 accumulate) + f32x2 (two bf16 pieces per operand) for every other GEMM",
               "bf16_train": "bf16 operands, fp32 accumulate in the forward pass, the data gradients and the weight gradients of conv1_2..conv5_3, fc6, fc7 (direct convolutions, "
                             "fp32 master weights); conv1_1, the decoder, loss and optimizer fp32"}


COMMITTED_NOTE = ("committed profile (profiles/pmc_traffic.json: separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                  "command on an MI355X, tools/collect_profiles.sh) -- not measured in this run (`--live-traffic` re-measures it; the default one-GPU training run does)")
LIVE_NOTE = "live (two rocprofv3 --pmc passes of this command, 2 steps each, run by bench.py after its timed regions)"

# kernel symbols of the HBM-bound groups (the library's HIP-event groups are keyed by what a launch is FOR; the PMC passes see symbols)
GROUP_SYMBOLS = {"wino_transform": r"fcn8s::wino_(?!gemm)\w+_kernel", "maxpool_fwd": r"fcn8s::maxpool_fwd", "maxpool_bwd": r"fcn8s::maxpool_bwd",
                 "softmax_xent": r"fcn8s::softmax_xent", "adam": r"fcn8s::tf_adam_kernel", "sgd_momentum": r"fcn8s::sgd_momentum_kernel"}


def committed_pmc():
    """The committed rocprofv3 PMC summary of this command (profiles/pmc_traffic.json, tools/pmc_summary.py); None if not collected."""
    try:
        return json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
    except Exception:
        return None


def kernel_traffic(d, kernel, note):
    """HBM bytes per launch of one kernel symbol out of a PMC summary."""
    if not d or "kernels" not in d:
        return None
    for name, v in d["kernels"].items():
        n = name.replace(" ", "")
        if n.startswith("voidfcn8s::" + kernel.replace(" ", "")) or n.startswith("fcn8s::" + kernel.replace(" ", "")):
            out = {"hbm_mb_per_launch": v["hbm_mb_per_launch"], "fetch_mb": v["fetch_mb_per_launch"], "write_mb": v["write_mb_per_launch"],
                   "source": d.get("source"), "traffic_source": note}
            if "fetch_mb_per_launch_uncorrected" in v:      # the x2 FETCH_SIZE correction is an upper bound for 64-byte row-segment loads
                out["hbm_mb_per_launch_lower_bound"] = round(v["fetch_mb_per_launch_uncorrected"] + v["write_mb_per_launch"], 3)
            return out
    return None


def group_traffic(d, group, note):
    """HBM megabytes per forward + backward pass of every kernel symbol of an HBM-bound group, out of a PMC summary.  The profiled command runs
    timed steps, a profiled pass and gradient checks: the number of passes it contains is the launch count of the loss kernel (one per pass)."""
    import re
    pat = GROUP_SYMBOLS.get(group)
    if not d or "kernels" not in d or not pat:
        return None
    passes = sum(v["launches"] for k, v in d["kernels"].items() if "softmax_xent_kernel" in k)
    if group in ("adam", "sgd_momentum"):       # (once per optimizer step, not per pass)
        passes = sum(v["launches"] for k, v in d["kernels"].items() if re.search(pat, k))
    if not passes:
        return None
    tot = lo = 0.0; syms = 0
    for k, v in d["kernels"].items():
        if re.search(pat, k):
            tot += v["hbm_mb_per_launch"] * v["launches"]; syms += 1
            lo += (v.get("fetch_mb_per_launch_uncorrected", v["fetch_mb_per_launch"]) + v["write_mb_per_launch"]) * v["launches"]
    if not syms:
        return None
    return {"hbm_mb_per_pass": round(tot / passes, 1), "hbm_mb_per_pass_lower_bound": round(lo / passes, 1), "kernel_symbols": syms, "passes_in_profile": passes,
            "traffic_source": note}


def live_pmc(argv):
    """Re-measure HBM traffic now: two rocprofv3 PMC passes (FETCH_SIZE; WRITE_SIZE; --kernel-trace only, as the microarch guide prescribes)
    over a 2-step run of this same command, summarised by tools/pmc_summary.py.  About a minute; runs after the timed regions, so it cannot
    disturb them.  Returns the summary (kernel symbol -> bytes per launch), {"error": ...} or None; any failure falls back to the committed profile."""
    import shutil
    import subprocess
    import tempfile
    if not shutil.which("rocprofv3"):
        return None
    tmp = tempfile.mkdtemp(prefix="fcn8s_pmc_")
    env = dict(os.environ, TMPDIR="/tmp")
    keep = [a for a in argv if a not in ("--live-traffic",)]
    base = [sys.executable, os.path.abspath(__file__)] + keep + ["--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--repeats", "1", "--no-live-traffic"]
    try:
        for ctr, d in (("FETCH_SIZE", "f"), ("WRITE_SIZE", "w")):
            subprocess.run(["rocprofv3", "--pmc", ctr, "--kernel-trace", "--output-format", "csv", "-d", os.path.join(tmp, d), "-o", "b", "--"] + base,
                           env=env, cwd="/tmp", stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=180, check=True)
        import glob
        f = glob.glob(os.path.join(tmp, "f", "**", "b_counter_collection.csv"), recursive=True)[0]
        w = glob.glob(os.path.join(tmp, "w", "**", "b_counter_collection.csv"), recursive=True)[0]
        out = os.path.join(tmp, "t.json")
        subprocess.run([sys.executable, os.path.join(ROOT, "tools", "pmc_summary.py"), f, w, out, "live: rocprofv3 --pmc passes inside this bench run"],
                       stdout=subprocess.DEVNULL, check=True)
        return json.load(open(out))
    except Exception as ex:
        return {"error": repr(ex)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def group_roof(group, v, precision):
    """a kernel group is bound by whichever roof it sits closer to -- max(algorithmic bytes / t / 8 TB/s, algorithmic flops / t / matrix peak of the
    arithmetic it runs in).  `v` = the library's HIP-event record of the group (ms, flops, bytes)."""
    if v["ms"] <= 0:
        return None
    hbm = v["bytes"] / (v["ms"] * 1e-3) / 1e9 / PEAK_HBM_GBS if v["bytes"] > 0 else 0.0
    if group.split(":")[0].endswith("_bf16"):
        peak = PEAK_BF16_MFMA_TFLOPS
    elif precision in ("f32x3", "bf16_fwd") and group.startswith(("wino_gemm", "fc7_", "tconv_")):
        peak = PEAK_BF16_MFMA_TFLOPS / 6.0
    elif precision in ("f32x2", "bf16_fwd_x2") and group.startswith(("wino_gemm", "fc7_", "tconv_")):
        peak = PEAK_BF16_MFMA_TFLOPS / 3.0
    else:
        peak = PEAK_F32_MFMA_TFLOPS
    mf = v["flops"] / (v["ms"] * 1e-3) / 1e12 / peak if v["flops"] > 0 else 0.0
    return {"bound": "hbm" if hbm >= mf else "mfma", "frac": round(max(hbm, mf), 4), "hbm_frac": round(hbm, 4), "mfma_frac": round(mf, 4), "mfma_peak_tflops": round(peak, 1)}


def _median(xs):
    xs = sorted(xs)
    n = len(xs)
    return xs[n // 2] if n % 2 else 0.5 * (xs[n // 2 - 1] + xs[n // 2])


def physical_cores():
    """Physical cores of this host (distinct thread-sibling sets in sysfs); os.cpu_count() if sysfs does not say."""
    try:
        import glob
        sib = {open(f).read().strip() for f in glob.glob("/sys/devices/system/cpu/cpu[0-9]*/topology/thread_siblings_list")}
        return len(sib) or (os.cpu_count() or 1)
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline_full(h, w, optimizer="sgd"):
    """SURVEY 8d's CPU baseline to the letter (the default `cpu_baseline` of the one-GPU training line since round 5; ~100 s): the oracle on
    ALL physical cores of this host (count printed), the bs2 training step (fwd + bwd + optimizer) at the bench resolution 3 warm-up + 10
    timed runs, median; beside it (c1) 256x256 forward + argmax and the bs1 forward, 1 + 3 runs each."""
    import torch
    from oracle import fcn8s_oracle as orc
    cores = physical_cores()
    torch.set_num_threads(cores)
    P = orc.init_params(20, seed=0)
    t_start = time.perf_counter()

    def timed(fn, warm=3, runs=10):
        for _ in range(warm):
            fn()
        ts = []
        for _ in range(runs):
            a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
        return ts

    img1 = np.random.default_rng(7).integers(0, 256, (1, 256, 256, 3), dtype=np.uint8)
    t_c1 = timed(lambda: np.argmax(orc.softmax(orc.forward(P, img1)), -1), warm=1, runs=3)      # (the two forward legs are side figures: 1 + 3 runs;
    img, lab = orc.synthetic_batch(2, h, w)                                                       #  the training leg below is SURVEY 8d's 3 + 10)
    t_c2 = timed(lambda: orc.forward(P, img[:1]), warm=1, runs=3)
    onehot = orc.one_hot(lab, 20).astype(np.float32)
    state = {"P": P, "m": {k: np.zeros_like(v) for k, v in P.items()}, "v": {k: np.zeros_like(v) for k, v in P.items()}, "t": 0}

    def train_step():
        Pc, m, v_ = state["P"], state["m"], state["v"]
        _, g, _ = orc.loss_and_grads(Pc, img, onehot)
        state["t"] += 1
        for k in Pc:
            if optimizer == "adam":
                Pc[k], m[k], v_[k] = orc.tf_adam_step(Pc[k], g[k], m[k], v_[k], state["t"], 1e-4)
            else:
                Pc[k], m[k] = orc.sgd_momentum_step(Pc[k], g[k], m[k], 1e-4)

    t_c3 = timed(train_step)
    med = _median(t_c3)
    return {"value": round(2.0 / med, 4), "unit": "images/sec", "cores": cores, "host_threads_available": os.cpu_count(), "kind": "port",
            "sample": "SURVEY 8d to the letter: bs2 training step (fwd+bwd+%s) at %dx%d on torch-CPU fp32, all %d physical cores, 3 warm-up + 10 timed, "
                      "median %.2f s (CPU restatement of the reference graph; TF1 unavailable)" % ("TF-Adam" if optimizer == "adam" else "SGD+momentum", w, h, cores, med),
            "train_step_s": [round(t, 2) for t in t_c3],
            "c1_256x256_fwd_argmax": {"median_s": round(_median(t_c1), 3), "runs": len(t_c1), "images_per_sec": round(1.0 / _median(t_c1), 3)},
            "fwd_%dx%d_bs1" % (w, h): {"median_s": round(_median(t_c2), 3), "runs": len(t_c2), "images_per_sec": round(1.0 / _median(t_c2), 3)},
            "total_s": round(time.perf_counter() - t_start, 1)}


def cpu_baseline(h, w, seconds_budget=40.0, optimizer="sgd"):
    """CPU restatement of the reference graph (oracle, kind 'port'), timed on this host's cores as BASELINE.md section 3 /
    SURVEY 8d prescribe: (c1) one 256x256 image forward + argmax, (c2) one 1024x512 image forward, (c3) bs1 training steps
    (fwd + bwd + the same optimizer as the GPU run) at the bench resolution -- each leg 1 warm-up + up to 3 timed runs,
    median reported; `value` is the training leg.  TF1 itself is not installable here."""
    import torch
    from oracle import fcn8s_oracle as orc
    ncpu = os.cpu_count() or 1
    P = orc.init_params(20, seed=0)
    t_start = time.perf_counter()
    img1 = np.random.default_rng(7).integers(0, 256, (1, 256, 256, 3), dtype=np.uint8)      # c1 = one 256x256 image, seed 7
    # thread count: torch-CPU on one image does not scale to every hardware thread of a large host (256 threads made the 256x256
    # forward pass take 10 s); pick the fastest of a few counts on the small leg and use it for all legs -- `cores` reports it
    cands = sorted({c for c in (ncpu // 2, 64, 32, 16) if 1 <= c <= ncpu}, reverse=True)
    best = (None, 1e30)
    for c in cands:
        torch.set_num_threads(c)
        orc.forward(P, img1)
        a = time.perf_counter(); orc.forward(P, img1); d = time.perf_counter() - a
        if d < best[1]:
            best = (c, d)
        if time.perf_counter() - t_start > 12.0:
            break
    cores = best[0]
    torch.set_num_threads(cores)

    def timed(fn, max_runs=3, leg_budget=30.0):
        fn()                                   # warm-up (thread pool, oneDNN primitive caches, page faults)
        ts, t0 = [], time.perf_counter()
        while len(ts) < max_runs and (not ts or time.perf_counter() - t0 + ts[-1] < leg_budget):
            a = time.perf_counter(); fn(); ts.append(time.perf_counter() - a)
        return ts

    t_c1 = timed(lambda: np.argmax(orc.softmax(orc.forward(P, img1)), -1), leg_budget=4.0)
    img, lab = orc.synthetic_batch(1, h, w)
    t_c2 = timed(lambda: orc.forward(P, img), max_runs=2, leg_budget=6.0)
    onehot = orc.one_hot(lab, 20).astype(np.float32)
    state = {"P": P, "m": {k: np.zeros_like(v) for k, v in P.items()}, "v": {k: np.zeros_like(v) for k, v in P.items()}, "t": 0}

    def train_step():
        Pc, m, v_ = state["P"], state["m"], state["v"]
        _, g, _ = orc.loss_and_grads(Pc, img, onehot)
        state["t"] += 1
        for k in Pc:
            if optimizer == "adam":
                Pc[k], m[k], v_[k] = orc.tf_adam_step(Pc[k], g[k], m[k], v_[k], state["t"], 1e-4)
            else:
                Pc[k], m[k] = orc.sgd_momentum_step(Pc[k], g[k], m[k], 1e-4)

    left = seconds_budget - (time.perf_counter() - t_start)
    t_c3 = timed(train_step, leg_budget=max(left * 0.78, 1.0))
    med = _median(t_c3)
    return {"value": round(1.0 / med, 4), "unit": "images/sec", "cores": cores, "host_threads_available": ncpu, "kind": "port",
            "sample": "bs1 training step (fwd+bwd+%s) of one %dx%d image on torch-CPU fp32: 1 warm-up + %d timed, median %.1f s "
                      "(CPU restatement of the reference graph; TF1 unavailable).  Deviations from SURVEY 8d, which asks for a bs2 step, "
                      "3 warm-up + 10 timed, all physical cores: bs1 and 1 + <= 3 runs keep the leg inside the ~30 s budget of a default bench "
                      "run (images/s does not depend on the batch size on the CPU), and %d of %d hardware threads because that count was "
                      "the fastest of %s on the 256x256 leg (one-image oneDNN convolutions slow down beyond it)"
                      % ("TF-Adam" if optimizer == "adam" else "SGD+momentum", w, h, len(t_c3), med, cores, ncpu, cands),
            "train_step_s": [round(t, 2) for t in t_c3],
            "c1_256x256_fwd_argmax": {"median_s": round(_median(t_c1), 3), "runs": len(t_c1), "images_per_sec": round(1.0 / _median(t_c1), 3)},
            "fwd_%dx%d_bs1" % (w, h): {"median_s": round(_median(t_c2), 3), "runs": len(t_c2), "images_per_sec": round(1.0 / _median(t_c2), 3)},
            "total_s": round(time.perf_counter() - t_start, 1)}


def metric_name(args):
    """BASELINE.json's metric string for its own configuration; any other size / batch / arithmetic says so in the metric itself."""
    W, H, N = args.width, args.height, args.batch
    return (("training images/sec at %dx%d bs%d" % (W, H, N) if args.mode == "train" else
             "inference images/sec at %dx%d" % (W, H) + ("" if N == 1 else " bs%d" % N)) + ("" if args.precision == "fp32" else " (%s arithmetic)" % args.precision))


def error_line(args, what, world=None):
    """The ONE JSON line of a run that did not finish: same identifying keys, `value` null and an `error` field saying why --
    so that a dead or hung rank gives the driver a record instead of a hang or an empty stdout."""
    world = world or int(os.environ.get("WORLD_SIZE", str(args.gpus)))
    return {"metric": metric_name(args), "value": None, "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": None, "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": DTYPE_LABEL[args.precision],
            "data": "synthetic", "config": {"workload": "FCN-8s %s step, %dx%d, %d images/GPU" % (args.mode, args.width, args.height, args.batch),
                                            "global_batch": args.batch * world, "parallelism": "dp%d" % world},
            "error": what}


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no torchrun environment: become the launcher of N ranks (one per GPU) and
    pass rank 0's JSON line through."""
    import socket
    import subprocess
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + argv
    # only the JSON line goes to stdout; whatever else the ranks print there (gloo's connection banner, ...) is moved to stderr
    p = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    seen = False
    for line in p.stdout:
        is_json = line.lstrip().startswith("{")
        seen = seen or is_json
        (sys.stdout if is_json else sys.stderr).write(line)
        sys.stdout.flush()
    rc = p.wait()
    if not seen:                                  # every rank died before rank 0 could speak (or rank 0 was killed outright): still ONE line
        print(json.dumps(error_line(args, "no rank printed a result; launcher exit code %d" % rc, world=args.gpus)), flush=True)
    return rc


def make_png_dataset(root, n, h, w, num_classes=20, seed=0):
    """n (image, label) PNG pairs in the Cityscapes directory convention the reference's BatchGenerator pairs by
    (data_generator/batch_generator.py:101-119): <root>/images/<city>/<name>_leftImg8bit.png, <root>/gt/<city>/<name>_gtFine_labelIds.png.
    Smooth structure + noise so that the PNGs compress (and decode) like photographs rather than like white noise."""
    from PIL import Image
    rng = np.random.default_rng(seed)
    img_dir, gt_dir = os.path.join(root, "images", "city"), os.path.join(root, "gt", "city")
    os.makedirs(img_dir, exist_ok=True); os.makedirs(gt_dir, exist_ok=True)
    yy, xx = np.mgrid[0:h, 0:w]
    for i in range(n):
        base = [127 + 90 * np.sin(xx / rng.uniform(20, 120) + rng.uniform(0, 6)) * np.cos(yy / rng.uniform(20, 120) + rng.uniform(0, 6)) for _ in range(3)]
        img = np.clip(np.stack(base, -1) + rng.normal(0, 6, (h, w, 3)), 0, 255).astype(np.uint8)
        lab = ((yy // 64) * 7 + (xx // 64) * 3 + i) % num_classes
        Image.fromarray(img).save(os.path.join(img_dir, "s%04d_leftImg8bit.png" % i))
        Image.fromarray(lab.astype(np.uint8)).save(os.path.join(gt_dir, "s%04d_gtFine_labelIds.png" % i))
    return os.path.join(root, "images"), os.path.join(root, "gt")


def e2e(args, emit=True):
    """FCN8s.train() end to end: PNG files -> BatchGenerator (decode, flip augmentation, `workers` processes) -> staging slots
    (pinned copy + H2D on the copy stream, one batch ahead) -> the training step, loss fetched every step as the reference does
    (fcn8s_tensorflow.py:551-578).  Prints one JSON line with the end-to-end rate next to the resident-input rate of the same
    process."""
    import tempfile
    import torch
    from fcn8s_tensorflow_amd.fcn8s import FCN8s
    from fcn8s_tensorflow_amd.batch_generator import BatchGenerator
    N, H, W = args.batch, args.height, args.width
    root = tempfile.mkdtemp(prefix="fcn8s_e2e_")
    t0 = time.perf_counter()
    img_root, gt_root = make_png_dataset(root, 4 * N, H, W)
    t_data = time.perf_counter() - t0
    gen = BatchGenerator(image_dirs=[img_root], image_file_extension='png', ground_truth_dirs=[gt_root],
                         image_name_split_separator='_leftImg8bit', ground_truth_suffix='_gtFine_labelIds',
                         check_existence=True, num_classes=20)
    train_gen = gen.generate(batch_size=N, convert_to_one_hot=True, flip=0.5, shuffle=True, workers=args.workers)
    # feeder alone: batches per second the host side can deliver (class-id form, what FCN8s.train pulls)
    train_gen.next_ids()
    tf0 = time.perf_counter()
    for _ in range(4):
        train_gen.next_ids()
    feeder_ips = 4 * N / (time.perf_counter() - tf0)
    model = FCN8s(vgg16_dir='synthetic:0', num_classes=20, device_id=args.device or 0)
    if args.precision != "fp32":
        model.engine.set_precision(args.precision)
    sched = lambda step: 1e-4
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        model.train(train_gen, epochs=1, steps_per_epoch=args.warmup, learning_rate_schedule=sched, record_summaries=False)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        model.train(train_gen, epochs=1, steps_per_epoch=args.steps, learning_rate_schedule=sched, record_summaries=False)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
    # the same step on device-resident inputs (what `--mode train` times), TF-Adam like FCN8s.train
    rng = np.random.default_rng(0)
    images = torch.from_numpy(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).cuda()
    labels = torch.from_numpy(rng.integers(0, 20, (N, H, W), dtype=np.uint8)).cuda()
    eng = model.engine
    for _ in range(2):
        eng.train_step(images, labels, 1e-4, fetch_loss=False)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        eng.train_step(images, labels, 1e-4, fetch_loss=False)
    torch.cuda.synchronize()
    dres = time.perf_counter() - t2
    value, resident = N * args.steps / dt, N * args.steps / dres
    out = {"metric": "training images/sec at %dx%d bs%d, end to end (PNG decode + augmentation + H2D + FCN8s.train)" % (W, H, N)
                     + ("" if args.precision == "fp32" else " (%s arithmetic)" % args.precision),
           "value": round(value, 3), "unit": "images/sec", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
           "dtype": DTYPE_LABEL[args.precision], "data": "synthetic PNG files (%d pairs, generated in %.1f s)" % (4 * N, t_data),
           "config": {"workload": "FCN8s.train() from BatchGenerator(workers=%d, flip 0.5, one-hot contract) over %dx%d PNG pairs, %d images/step, TF-Adam, loss fetched every step"
                                  % (args.workers, W, H, N), "global_batch": N, "parallelism": "dp1"},
           "resident_input_images_per_sec": round(resident, 3), "e2e_over_resident": round(value / resident, 4),
           "feeder_alone_images_per_sec": round(feeder_ips, 1), "host_cores": os.cpu_count()}
    train_gen.close()
    with contextlib.redirect_stdout(io.StringIO()):
        model.close()
    import shutil
    shutil.rmtree(root, ignore_errors=True)
    if emit:
        print(json.dumps(out), flush=True)
    return out


def secondary_legs(args, dev, budget_s=75.0):
    """What earlier rounds claimed from builder-run profiles, now inside the driver-timed default line: short legs, run
    AFTER the headline regions on engines of their own, each 2 warm-up + a few timed steps between device synchronisations, inputs resident:
    the f32x3 arithmetic at config 3's shape, config 5's per-GPU shape (2048x1024, 4 images) in fp32 and in its own arithmetic (bf16 forward,
    and the bf16_train mode), batch-1 inference (config 2), and ten steps of FCN8s.train() end to end from PNG files.  Never the headline."""
    import argparse as _ap
    import torch
    from fcn8s_tensorflow_amd import _lib as L
    from fcn8s_tensorflow_amd.engine import Engine
    t_start = time.perf_counter()
    out = {"note": "side figures of this same run (after the timed regions; 2 warm-up + `steps` timed steps each, resident synthetic inputs); not the headline metric"}
    opt = L.OPT_TF_ADAM if args.optimizer == "adam" else L.OPT_SGD_MOMENTUM

    def train_leg(precision, N, H, W, steps):
        e = Engine(20, device_id=dev, seed=1234, precision=precision)
        try:
            e.init_params(seed=0)
            rng = np.random.default_rng(1234)
            images = torch.from_numpy(rng.integers(0, 256, (N, H, W, 3), dtype=np.uint8)).cuda()
            labels = torch.from_numpy(rng.integers(0, 20, (N, H, W), dtype=np.uint8)).cuda()
            for _ in range(2):
                e.train_step(images, labels, 1e-4, keep_prob=0.5, optimizer=opt, fetch_loss=False)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                e.train_step(images, labels, 1e-4, keep_prob=0.5, optimizer=opt, fetch_loss=False)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            loss = e.forward_backward(images, labels, keep_prob=1.0)
            return {"images_per_sec": round(N * steps / dt, 2), "ms_per_step": round(dt / steps * 1e3, 3), "steps": steps, "dtype": DTYPE_LABEL[precision],
                    "workload": "train step %dx%d, %d images" % (W, H, N), "final_loss": round(float(loss), 5)}
        finally:
            e.close()

    def infer_leg(steps, precision="fp32", N=1):
        e = Engine(20, device_id=dev, seed=1234, precision=precision)
        try:
            e.init_params(seed=0)
            e.freeze(True)
            images = torch.from_numpy(np.random.default_rng(1234).integers(0, 256, (N, 512, 1024, 3), dtype=np.uint8)).cuda()
            for _ in range(5):
                e.predict(images, argmax=True)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                e.predict(images, argmax=True)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            return {"images_per_sec": round(N * steps / dt, 1), "ms_per_image": round(dt / steps / N * 1e3, 4), "ms_per_batch": round(dt / steps * 1e3, 4), "steps": steps,
                    "dtype": "f32" if precision == "fp32" else DTYPE_LABEL[precision],
                    "workload": "predict (argmax) 1024x512, %d image%s, frozen parameters%s" % (N, "" if N == 1 else "s", " (BASELINE config 2)" if (N == 1 and precision == "fp32") else "")}
        finally:
            e.close()

    def e2e_leg():
        a = _ap.Namespace(**vars(args))
        a.steps, a.warmup, a.workers, a.precision, a.device = 10, 2, min(32, max(2, (os.cpu_count() or 4) // 4)), "fp32", dev
        r = e2e(a, emit=False)
        return {k: r[k] for k in ("value", "ms_per_step", "steps", "resident_input_images_per_sec", "e2e_over_resident", "feeder_alone_images_per_sec", "data")} | {
            "workload": r["config"]["workload"]}

    legs = [("c3_f32x3", lambda: train_leg("f32x3", args.batch, args.height, args.width, 8)),
            ("c5_shape_fp32", lambda: train_leg("fp32", 4, 1024, 2048, 5)),
            ("c5_shape_bf16_fwd", lambda: train_leg("bf16_fwd", 4, 1024, 2048, 5))]
    if "bf16_train" in DTYPE_LABEL:
        legs.append(("c5_shape_bf16_train", lambda: train_leg("bf16_train", 4, 1024, 2048, 5)))
    legs += [("c2_inference_bs1", lambda: infer_leg(100))]
    if "bf16_train" in DTYPE_LABEL:     # serving in config 5's arithmetic: batch 1 and batch 16 (round 6: prediction takes the training pass's data flow)
        legs += [("inference_bs1_bf16_train", lambda: infer_leg(100, "bf16_train", 1)), ("inference_bs16_bf16_train", lambda: infer_leg(20, "bf16_train", 16))]
    legs += [("c3_end_to_end_10_steps", e2e_leg)]
    for name, fn in legs:
        if time.perf_counter() - t_start > budget_s:
            out[name] = {"skipped": "the secondary legs' %.0f s budget was used up" % budget_s}
            continue
        t0 = time.perf_counter()
        try:
            out[name] = fn()
            out[name]["leg_seconds"] = round(time.perf_counter() - t0, 1)
        except Exception as ex:
            out[name] = {"error": repr(ex)}
    out["total_s"] = round(time.perf_counter() - t_start, 1)
    return out


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
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16_fc", "f32x3", "bf16_fwd", "f32x2", "bf16_fwd_x2", "bf16_train"],
                    help="fp32 = the reference's arithmetic (the headline number); bf16_fc = BASELINE config 5's mode (forward "
                         "fc6/fc7 with bf16 operands on the bf16 MFMA, fp32 accumulate) -- reported as dtype 'bf16_fc+f32'")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-full", action="store_true", help="(default since round 5) `cpu_baseline` = SURVEY 8d to the letter: bs2 training step, 3 warm-up + 10 "
                    "timed, all physical cores, ~100 s; the bounded sample of earlier rounds is reported beside it as `cpu_baseline_quick`")
    ap.add_argument("--cpu-baseline-quick", action="store_true", help="only the bounded ~25 s CPU sample (bs1, 1 + 3 runs, fastest thread count) as `cpu_baseline`")
    ap.add_argument("--no-secondary", action="store_true", help="skip the `secondary` legs (f32x3, config 5's shape, batch-1 inference, end to end) of the default one-GPU training line")
    ap.add_argument("--comm", default="torch", choices=["torch", "native"], help="who moves the gradient buckets for --gpus > 1: torch.distributed "
                    "(the default) or the library's own RCCL communicator behind the C ABI (fcn8s_comm_init / fcn8s_allreduce_bucket)")
    ap.add_argument("--mode", default="train", choices=["train", "infer", "e2e"],
                    help="train = the headline step on HBM-resident synthetic batches; infer = serving loop; e2e = FCN8s.train() fed by "
                         "BatchGenerator from generated PNG files (decode + augmentation + H2D inside the timed region)")
    ap.add_argument("--workers", type=int, default=8, help="--mode e2e: BatchGenerator decode workers")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL); 'gloo' lets the "
                    "multi-rank path be exercised on a single-GPU box together with --device")
    ap.add_argument("--device", type=int, default=None, help="HIP device ordinal (default: LOCAL_RANK)")
    ap.add_argument("--repeats", type=int, default=3, help="how many times the timed region of exactly --steps steps is run; value / ms_per_step "
                    "are those of the median region, all regions are listed in `timed_regions_ms_per_step`")
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE", help="library algorithm option (fcn8s_set_option), e.g. winograd_tile=4")
    ap.add_argument("--live-traffic", action="store_true", help="measure roofline.traffic now with two rocprofv3 --pmc passes (about a minute) instead of "
                    "reading the committed profile.  Default: on for the plain one-GPU training run with the CPU baseline (the driver's "
                    "bench line), when rocprofv3 is on PATH and this process is not itself being profiled; off otherwise")
    ap.add_argument("--no-live-traffic", action="store_true", help="never spawn the rocprofv3 passes; roofline.traffic comes from profiles/pmc_traffic.json")
    ap.add_argument("--rank-timeout", type=float, default=600.0, help="seconds a collective (or the rendezvous) may wait for a peer before the process "
                    "group gives up; a rank that dies or hangs then ends the run with an `error` line instead of blocking it forever")
    args = ap.parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(self_launch(args, sys.argv[1:]))
    if args.mode == "e2e":
        return e2e(args)
    rank = int(os.environ.get("RANK", "0"))
    state = {"done": False}

    def on_term(signum, frame):                   # torchrun ends the surviving ranks with SIGTERM when one of them has failed
        if rank == 0 and not state["done"]:
            state["done"] = True
            print(json.dumps(error_line(args, "terminated by the launcher (signal %d): a peer rank failed or timed out" % signum)), flush=True)
        os._exit(1)

    if "WORLD_SIZE" in os.environ:
        import signal
        import threading
        signal.signal(signal.SIGTERM, on_term)
        # the Python-level handler only runs when the main thread is back in the interpreter; a rank blocked inside a collective or a
        # device synchronisation never is.  The C-level handler still writes the signal number to the wake-up pipe at once: a watcher
        # thread reads it there and speaks for the main thread.
        rfd, wfd = os.pipe()
        os.set_blocking(wfd, False)
        signal.set_wakeup_fd(wfd, warn_on_full_buffer=False)

        def watch():
            while True:
                b = os.read(rfd, 1)
                if b and b[0] == signal.SIGTERM:
                    on_term(signal.SIGTERM, None)

        threading.Thread(target=watch, daemon=True).start()
    try:
        run(args, state)
    except BaseException as ex:                   # (SystemExit / KeyboardInterrupt included: the line is the record either way)
        if rank == 0 and not state["done"]:
            state["done"] = True
            print(json.dumps(error_line(args, "rank 0: %r" % (ex,))), flush=True)
        if isinstance(ex, Exception):
            import traceback
            traceback.print_exc()
            sys.stdout.flush(); sys.stderr.flush()
            os._exit(1)                           # not sys.exit: a process group whose peer is gone can block in its destructor
        raise


def run(args, state):
    import torch
    import torch.distributed as dist
    from fcn8s_tensorflow_amd import _lib as L
    from fcn8s_tensorflow_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dev = local_rank if args.device is None else args.device
    under_launcher = "WORLD_SIZE" in os.environ and "RANK" in os.environ      # torchrun, even with one rank: RCCL is initialised and used
    numa = None
    if under_launcher and world > 1:              # each rank (and every thread / process it starts from here on) next to its GPU's memory controller:
        from fcn8s_tensorflow_amd.dp import bind_to_gpu_numa      # before the process group and the engine exist, so that their threads inherit the mask
        numa = bind_to_gpu_numa(dev, int(os.environ.get("LOCAL_WORLD_SIZE", world)) if args.device is None else 1)
    if under_launcher:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        torch.cuda.set_device(dev)
        import datetime
        os.environ.setdefault("TORCH_NCCL_ASYNC_ERROR_HANDLING", "1")       # a failed / timed-out collective tears the process down instead of hanging it
        tmo = datetime.timedelta(seconds=args.rank_timeout)
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev), timeout=tmo)
        else:
            dist.init_process_group(args.backend, timeout=tmo)

    trace = os.environ.get("FCN8S_BENCH_TRACE")
    def mark(msg):
        if trace:
            print("[bench rank %d] %s" % (rank, msg), file=sys.stderr, flush=True)
    N, H, W = args.batch, args.height, args.width
    mark("process group up; creating engine")
    options = {kv.split("=", 1)[0]: int(kv.split("=", 1)[1]) for kv in args.option}
    eng = Engine(20, device_id=dev, seed=1234 + rank, precision=args.precision, options=options)
    mark("numa: %s" % numa)
    if under_launcher and args.comm == "native":
        eng.comm_init_native()                    # the 128-byte id travels through the torch group once; the collectives are the library's
    eng.dp_always = under_launcher                # a one-rank process group still runs the bucketed all-reduces (RCCL with one rank)
    eng.replica_check_every = 0                   # the replica guard of FCN8s.train stays out of the measurement (and the local-only leg below lets replicas drift on purpose)
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
        if under_launcher:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step()
        mark("warmup step %d enqueued" % i)
    if os.environ.get("FCN8S_BENCH_FAIL_RANK") == str(rank):      # test hook: this rank dies after its warm-up steps (tests/test_multigpu_gpu.py)
        raise RuntimeError("FCN8S_BENCH_FAIL_RANK: rank %d fails on purpose" % rank)
    fence()
    mark("warmup done")
    # ---- the timed region: exactly K steps between two fences (barrier + device sync), no in-library event recording; the MAX over
    # ranks is the region's time.  The region is run --repeats times; the median region is the one reported (all are listed).
    regions, per_rank_regions = [], []
    for _ in range(max(1, args.repeats)):
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt_local = time.perf_counter() - t0
        if under_launcher:
            allt = [torch.zeros(1, dtype=torch.float64, device="cuda") for _ in range(world)]
            dist.all_gather(allt, torch.tensor([dt_local], dtype=torch.float64, device="cuda"))
            per_rank = [float(x.item()) for x in allt]
        else:
            per_rank = [dt_local]
        regions.append(max(per_rank)); per_rank_regions.append(per_rank)
    mark("timed regions done")
    order = sorted(range(len(regions)), key=lambda i: regions[i])
    mid = order[(len(order) - 1) // 2]
    dt = regions[mid]
    per_rank_ms = [round(t / args.steps * 1e3, 3) for t in per_rank_regions[mid]]
    # ---- a second pass over the same steps with the library's HIP events on (recorded on the launch stream, one pair per
    # kernel launch): per-kernel durations for the roofline block.  Kept out of the timed region (the event pairs cost ~1 %).
    psteps = min(args.steps, 10)
    eng.profile(True)
    eng.profile_reset()
    tp = time.perf_counter()
    for _ in range(psteps):
        step()
    fence()
    dtp = time.perf_counter() - tp
    prof = eng.profile_results()
    eng.profile(False)
    for v in prof.values():                       # normalise to the timed region's step count (the report below divides by args.steps)
        for k in ("ms", "flops", "bytes"):
            v[k] *= args.steps / psteps
        v["launches"] = int(round(v["launches"] * args.steps / psteps))
    # ---- data-parallel runs: what the gradient exchange costs
    comm = None
    if under_launcher and args.mode == "train":
        replicas_ok = None
        if world > 1:
            try:
                replicas_ok = bool(eng.check_replicas())      # every step so far exchanged its gradients: the replicas must hold the same bits
            except RuntimeError as ex:
                replicas_ok = False
                mark("replica check failed: %s" % ex)
        per_bucket = []
        for off, n in eng.buckets:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            view = eng.flat_grads[off:off + n]
            dist.all_reduce(view); torch.cuda.synchronize()
            e0.record()
            for _ in range(3):
                dist.all_reduce(view)
            e1.record(); torch.cuda.synchronize()
            per_bucket.append(round(e0.elapsed_time(e1) / 3, 3))
        eng.flat_grads.zero_()                    # (the standalone runs summed garbage into the buffer; every step rewrites it anyway)
        fence()
        tl = time.perf_counter()
        for _ in range(psteps):                   # the same step without the exchange (each rank alone)
            eng.train_step(images, labels, 1e-4, keep_prob=0.5, l2_rate=0.0, optimizer=opt, fetch_loss=False, reduce=False)
        fence()
        local_ms = (time.perf_counter() - tl) / psteps * 1e3
        tt = torch.tensor([local_ms], dtype=torch.float64, device="cuda")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        local_ms = float(tt.item())
        # ready -> complete timestamps of each bucket's all-reduce inside the step (ms after the step's first kernel, i.e. the forward pass
        # included; "issue" = the moment the bucket's last gradient kernel ended; averaged over psteps)
        eng.comm_trace = []
        fence()
        for _ in range(psteps):
            eng.train_step(images, labels, 1e-4, keep_prob=0.5, l2_rate=0.0, optimizer=opt, fetch_loss=False)
        fence()
        tr, eng.comm_trace = eng.comm_trace, None
        nb = len(eng.buckets)
        issue, done, span = [0.0] * nb, [0.0] * nb, 0.0
        start = None
        for kind, a, b in tr:
            if kind == "step":
                start = a
            elif kind == "end":
                span += start.elapsed_time(a) / psteps
            else:
                issue[kind] += start.elapsed_time(a) / psteps
                done[kind] += start.elapsed_time(b) / psteps
        busy, cur_s, cur_e = 0.0, None, None            # union of the [issue, complete] intervals
        for s_, e_ in sorted(zip(issue, done)):
            if cur_e is None or s_ > cur_e:
                busy += (cur_e - cur_s) if cur_e is not None else 0.0
                cur_s, cur_e = s_, e_
            else:
                cur_e = max(cur_e, e_)
        busy += (cur_e - cur_s) if cur_e is not None else 0.0
        exposed = dt / args.steps * 1e3 - local_ms
        try:
            rccl_version = ".".join(str(x) for x in torch.cuda.nccl.version()) if args.backend == "nccl" else None
        except Exception:
            rccl_version = None
        if rccl_version is None:
            v = eng.comm_info()["rccl_version"]   # the library's own view (dlopen of librccl); 22707 = 2.27.7
            rccl_version = "%d.%d.%d" % (v // 10000, v // 100 % 100, v % 100) if v else None
        ws_ = dist.get_world_size()
        comm = {"backend": args.backend, "collectives_by": "libfcn8s_hip (fcn8s_allreduce_bucket)" if eng.native_comm else "torch.distributed",
                "rccl_ranks": dist.get_world_size() if args.backend == "nccl" else 0, "ranks": dist.get_world_size(),
                "rccl_version": rccl_version, "nccl_env": {k: v for k, v in sorted(os.environ.items()) if k.startswith(("NCCL_", "RCCL_", "HSA_ENABLE_IPC"))},
                "numa": numa, "replicas_identical_after_timed_steps": replicas_ok,
                "bucket_names": ["fc7+decoder", "fc6", "conv4+conv5", "conv1..conv3"][:len(eng.buckets)],
                "bucket_mb": [round(n * 4 / 1e6, 1) for _, n in eng.buckets],
                "allreduce_ms_per_bucket_standalone": per_bucket,
                # bus bandwidth of a ring all-reduce: 2 (n - 1) / n x bytes / time -- the figure to hold against one xGMI link (~153 GB/s)
                "allreduce_gbs_per_bucket_standalone": [round(n * 4 / 1e9 / (t * 1e-3), 1) if t > 0 else None for (_, n), t in zip(eng.buckets, per_bucket)],
                "allreduce_busbw_gbs_per_bucket_standalone": [round(2.0 * (ws_ - 1) / ws_ * n * 4 / 1e9 / (t * 1e-3), 1) if t > 0 else None
                                                              for (_, n), t in zip(eng.buckets, per_bucket)],
                "local_only_ms_per_step": round(local_ms, 3),
                "exposed_comm_ms_per_step": round(exposed, 3),
                "bucket_issue_ms": [round(x, 3) for x in issue], "bucket_complete_ms": [round(x, 3) for x in done],
                "step_span_ms_traced": round(span, 3), "comm_busy_ms_per_step": round(busy, 3),
                "overlap_frac": (round(max(0.0, min(1.0, 1.0 - max(exposed, 0.0) / busy)), 4) if busy > 0 else None)}
        eng.broadcast_params(0)                   # the local-only steps let the replicas drift; re-align before the final loss
    loss = eng.forward_backward(images, labels, keep_prob=1.0) if args.mode == "train" else None
    fetch_ms, bcheck = None, None
    if args.mode == "train" and world == 1:
        # the same steps with the loss scalar fetched every step, as the reference's sess.run does
        fence()
        tf_ = time.perf_counter()
        for _ in range(args.steps):
            eng.train_step(images, labels, 1e-4, keep_prob=0.5, l2_rate=0.0, optimizer=opt, fetch_loss=True)
        fence()
        fetch_ms = round((time.perf_counter() - tf_) / args.steps * 1e3, 3)
        # is the backward pass the derivative of the forward pass?  Central difference of the loss along the gradient on the batch:
        # (L(theta + eps g) - L(theta - eps g)) / (2 eps |g|^2) must be 1 (tests/test_fullsize_gpu.py does the same at 4 images).
        try:
            keep = eng.flat_params.clone()
            theta = keep
            nb_ = N            # (the whole batch: the check's launches then have the shapes of the timed steps and do not skew per-kernel averages in a profile of this command)
            eng.forward_backward(images[:nb_], labels[:nb_], keep_prob=1.0)
            g = eng.flat_grads.clone()
            norm2 = float((g.double() ** 2).sum())
            ratios = []
            for target in (2e-5, 4e-5, 6e-5):
                eps = target / max(norm2, 1e-30)
                eng.flat_params.copy_(theta - eps * g); lm = eng.forward_backward(images[:nb_], labels[:nb_], keep_prob=1.0)
                eng.flat_params.copy_(theta + eps * g); lp = eng.forward_backward(images[:nb_], labels[:nb_], keep_prob=1.0)
                ratios.append(round((lp - lm) / (2 * target), 4))
            eng.flat_params.copy_(keep)
            bcheck = {"directional_derivative_ratios": ratios, "ok": bool(any(0.97 < r < 1.03 for r in ratios)),
                      "what": "central difference of the loss along its own gradient / |g|^2 at three step sizes (the whole batch, keep_prob 1): 1 = the backward pass is the derivative of the forward pass"}
        except Exception as ex:
            bcheck = {"error": repr(ex)}

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
            profiled = any(k.startswith(("ROCPROF", "ROCP_")) or k == "HSA_TOOLS_LIB" for k in os.environ)
            auto_live = world == 1 and not under_launcher and args.mode == "train" and not args.no_cpu_baseline and not profiled
            pmc = live_pmc(sys.argv[1:]) if (args.live_traffic or auto_live) and not args.no_live_traffic else None
            pmc_note = LIVE_NOTE
            tr = kernel_traffic(pmc, dom, pmc_note) if pmc and "error" not in pmc else None
            if not tr:
                pmc, pmc_note = committed_pmc(), COMMITTED_NOTE
                tr = kernel_traffic(pmc, dom, pmc_note)
            # f32x3 mode: six bf16 MFMA products per fp32-equivalent multiply-add -> peak = bf16 dense peak / 6
            # (f32x2: three products)
            peak = (PEAK_BF16_MFMA_TFLOPS / 6.0 if "_x3_" in dom else PEAK_BF16_MFMA_TFLOPS / 3.0 if "_x2_" in dom
                    else PEAK_BF16_MFMA_TFLOPS if ("conv_bf16" in dom or "wgrad_bf16" in dom) else PEAK_F32_MFMA_TFLOPS)
            roof = {"kernel": dom, "bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1),
                    "unit": "TFLOP/s", "frac": round(ach / peak, 4),
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
                        "algorithmic_mb_per_step": round(g["bytes"] / args.steps / 1e6, 1),
                        "share_of_step_time": round(g["ms"] / (dt * 1e3), 3)}
            # the group's HBM-side bytes from the PMC passes (every kernel symbol of the group, per forward + backward pass) against its algorithmic
            # bytes: well above 1 = re-reads (the backward gathers of the Winograd transforms), below 1 = reads served by the Infinity Cache
            gt = group_traffic(pmc if (dom and pmc and "error" not in pmc) else committed_pmc(), hdom, pmc_note if dom else COMMITTED_NOTE)
            if gt:
                hbm_roof["traffic"] = round(gt["hbm_mb_per_pass"] * 1e6); hbm_roof["traffic_unit"] = "bytes/step (all kernels of the group)"
                hbm_roof["traffic_over_algorithmic"] = round(gt["hbm_mb_per_pass"] * 1e6 / (g["bytes"] / args.steps), 3)
                hbm_roof["traffic_over_algorithmic_lower_bound"] = round(gt["hbm_mb_per_pass_lower_bound"] * 1e6 / (g["bytes"] / args.steps), 3)
                hbm_roof["traffic_detail"] = gt
        out = {
            # BASELINE.json's metric string for its own configuration; any other size / batch / arithmetic says so in the metric itself
            "metric": metric_name(args),
            "value": round(value, 3), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": DTYPE_LABEL[args.precision], "data": "synthetic",
            "config": {"workload": "FCN-8s (VGG-16, fc6 7x7, 20 classes) %s step, %dx%d, %d images/GPU, %s, keep_prob 0.5"
                                   % (args.mode, W, H, N, "TF-Adam" if args.optimizer == "adam" else "SGD+momentum"),
                       "global_batch": N * world, "parallelism": "dp%d" % world},
            # direct-convolution flop count per second; the Winograd layers execute 2.25-4x fewer
            # multiplies than that count, so this "effective" rate may exceed the matrix-core peak
            "effective_tflops_direct_conv_count": round(gflop_img * N * world * args.steps / dt / 1e3, 2),
            "fp32_direct_conv_ceiling_images_per_sec_per_gpu": round(PEAK_F32_MFMA_TFLOPS * 1e3 / gflop_img, 1),
            "roofline": roof,
            "roofline_hbm": hbm_roof,
            "timed_regions_ms_per_step": [round(t / args.steps * 1e3, 3) for t in regions],
            "timed_regions": {"count": len(regions), "reported": "median region", "min_ms_per_step": round(min(regions) / args.steps * 1e3, 3),
                              "max_ms_per_step": round(max(regions) / args.steps * 1e3, 3)},
            "per_rank_ms": per_rank_ms,
            "options": options,
            "profiled_pass": {"steps": psteps, "ms_per_step": round(dtp / psteps * 1e3, 3)},
            "rccl_ranks": dist.get_world_size() if (under_launcher and args.backend == "nccl") else 0,
            "comm": comm,
            "kernel_groups_ms_per_step": {k: round(v["ms"] / args.steps, 3) for k, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])},
            "kernel_groups_tflops": {k: round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) for k, v in prof.items() if v["flops"] > 0 and v["ms"] > 0},
            "kernel_groups_gbs": {k: round(v["bytes"] / (v["ms"] * 1e-3) / 1e9, 1) for k, v in prof.items() if v["bytes"] > 0 and v["ms"] > 0},
            # which roof bounds each group: max(bytes / t / 8 TB/s, flops / t / matrix peak) -- the score heads, conv1_1 and the 4 x 4 transposed
            # convolutions come out HBM-bound, as the survey classes them
            "kernel_groups_roof": {k: group_roof(k, v, args.precision) for k, v in prof.items() if v["ms"] > 0},
            "final_loss": loss,
            # synthetic labels are uniform noise, so nothing can be learned: with the reference's decoder init (sigma 1e-3 / 1e-2,
            # fcn8s_tensorflow.py:159-160) the logits stay ~0 and the loss stays at ln 20 -- a NaN, a diverging or a sign-flipped update
            # shows here; `backward_check` below is what shows a wrong gradient
            "expected_final_loss": round(float(np.log(20.0)), 5) if args.mode == "train" else None,
            "final_loss_ok": (bool(abs(loss - np.log(20.0)) < 2e-3) if (args.mode == "train" and loss is not None) else None),
            "backward_check": bcheck,
            # the timed steps queue the loss kernel but do not copy its scalar to the host (the reference fetches total_loss in every
            # sess.run, fcn8s_tensorflow.py:554-572); the same K steps with the fetch are timed separately, after the regions
            "loss_fetch": "deferred" if args.mode == "train" else None,
            "ms_per_step_with_loss_fetch": fetch_ms,
        }
        default_line = (world == 1 and not under_launcher and args.mode == "train" and not args.no_cpu_baseline
                        and (N, H, W) == (16, 512, 1024) and args.precision == "fp32")
    eng.close()
    if rank == 0:
        if default_line and not args.no_secondary:
            try:
                out["secondary"] = secondary_legs(args, dev)
            except Exception as ex:
                out["secondary"] = {"error": repr(ex)}
        if world == 1 and not args.no_cpu_baseline:
            try:
                if args.cpu_baseline_quick:
                    out["cpu_baseline"] = cpu_baseline(H, W, optimizer=args.optimizer)
                else:
                    out["cpu_baseline"] = cpu_baseline_full(H, W, optimizer=args.optimizer)
                    out["cpu_baseline_quick"] = cpu_baseline(H, W, optimizer=args.optimizer)
            except Exception as ex:  # the oracle is only a reported baseline
                out["cpu_baseline"] = {"error": repr(ex)}
        state["done"] = True
        print(json.dumps(out), flush=True)
    if under_launcher:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
