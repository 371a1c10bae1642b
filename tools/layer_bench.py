#!/usr/bin/env python3
"""Per-layer kernel times of one training step (in-library HIP events, detail mode)."""
import argparse, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=16); ap.add_argument("--height", type=int, default=512)
    ap.add_argument("--width", type=int, default=1024); ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--precision", default="fp32", choices=["fp32", "bf16_fc", "f32x3", "bf16_fwd", "f32x2", "bf16_fwd_x2", "bf16_train"])
    ap.add_argument("--option", action="append", default=[], metavar="KEY=VALUE")
    ap.add_argument("--infer", action="store_true", help="predict() (frozen parameters, argmax) instead of a training step")
    args = ap.parse_args()
    import torch
    from fcn8s_tensorflow_amd.engine import Engine
    e = Engine(20, precision=args.precision, options={kv.split("=")[0]: int(kv.split("=")[1]) for kv in args.option}); e.init_params(0)
    rng = np.random.default_rng(0)
    img = torch.from_numpy(rng.integers(0, 256, (args.batch, args.height, args.width, 3), dtype=np.uint8)).cuda()
    lab = torch.from_numpy(rng.integers(0, 20, (args.batch, args.height, args.width), dtype=np.uint8)).cuda()
    if args.infer:
        e.freeze(True)
        step = lambda: e.predict(img, argmax=True)
    else:
        step = lambda: e.train_step(img, lab, 1e-4, fetch_loss=False)
    step(); step(); torch.cuda.synchronize()
    e.profile(2); e.profile_reset()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    tot = 0
    for k, v in e.profile_results().items():
        if k.startswith("kernel:"):
            continue
        ms = v["ms"] / args.steps; tot += ms
        extra = "%7.1f TF/s" % (v["flops"] / v["ms"] / 1e9) if v["flops"] else "%7.1f GB/s" % (v["bytes"] / v["ms"] / 1e6)
        print("%-34s %8.3f ms  %s" % (k, ms, extra))
    print("total %.2f ms/step -> %.1f img/s" % (tot, args.batch / tot * 1e3))

if __name__ == "__main__":
    main()
