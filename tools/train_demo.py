#!/usr/bin/env python
"""End-to-end training demonstration on a LEARNABLE synthetic segmentation task (Cityscapes and the pretrained VGG-16 are not
available offline, SURVEY 8c: the reference's mIoU anchors cannot be reproduced).  PNG pairs on disk -> BatchGenerator (flip,
brightness, worker processes) -> FCN8s.train() with TF-Adam, dropout 0.5, evaluation on a held-out set every epoch -> save ->
restore in a second model -> evaluate again (must reproduce the numbers) -> predict.

The task: every image is a noisy background plus 5-9 random rectangles, each filled with the colour of its class (19 object
classes, class 0 = background) plus Gaussian noise; the label map holds the class ids.  A randomly initialised VGG-16 encoder has to
learn colour -> class through the whole FCN-8s path.  Runs on the GPU box:

    python tools/train_demo.py --out gpurun_out/train_demo.json       (then copy to profiles/)
"""
import argparse
import contextlib
import io
import json
import os
import shutil
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def palette(num_classes, seed=0):
    rng = np.random.default_rng(seed)
    pal = rng.integers(30, 226, (num_classes, 3))
    pal[0] = (110, 110, 110)
    return pal


def make_dataset(root, n, h, w, num_classes, seed):
    from PIL import Image
    rng = np.random.default_rng(seed)
    pal = palette(num_classes)
    img_dir, gt_dir = os.path.join(root, "images", "city"), os.path.join(root, "gt", "city")
    os.makedirs(img_dir, exist_ok=True); os.makedirs(gt_dir, exist_ok=True)
    for i in range(n):
        lab = np.zeros((h, w), np.uint8)
        for _ in range(rng.integers(5, 10)):
            c = int(rng.integers(1, num_classes))
            rh, rw = int(rng.integers(h // 6, h // 2)), int(rng.integers(w // 8, w // 3))
            y0, x0 = int(rng.integers(0, h - rh)), int(rng.integers(0, w - rw))
            lab[y0:y0 + rh, x0:x0 + rw] = c
        img = np.clip(pal[lab] + rng.normal(0, 12, (h, w, 3)), 0, 255).astype(np.uint8)
        Image.fromarray(img).save(os.path.join(img_dir, "s%04d_leftImg8bit.png" % i))
        Image.fromarray(lab).save(os.path.join(gt_dir, "s%04d_gtFine_labelIds.png" % i))
    return os.path.join(root, "images"), os.path.join(root, "gt")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "train_demo.json"))
    ap.add_argument("--height", type=int, default=256); ap.add_argument("--width", type=int, default=512)
    ap.add_argument("--train-images", type=int, default=96); ap.add_argument("--val-images", type=int, default=24)
    ap.add_argument("--batch", type=int, default=8); ap.add_argument("--epochs", type=int, default=60)
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--precision", default="fp32")
    ap.add_argument("--deterministic", action="store_true", help="library option `deterministic` (split reductions joined in a fixed order) and seeded host RNGs, batches "
                    "made in-process: two runs must then give the SAME history, bit for bit")
    ap.add_argument("--seed", type=int, default=1234)
    args = ap.parse_args()
    if args.deterministic:
        import random
        random.seed(args.seed); np.random.seed(args.seed)
    from fcn8s_tensorflow_amd.fcn8s import FCN8s
    from fcn8s_tensorflow_amd.batch_generator import BatchGenerator
    C = 20
    root = tempfile.mkdtemp(prefix="fcn8s_demo_")
    try:
        ti, tg = make_dataset(os.path.join(root, "train"), args.train_images, args.height, args.width, C, seed=1)
        vi, vg = make_dataset(os.path.join(root, "val"), args.val_images, args.height, args.width, C, seed=2)
        mk = lambda i, g: BatchGenerator(image_dirs=[i], image_file_extension='png', ground_truth_dirs=[g], image_name_split_separator='_leftImg8bit',
                                         ground_truth_suffix='_gtFine_labelIds', check_existence=True, num_classes=C)
        train_gen = mk(ti, tg).generate(batch_size=args.batch, convert_to_one_hot=True, flip=0.5, brightness=(0.8, 1.25, 0.5), shuffle=True, workers=0 if args.deterministic else 8)
        val_set = mk(vi, vg)
        val_gen = val_set.generate(batch_size=args.batch, convert_to_one_hot=True, shuffle=False, workers=0 if args.deterministic else 4)
        model = FCN8s(vgg16_dir='synthetic:0', num_classes=C)
        model.engine.set_precision(args.precision)
        if args.deterministic:
            model.engine.set_option("deterministic", 1)
        steps, vsteps = args.train_images // args.batch, args.val_images // args.batch
        history = []
        t0 = time.perf_counter()
        for epoch in range(args.epochs):
            with contextlib.redirect_stdout(io.StringIO()):
                model.train(train_gen, epochs=1, steps_per_epoch=steps, learning_rate_schedule=lambda s: args.lr, keep_prob=0.5, l2_regularization=0.0,
                            eval_dataset='val', eval_frequency=1, val_generator=val_gen, val_steps=vsteps, metrics={'loss', 'mean_iou', 'accuracy'},
                            save_during_training=False, record_summaries=True, summaries_dir=os.path.join(root, 'tb'), summaries_name='demo',
                            summaries_frequency=steps)
            rec = dict(epoch=epoch + 1, step=model.g_step, train_loss=round(float(model.training_loss), 4),
                       **{("val_" + n): round(float(v), 4) for n, v in zip(model.metric_names, model.metric_values)})
            history.append(rec)
            print(rec, flush=True)
        train_s = time.perf_counter() - t0
        # save -> restore in a second model -> the same evaluation must give the same numbers
        with contextlib.redirect_stdout(io.StringIO()):
            model.save(model_save_dir=os.path.join(root, 'ckpt'), saver='saved_model', tags=['default'], name='demo', include_global_step=True,
                       include_last_training_loss=True, include_metrics=True, force_save=True)
        saved = [d for d in os.listdir(os.path.join(root, 'ckpt'))][0]
        val_gen2 = mk(vi, vg).generate(batch_size=args.batch, convert_to_one_hot=True, shuffle=False, workers=0)
        with contextlib.redirect_stdout(io.StringIO()):
            model.evaluate(val_gen2, vsteps, metrics={'loss', 'mean_iou', 'accuracy'}, dataset='val')
        final = dict(zip(model.metric_names, (float(v) for v in model.metric_values)))
        restored = FCN8s(model_load_dir=os.path.join(root, 'ckpt', saved), tags=['default'])
        restored.engine.set_precision(args.precision)
        val_gen3 = mk(vi, vg).generate(batch_size=args.batch, convert_to_one_hot=True, shuffle=False, workers=0)
        with contextlib.redirect_stdout(io.StringIO()):
            restored.evaluate(val_gen3, vsteps, metrics={'loss', 'mean_iou', 'accuracy'}, dataset='val')
        again = dict(zip(restored.metric_names, (float(v) for v in restored.metric_values)))
        # predict on one validation batch: pixel accuracy of the argmax against the labels
        imgs, onehot = next(mk(vi, vg).generate(batch_size=4, convert_to_one_hot=True, shuffle=False))
        pred = restored.predict(imgs, argmax=True)
        acc = float((pred == onehot.argmax(-1)).mean())
        out = {"tool": "tools/train_demo.py", "task": "synthetic rectangles, colour -> class, %d classes, %dx%d, %d train / %d val images, batch %d, TF-Adam lr %g, keep_prob 0.5, "
                       "random-init VGG-16 (no pretrained weights offline), flip + brightness augmentation, precision %s%s"
                       % (C, args.width, args.height, args.train_images, args.val_images, args.batch, args.lr, args.precision,
                          ", deterministic mode, host RNG seed %d" % args.seed if args.deterministic else ""),
               "history": history, "train_seconds": round(train_s, 1), "train_images_per_sec_incl_eval_and_feeder": round(args.epochs * steps * args.batch / train_s, 1),
               "final_eval": final, "eval_after_save_and_restore": again, "restored_global_step": restored.g_step if restored.g_step is not None else restored.engine.global_step,
               "predict_pixel_accuracy_4_val_images": round(acc, 4),
               "event_files": sorted(os.listdir(os.path.join(root, 'tb', 'demo')))}
        assert all(abs(final[k] - again[k]) < 1e-6 * max(1.0, abs(final[k])) for k in final), (final, again)
        os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
        print("final", final, "restored", again, "predict acc", acc)
        model.close(); restored.close()
        for g in (train_gen, val_gen):
            g.close()
    finally:
        shutil.rmtree(root, ignore_errors=True)


if __name__ == "__main__":
    main()
