#!/usr/bin/env python3
"""Generates tests/golden/*.npz by RUNNING the reference modules that still import
in this container (never shipped: only the input/output vectors are committed).

    python tests/golden/make_golden.py        # needs /root/reference

Sources exercised (reference file:line):
  helpers/ground_truth_conversion_utils.py:3-24   convert_IDs_to_IDs   (LUT gather)
  helpers/ground_truth_conversion_utils.py:80-88  one-hot <-> ids      (np.bool patched: removed in NumPy >= 1.24)
  cityscapesscripts/helpers/labels.py:62-99,185-187 (ids_to_trainIds_array)  the 35 -> 20 trainId table (exec'd up to :188; :191 overflows on NumPy 2)
  cityscapesscripts/evaluation/addToConfusionMatrix_impl.c:3-17  compiled by oracle/Makefile into oracle/_ref/
  data_generator/batch_generator.py:16-130,140-417  BatchGenerator contract (cv2 stubbed, scipy.misc.imread -> PIL)
  cityscapesscripts/evaluation/evalPixelLevelSemanticLabeling.py:173-182,229-335  official IoU scoring functions (labels / PIL stubbed)
  cityscapesscripts/evaluation/evalPixelLevelSemanticLabeling.py:72-106,454-498,550-595  prediction lookup + the file-pair evaluation loop on PNG pairs
"""
import ctypes
import os
import sys
import tempfile
import types

import numpy as np

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))


def main():
    if not hasattr(np, "bool"):
        np.bool = bool                     # the reference predates NumPy 1.24
    sys.path.insert(0, REF)
    rng = np.random.default_rng(2024)

    # ---- one-hot / id helpers --------------------------------------------------
    from helpers.ground_truth_conversion_utils import (convert_IDs_to_IDs, convert_IDs_to_one_hot,
                                                       convert_one_hot_to_IDs)
    ids = rng.integers(0, 20, (6, 9), dtype=np.uint8)
    onehot = convert_IDs_to_one_hot(ids, 20)
    back = convert_one_hot_to_IDs(onehot)
    np.savez_compressed(os.path.join(HERE, "onehot.npz"), ids=ids, onehot=onehot, back=back)

    # ---- label table: exec labels.py up to the LUT definitions ------------------
    src = open(os.path.join(REF, "cityscapesscripts/helpers/labels.py")).read().split("\n")
    ns = {}
    exec("\n".join(src[:188]), ns)
    lut = np.asarray(ns["ids_to_trainIds_array"])
    raw = rng.integers(0, len(lut), (5, 7), dtype=np.uint8)
    np.savez_compressed(os.path.join(HERE, "id_lut.npz"), lut=lut, raw=raw, mapped=convert_IDs_to_IDs(raw, lut),
                        names=np.array([l.name for l in ns["labels"]]), ids=np.array([l.id for l in ns["labels"]]),
                        train_ids=np.array([l.trainId for l in ns["labels"]]),
                        colors=np.array([l.color for l in ns["labels"]], dtype=np.int64))

    # ---- native confusion matrix (the reference's only native code) -------------
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "_ref", "libaddToConfusionMatrix.so"))
    lib.addToConfusionMatrix.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint, ctypes.c_uint,
                                         ctypes.c_void_p, ctypes.c_uint]
    lib.addToConfusionMatrix.restype = None
    pred = rng.integers(0, 20, (33, 47), dtype=np.uint8)
    gt = rng.integers(0, 20, (33, 47), dtype=np.uint8)
    gt[:, :10] = 4                                               # some classes dominate, some vanish
    conf = np.zeros((20, 20), np.uint64)
    for _ in range(2):                                          # accumulates across calls
        lib.addToConfusionMatrix(pred.ctypes.data, gt.ctypes.data, pred.shape[1], pred.shape[0], conf.ctypes.data, 20)
    ka_pred = np.array([0, 1, 2, 1, 0, 2], np.uint8); ka_gt = np.array([0, 1, 1, 1, 2, 2], np.uint8)
    ka = np.zeros((3, 3), np.uint64)
    lib.addToConfusionMatrix(ka_pred.ctypes.data, ka_gt.ctypes.data, 6, 1, ka.ctypes.data, 3)
    np.savez_compressed(os.path.join(HERE, "confmat.npz"), pred=pred, gt=gt, conf=conf.astype(np.int64),
                        ka_pred=ka_pred, ka_gt=ka_gt, ka_conf=ka.astype(np.int64))

    # ---- BatchGenerator contract ---------------------------------------------------
    from PIL import Image
    import scipy.misc
    sys.modules["cv2"] = types.ModuleType("cv2")                  # absent here; only used by resize/flip/translate/scale
    scipy.misc.imread = lambda p: np.asarray(Image.open(p))
    scipy.misc.imsave = lambda p, a: Image.fromarray(a).save(p)
    from data_generator.batch_generator import BatchGenerator
    imgs = rng.integers(0, 256, (3, 8, 16, 3), dtype=np.uint8)
    gts = rng.integers(0, 20, (3, 8, 16), dtype=np.uint8)
    with tempfile.TemporaryDirectory() as d:
        idir, gdir = os.path.join(d, "img", "city"), os.path.join(d, "gt", "city")
        os.makedirs(idir); os.makedirs(gdir)
        for i in range(3):
            Image.fromarray(imgs[i]).save(os.path.join(idir, "city_%06d_leftImg8bit.png" % i))
            Image.fromarray(gts[i]).save(os.path.join(gdir, "city_%06d_gtFine_labelIds.png" % i))
        gen = BatchGenerator(image_dirs=[os.path.join(d, "img")], image_file_extension="png",
                             ground_truth_dirs=[os.path.join(d, "gt")], image_name_split_separator="leftImg8bit",
                             ground_truth_suffix="gtFine_labelIds", check_existence=True, num_classes=20)
        gen.image_paths.sort()
        g = gen.generate(batch_size=2, convert_to_one_hot=True, shuffle=False)
        out = {"imgs": imgs, "gts": gts, "num_files": gen.get_num_files()}
        for b in range(3):                                        # full, short (last of the pass), wrapped
            x, y = next(g)
            out["x%d" % b] = x; out["y%d" % b] = y
        g2 = gen.generate(batch_size=3, convert_to_one_hot=False, shuffle=False)
        x, y = next(g2)
        out["x_ids"] = x; out["y_ids"] = y
        np.random.seed(7)
        g3 = gen.generate(batch_size=3, convert_to_one_hot=False, random_crop=(12, 10), void_class_id=0, shuffle=False)
        x, y = next(g3)
        out["x_crop"] = x; out["y_crop"] = y
        np.savez_compressed(os.path.join(HERE, "batchgen_contract.npz"), **out)
    # ---- official pixel-level scoring: the reference evaluator's own functions on a random confusion matrix --------
    import PIL
    if not hasattr(PIL, "PILLOW_VERSION"):
        PIL.PILLOW_VERSION = PIL.__version__                     # csHelpers.py:16 predates Pillow 7
    lab_mod = types.ModuleType("labels")
    exec("\n".join(src[:188]), lab_mod.__dict__)
    sys.modules["labels"] = lab_mod                               # labels.py:191 overflows on NumPy 2; the table above it is intact
    sys.path.insert(0, os.path.join(REF, "cityscapesscripts", "helpers"))
    sys.path.insert(0, os.path.join(REF, "cityscapesscripts", "evaluation"))
    os.environ.setdefault("CITYSCAPES_DATASET", "/tmp")
    import evalPixelLevelSemanticLabeling as ev
    conf = ev.generateMatrix(ev.args)
    conf[:] = rng.integers(0, 5000, conf.shape)
    conf[:, 22] = 0; conf[22, :] = 0                             # an evaluated class that never occurs -> nan, left out of the mean
    cls = {ev.id2label[l].name: ev.getIouScoreForLabel(l, conf, ev.args) for l in ev.args.evalLabels}
    cat = {c: ev.getIouScoreForCategory(c, conf, ev.args) for c in ev.category2labels.keys()}
    np.savez_compressed(os.path.join(HERE, "cityscapes_scores.npz"), conf=conf.astype(np.int64),
                        class_names=np.array(list(cls)), class_scores=np.array(list(cls.values()), dtype=np.float64),
                        class_avg=ev.getScoreAverage(cls, ev.args),
                        cat_names=np.array(list(cat)), cat_scores=np.array(list(cat.values()), dtype=np.float64),
                        cat_avg=ev.getScoreAverage(cat, ev.args),
                        trainids_to_ids=np.array([0] + [ns["trainIds_to_ids_dict"][t] for t in range(1, 20)], dtype=np.uint8))   # labels.py:188-192 (its loop overflows uint8 on the id -1 label under NumPy 2; [0] is forced to 0 there)
    # ---- the evaluator's file loop: getPrediction (:72-106) and evaluateImgLists / evaluatePair (:454-498, :550-595) on PNG pairs ----
    ev.CSUPPORT = False                                            # its python loop (the compiled loop is pinned through confmat.npz)
    ev.args.evalInstLevelScore = False; ev.args.evalPixelAccuracy = False; ev.args.quiet = True; ev.args.JSONOutput = False
    names = ["frankfurt_000000_000294", "frankfurt_000001_007973", "munster_000005_000019"]
    gts = rng.integers(0, 34, (3, 12, 20), dtype=np.uint8)
    preds = ce_like = None
    tid = lut[gts]                                                 # the perfect train-id map, then damaged
    preds = np.array([ns["trainIds_to_ids_dict"].get(int(t), 0) if t > 0 else 0 for t in tid.ravel()], np.uint8).reshape(tid.shape)
    flip = rng.random(preds.shape) < 0.3
    preds[flip] = np.array([7, 8, 11, 21, 23, 24, 26, 33], np.uint8)[rng.integers(0, 8, int(flip.sum()))]
    with tempfile.TemporaryDirectory() as d:
        gdir, pdir = os.path.join(d, "gtFine", "val"), os.path.join(d, "results")
        gt_files = []
        for i, nm in enumerate(names):
            city = nm.split("_")[0]
            os.makedirs(os.path.join(gdir, city), exist_ok=True); os.makedirs(os.path.join(pdir, "sub"), exist_ok=True)
            gf = os.path.join(gdir, city, nm + "_gtFine_labelIds.png"); gt_files.append(gf)
            Image.fromarray(gts[i]).save(gf)
            Image.fromarray(preds[i]).save(os.path.join(pdir, "sub", nm + "_leftImg8bit.png"))
        ev.args.predictionPath = pdir; ev.args.predictionWalk = None
        pred_files = [ev.getPrediction(ev.args, g) for g in gt_files]
        res = ev.evaluateImgLists(pred_files, gt_files, ev.args)
        np.savez_compressed(os.path.join(HERE, "cityscapes_filepairs.npz"), names=np.array(names), gts=gts, preds=preds,
                            matched=np.array([os.path.relpath(p, pdir) for p in pred_files]),
                            conf=np.asarray(res["confMatrix"], dtype=np.int64),
                            class_names=np.array(list(res["classScores"])), class_scores=np.array(list(res["classScores"].values()), dtype=np.float64),
                            class_avg=res["averageScoreClasses"],
                            cat_names=np.array(list(res["categoryScores"])), cat_scores=np.array(list(res["categoryScores"].values()), dtype=np.float64),
                            cat_avg=res["averageScoreCategories"])
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
