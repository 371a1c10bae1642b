"""Official Cityscapes pixel-level scoring of FCN-8s predictions.

`FCN8s.evaluate()` reproduces the reference's in-graph metric -- `tf.metrics.mean_iou` over the 20
train ids *including* void (fcn8s_tensorflow.py:291-293).  The number comparable with the literature is
the 19-class score of `cityscapesscripts/evaluation/evalPixelLevelSemanticLabeling.py`: predictions are
mapped back to label ids, a 34x34 confusion matrix conf[gt, pred] is accumulated over all pixels
(:173-182, native loop addToConfusionMatrix_impl.c:10-16), and per label
    IoU = tp / (tp + fp + fn),   fp counted only on pixels whose ground truth is NOT ignored (:229-255),
averaged over the labels with a defined score (:286-295).

The label table is the reference author's modified one (`cityscapesscripts/helpers/labels.py:62-99`:
trainId 0 = every ignored label, 1..19 = the evaluated classes).  The confusion matrix is accumulated by
the library's HIP kernel when the inputs live on the GPU, by NumPy otherwise.
"""
from __future__ import annotations

import ctypes as C
import fnmatch
import glob
import math
import os
from collections import OrderedDict

import numpy as np

#        name                    id  trainId  category        catId  hasInstances ignoreInEval  color
LABELS = [
    ('unlabeled',             0,  0, 'void',          0, False, True,  (0, 0, 0)),
    ('ego vehicle',           1,  0, 'void',          0, False, True,  (0, 0, 0)),
    ('rectification border',  2,  0, 'void',          0, False, True,  (0, 0, 0)),
    ('out of roi',            3,  0, 'void',          0, False, True,  (0, 0, 0)),
    ('static',                4,  0, 'void',          0, False, True,  (0, 0, 0)),
    ('dynamic',               5,  0, 'void',          0, False, True,  (111, 74, 0)),
    ('ground',                6,  0, 'void',          0, False, True,  (81, 0, 81)),
    ('road',                  7,  1, 'flat',          1, False, False, (128, 64, 128)),
    ('sidewalk',              8,  2, 'flat',          1, False, False, (244, 35, 232)),
    ('parking',               9,  0, 'flat',          1, False, True,  (250, 170, 160)),
    ('rail track',           10,  0, 'flat',          1, False, True,  (230, 150, 140)),
    ('building',             11,  3, 'construction',  2, False, False, (70, 70, 70)),
    ('wall',                 12,  4, 'construction',  2, False, False, (102, 102, 156)),
    ('fence',                13,  5, 'construction',  2, False, False, (190, 153, 153)),
    ('guard rail',           14,  0, 'construction',  2, False, True,  (180, 165, 180)),
    ('bridge',               15,  0, 'construction',  2, False, True,  (150, 100, 100)),
    ('tunnel',               16,  0, 'construction',  2, False, True,  (150, 120, 90)),
    ('pole',                 17,  6, 'object',        3, False, False, (153, 153, 153)),
    ('polegroup',            18,  0, 'object',        3, False, True,  (153, 153, 153)),
    ('traffic light',        19,  7, 'object',        3, False, False, (250, 170, 30)),
    ('traffic sign',         20,  8, 'object',        3, False, False, (220, 220, 0)),
    ('vegetation',           21,  9, 'nature',        4, False, False, (107, 142, 35)),
    ('terrain',              22, 10, 'nature',        4, False, False, (152, 251, 152)),
    ('sky',                  23, 11, 'sky',           5, False, False, (70, 130, 180)),
    ('person',               24, 12, 'human',         6, True,  False, (220, 20, 60)),
    ('rider',                25, 13, 'human',         6, True,  False, (255, 0, 0)),
    ('car',                  26, 14, 'vehicle',       7, True,  False, (0, 0, 142)),
    ('truck',                27, 15, 'vehicle',       7, True,  False, (0, 0, 70)),
    ('bus',                  28, 16, 'vehicle',       7, True,  False, (0, 60, 100)),
    ('caravan',              29,  0, 'vehicle',       7, True,  True,  (0, 0, 90)),
    ('trailer',              30,  0, 'vehicle',       7, True,  True,  (0, 0, 110)),
    ('train',                31, 17, 'vehicle',       7, True,  False, (0, 80, 100)),
    ('motorcycle',           32, 18, 'vehicle',       7, True,  False, (0, 0, 230)),
    ('bicycle',              33, 19, 'vehicle',       7, True,  False, (119, 11, 32)),
    ('license plate',        -1,  0, 'vehicle',       7, False, True,  (0, 0, 142)),
]
NUM_IDS = 34                                                   # label ids 0..33 (license plate has id -1)

# labels.py:185-192 -- id -> trainId (35 entries; the last one is the -1 label) and trainId -> id
IDS_TO_TRAINIDS_ARRAY = np.zeros(35, np.uint8)
for _n, _id, _tid, *_rest in LABELS:
    IDS_TO_TRAINIDS_ARRAY[_id] = _tid
TRAINIDS_TO_IDS_ARRAY = np.zeros(20, np.uint8)
for _n, _id, _tid, *_rest in LABELS:
    if _tid > 0:
        TRAINIDS_TO_IDS_ARRAY[_tid] = _id
# labels.py:218 -- overlay colours for predict_and_save (alpha 127; void transparent)
TRAINIDS_TO_RGBA_DICT = {0: (0, 0, 0, 0)}
TRAINIDS_TO_RGBA_DICT.update({tid: tuple(col) + (127,) for _n, _id, tid, _c, _ci, _h, _ig, col in LABELS if tid > 0})

IGNORED_IDS = np.array([i for _n, i, *_r in LABELS if i >= 0 and _r[4]], dtype=np.int64)
EVAL_IDS = np.array([i for _n, i, *_r in LABELS if i >= 0 and not _r[4]], dtype=np.int64)       # the 19 evaluated labels
ID_TO_NAME = {i: n for n, i, *_r in LABELS}
CATEGORY_TO_IDS = OrderedDict()
for _n, _id, _tid, _cat, *_r in LABELS:
    if _id >= 0:
        CATEGORY_TO_IDS.setdefault(_cat, []).append(_id)


def confusion_add(conf, gt_ids, pred_ids):
    """conf[gt, pred] += 1 over all pixels.  gt_ids / pred_ids: label-id maps of equal shape (NumPy, or torch
    tensors on the GPU: then the library's confusion kernel is used); conf: (34, 34) int64 array, updated in place."""
    try:
        import torch
        on_gpu = isinstance(gt_ids, torch.Tensor) and gt_ids.is_cuda
    except Exception:
        on_gpu = False
    if on_gpu:
        import torch
        from . import _lib as L
        g = gt_ids.to(torch.uint8).contiguous().view(-1)
        p = pred_ids.to(device=g.device, dtype=torch.int64).contiguous().view(-1)
        d = torch.zeros(NUM_IDS * NUM_IDS, dtype=torch.int64, device=g.device)
        stream = C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)
        L.check(L.lib.fcn8s_op_confusion(stream, C.c_void_p(g.data_ptr()), C.c_void_p(p.data_ptr()), g.numel(),
                                         C.c_void_p(d.data_ptr()), NUM_IDS))
        conf += d.cpu().numpy().reshape(NUM_IDS, NUM_IDS)
        return conf
    g = np.asarray(gt_ids).astype(np.int64).ravel(); p = np.asarray(pred_ids).astype(np.int64).ravel()
    ok = (g >= 0) & (g < NUM_IDS) & (p >= 0) & (p < NUM_IDS)
    conf += np.bincount(g[ok] * NUM_IDS + p[ok], minlength=NUM_IDS * NUM_IDS).reshape(NUM_IDS, NUM_IDS)
    return conf


def iou_for_label(label, conf):
    """evalPixelLevelSemanticLabeling.py:229-255"""
    if label in IGNORED_IDS:
        return float('nan')
    tp = int(conf[label, label])
    fn = int(conf[label, :].sum()) - tp
    not_ignored = [l for l in EVAL_IDS if l != label]
    fp = int(conf[not_ignored, label].sum())
    denom = tp + fp + fn
    return float('nan') if denom == 0 else tp / denom


def iou_for_category(category, conf):
    """evalPixelLevelSemanticLabeling.py:298-335: tp/fn over the category's non-ignored labels, fp from every other
    non-ignored label predicted as one of them."""
    ids = [l for l in CATEGORY_TO_IDS[category] if l not in IGNORED_IDS]
    if not ids:
        return float('nan')
    tp = int(conf[np.ix_(ids, ids)].sum())
    fn = int(conf[ids, :].sum()) - tp
    others = [l for l in EVAL_IDS if l not in ids]
    fp = int(conf[np.ix_(others, ids)].sum())
    denom = tp + fp + fn
    return float('nan') if denom == 0 else tp / denom


def score_average(scores):
    """nan-aware mean (:286-295)"""
    vals = [s for s in scores.values() if not math.isnan(s)]
    return float('nan') if not vals else sum(vals) / len(vals)


class PixelLevelEvaluator:
    """Accumulates the official confusion matrix from FCN-8s predictions (train ids, as `FCN8s.predict`
    returns them) and Cityscapes `*_gtFine_labelIds` ground truth, and reports the 19-class scores."""

    def __init__(self):
        self.conf = np.zeros((NUM_IDS, NUM_IDS), np.int64)

    def add(self, pred_train_ids, gt_label_ids):
        try:
            import torch
            if isinstance(pred_train_ids, torch.Tensor):
                lut = torch.as_tensor(TRAINIDS_TO_IDS_ARRAY.astype(np.int64), device=pred_train_ids.device)
                pred_ids = lut[pred_train_ids.long()]
                gt = gt_label_ids if isinstance(gt_label_ids, torch.Tensor) else torch.as_tensor(np.asarray(gt_label_ids))
                confusion_add(self.conf, gt.to(pred_ids.device), pred_ids)
                return
        except ImportError:
            pass
        pred_ids = TRAINIDS_TO_IDS_ARRAY[np.asarray(pred_train_ids).astype(np.int64)]
        confusion_add(self.conf, np.asarray(gt_label_ids), pred_ids)

    def class_scores(self):
        return OrderedDict((ID_TO_NAME[int(l)], iou_for_label(int(l), self.conf)) for l in range(NUM_IDS))

    def category_scores(self):
        return OrderedDict((c, iou_for_category(c, self.conf)) for c in CATEGORY_TO_IDS)

    def results(self):
        cs, cat = self.class_scores(), self.category_scores()
        return {"classScores": cs, "averageScoreClasses": score_average(cs),
                "categoryScores": cat, "averageScoreCategories": score_average(cat)}


# ---------------------------------------------------------------------------------------------------------
# files either side of the scoring: label-id PNG export and the evaluator's prediction / ground-truth file loop
# ---------------------------------------------------------------------------------------------------------
def save_label_id_png(path, pred_train_ids):
    """Write one prediction (train ids, (H,W), as `FCN8s.predict` returns them) as the single-channel uint8 label-id PNG the
    official evaluator reads (evalPixelLevelSemanticLabeling.py:553-555; ids via labels.py:188-192)."""
    from PIL import Image
    a = np.asarray(pred_train_ids)
    if a.ndim != 2:
        raise ValueError("expected one (H, W) map of train ids, got shape {}".format(a.shape))
    Image.fromarray(TRAINIDS_TO_IDS_ARRAY[a.astype(np.int64)]).save(path)


def cs_file_info(file_name):
    """csHelpers.getCsFileInfo: (city, sequenceNb, frameNb) of `<city>_<seq>_<frame>_<type>[_<type2>].<ext>`."""
    parts = os.path.basename(file_name).split('_')
    if len(parts) < 4:
        raise ValueError("Cannot parse given filename ({}). Does not seem to be a valid Cityscapes file.".format(file_name))
    return parts[0], parts[1], parts[2]


def walk_predictions(prediction_path):
    return [(root, files) for root, _, files in os.walk(prediction_path)]


def find_prediction(prediction_path, ground_truth_file, walk=None):
    """evalPixelLevelSemanticLabeling.py:72-106: the one file `<city>_<seq>_<frame>*.png` anywhere below `prediction_path`
    (`walk`: a walk_predictions() result to reuse, as the evaluator walks the tree once)."""
    if walk is None:
        walk = walk_predictions(prediction_path)
    city, seq, frame = cs_file_info(ground_truth_file)
    pattern = "{}_{}_{}*.png".format(city, seq, frame)
    found = None
    for root, files in walk:
        for f in fnmatch.filter(files, pattern):
            if found is not None:
                raise ValueError("Found multiple predictions for ground truth {}".format(ground_truth_file))
            found = os.path.join(root, f)
    if found is None:
        raise ValueError("Found no prediction for ground truth {}".format(ground_truth_file))
    return found


def evaluate_file_pairs(prediction_files, ground_truth_files, device=None):
    """evaluateImgLists / evaluatePair (evalPixelLevelSemanticLabeling.py:454-498, 550-595): accumulate conf[gt, pred] over
    pairs of label-id PNGs with the evaluator's checks, then the class / category scores.  `device`: a torch cuda device
    to count on the GPU (the library's confusion kernel), None = NumPy."""
    from PIL import Image
    if len(prediction_files) != len(ground_truth_files):
        raise ValueError("List of images for prediction and groundtruth are not of equal size.")
    ev = PixelLevelEvaluator()
    pixels = 0
    for pf, gf in zip(prediction_files, ground_truth_files):
        pred, gt = np.array(Image.open(pf)), np.array(Image.open(gf))
        if pred.ndim != 2:
            raise ValueError("Predicted image has multiple channels.")
        if pred.shape[1] != gt.shape[1]:
            raise ValueError("Image widths of " + pf + " and " + gf + " are not equal.")
        if pred.shape[0] != gt.shape[0]:
            raise ValueError("Image heights of " + pf + " and " + gf + " are not equal.")
        if gt.max() >= NUM_IDS:
            raise ValueError("Unknown label with id {:}".format(int(gt.max())))
        if device is not None:
            import torch
            confusion_add(ev.conf, torch.from_numpy(gt.astype(np.uint8)).to(device), torch.from_numpy(pred.astype(np.int64)).to(device))
        else:
            confusion_add(ev.conf, gt, pred)
        pixels += pred.size
        if int(ev.conf.sum()) != pixels:
            raise ValueError("Number of analyzed pixels and entries in confusion matrix disagree: contMatrix {}, pixels {}".format(int(ev.conf.sum()), pixels))
    res = ev.results()
    res["confMatrix"] = ev.conf
    res["nbPixels"] = pixels
    return res


def evaluate_directory(ground_truth_search, prediction_path, device=None):
    """The evaluator's no-argument mode (:667-676): every ground-truth file matching the glob (the official one is
    `<cityscapes>/gtFine/val/*/*_gtFine_labelIds.png`) against its prediction below `prediction_path`."""
    gts = sorted(glob.glob(ground_truth_search))
    if not gts:
        raise ValueError("Cannot find any ground truth images to use for evaluation. Searched for: {}".format(ground_truth_search))
    walk = walk_predictions(prediction_path)
    return evaluate_file_pairs([find_prediction(prediction_path, g, walk) for g in gts], gts, device)
