"""Official Cityscapes pixel-level scoring (cityscapes_eval.py) against golden vectors produced by the reference's
own evaluator functions (tests/golden/make_golden.py) and against its native confusion-matrix loop.  CPU only;
the GPU accumulation path is covered in tests/test_ops_gpu.py::test_confusion_matrix and below under -m gpu."""
import os

import numpy as np
import pytest

from fcn8s_tensorflow_amd import cityscapes_eval as ce

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_label_table_matches_reference():
    l = np.load(os.path.join(GOLD, "id_lut.npz"))
    np.testing.assert_array_equal(ce.IDS_TO_TRAINIDS_ARRAY, l["lut"])
    assert [n for n, *_ in ce.LABELS] == list(l["names"]) and [r[0] for _, *r in ce.LABELS] == list(l["ids"])
    np.testing.assert_array_equal(np.array([r[7] for r in ce.LABELS]), l["colors"])
    d = np.load(os.path.join(GOLD, "cityscapes_scores.npz"))
    np.testing.assert_array_equal(ce.TRAINIDS_TO_IDS_ARRAY, d["trainids_to_ids"])
    assert len(ce.EVAL_IDS) == 19 and ce.TRAINIDS_TO_RGBA_DICT[1] == (128, 64, 128, 127)
    # round trip: evaluated ids -> train ids -> ids
    np.testing.assert_array_equal(ce.TRAINIDS_TO_IDS_ARRAY[ce.IDS_TO_TRAINIDS_ARRAY[ce.EVAL_IDS]], ce.EVAL_IDS)


def test_scores_match_reference_evaluator():
    d = np.load(os.path.join(GOLD, "cityscapes_scores.npz"))
    ev = ce.PixelLevelEvaluator()
    ev.conf[:] = d["conf"]
    res = ev.results()
    got = np.array([res["classScores"][n] for n in d["class_names"]])
    np.testing.assert_allclose(got, d["class_scores"], rtol=0, atol=1e-15, equal_nan=True)
    assert abs(res["averageScoreClasses"] - float(d["class_avg"])) < 1e-15
    gotc = np.array([res["categoryScores"][n] for n in d["cat_names"]])
    np.testing.assert_allclose(gotc, d["cat_scores"], rtol=0, atol=1e-15, equal_nan=True)
    assert abs(res["averageScoreCategories"] - float(d["cat_avg"])) < 1e-15
    assert np.isnan(res["classScores"]["terrain"]) and np.isnan(res["classScores"]["unlabeled"])   # absent / ignored


def test_accumulation_maps_train_ids_back_and_ignores_void_ground_truth():
    rng = np.random.default_rng(0)
    gt = rng.integers(0, 34, (2, 16, 32)).astype(np.uint8)             # label ids as in *_gtFine_labelIds.png
    pred_tid = ce.IDS_TO_TRAINIDS_ARRAY[gt].astype(np.int64)           # a perfect prediction in train ids (void -> 0)
    ev = ce.PixelLevelEvaluator()
    ev.add(pred_tid, gt)
    res = ev.results()
    for name, s in res["classScores"].items():
        assert np.isnan(s) or s == 1.0                                   # pixels on ignored ground truth never count as fp
    assert res["averageScoreClasses"] == 1.0
    # the confusion matrix equals the reference's native loop on the same pair
    import ctypes as C
    so = os.path.join(os.path.dirname(GOLD), "..", "oracle", "_ref", "libaddToConfusionMatrix.so")
    if os.path.isfile(so):
        ref = C.CDLL(so)
        conf = np.zeros((34, 34), np.uint64)
        p = np.ascontiguousarray(ce.TRAINIDS_TO_IDS_ARRAY[pred_tid]); g = np.ascontiguousarray(gt)
        ref.addToConfusionMatrix(p.ctypes.data_as(C.c_void_p), g.ctypes.data_as(C.c_void_p), p.shape[2], p.shape[0] * p.shape[1],
                                 conf.ctypes.data_as(C.c_void_p), 34)
        np.testing.assert_array_equal(ev.conf, conf.astype(np.int64))


@pytest.mark.gpu
def test_gpu_accumulation_equals_host():
    import torch
    rng = np.random.default_rng(1)
    gt = rng.integers(0, 34, (2, 64, 128)).astype(np.uint8)
    pred = rng.integers(0, 20, (2, 64, 128)).astype(np.int64)
    a, b = ce.PixelLevelEvaluator(), ce.PixelLevelEvaluator()
    a.add(pred, gt)
    b.add(torch.from_numpy(pred).cuda(), torch.from_numpy(gt).cuda())
    np.testing.assert_array_equal(a.conf, b.conf)
    assert a.results()["averageScoreClasses"] == b.results()["averageScoreClasses"]
