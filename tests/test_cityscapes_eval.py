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


def test_file_pair_loop_matches_reference_evaluator(tmp_path):
    """Label-id PNG pairs on disk -> prediction lookup by `<city>_<seq>_<frame>*.png` -> confusion matrix -> scores, against what the
    reference's evaluateImgLists / getPrediction produced on the same files (tests/golden/make_golden.py)."""
    from PIL import Image
    d = np.load(os.path.join(GOLD, "cityscapes_filepairs.npz"))
    gdir, pdir = tmp_path / "gtFine" / "val", tmp_path / "results"
    gts = []
    for i, nm in enumerate(d["names"]):
        city = str(nm).split("_")[0]
        (gdir / city).mkdir(parents=True, exist_ok=True); (pdir / "sub").mkdir(parents=True, exist_ok=True)
        Image.fromarray(d["gts"][i]).save(gdir / city / (str(nm) + "_gtFine_labelIds.png"))
        Image.fromarray(d["preds"][i]).save(pdir / "sub" / (str(nm) + "_leftImg8bit.png"))
        gts.append(str(gdir / city / (str(nm) + "_gtFine_labelIds.png")))
    matched = [ce.find_prediction(str(pdir), g) for g in gts]
    assert [os.path.relpath(m, str(pdir)) for m in matched] == list(d["matched"])
    res = ce.evaluate_directory(str(gdir / "*" / "*_gtFine_labelIds.png"), str(pdir))
    np.testing.assert_array_equal(res["confMatrix"], d["conf"])
    got = np.array([res["classScores"][n] for n in d["class_names"]])
    np.testing.assert_allclose(got, d["class_scores"], rtol=0, atol=1e-15, equal_nan=True)
    assert abs(res["averageScoreClasses"] - float(d["class_avg"])) < 1e-15
    gotc = np.array([res["categoryScores"][n] for n in d["cat_names"]])
    np.testing.assert_allclose(gotc, d["cat_scores"], rtol=0, atol=1e-15, equal_nan=True)
    assert abs(res["averageScoreCategories"] - float(d["cat_avg"])) < 1e-15
    assert res["nbPixels"] == d["gts"].size
    # the evaluator's error cases
    Image.fromarray(d["preds"][0]).save(pdir / (str(d["names"][0]) + "_dup.png"))
    with pytest.raises(ValueError, match="multiple predictions"):
        ce.find_prediction(str(pdir), gts[0])
    with pytest.raises(ValueError, match="no prediction"):
        ce.find_prediction(str(pdir), str(gdir / "x" / "bonn_000000_000001_gtFine_labelIds.png"))
    bad = tmp_path / "bad.png"; Image.fromarray(d["preds"][0][:, :10]).save(bad)
    with pytest.raises(ValueError, match="widths"):
        ce.evaluate_file_pairs([str(bad)], [gts[0]])


def test_label_id_png_export_round_trip(tmp_path):
    rng = np.random.default_rng(3)
    tid = rng.integers(0, 20, (9, 13))
    ce.save_label_id_png(str(tmp_path / "a_000000_000001_pred.png"), tid)
    from PIL import Image
    back = np.array(Image.open(tmp_path / "a_000000_000001_pred.png"))
    assert back.dtype == np.uint8 and back.ndim == 2
    np.testing.assert_array_equal(back, ce.TRAINIDS_TO_IDS_ARRAY[tid])
    np.testing.assert_array_equal(ce.IDS_TO_TRAINIDS_ARRAY[back], tid)          # void (0) maps to id 0 and back
