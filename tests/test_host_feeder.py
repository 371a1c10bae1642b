"""The host feeder drop-in (BatchGenerator + label conversions) against golden vectors captured
from the reference (tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest
from PIL import Image

from fcn8s_tensorflow_amd import ground_truth_conversion_utils as gt
from fcn8s_tensorflow_amd.batch_generator import BatchGenerator, DataError

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def test_one_hot_and_lut_golden():
    d = np.load(os.path.join(GOLD, "onehot.npz"))
    oh = gt.convert_IDs_to_one_hot(d["ids"], 20)
    assert oh.dtype == np.bool_
    np.testing.assert_array_equal(oh, d["onehot"])
    np.testing.assert_array_equal(gt.convert_one_hot_to_IDs(oh), d["back"])
    l = np.load(os.path.join(GOLD, "id_lut.npz"))
    np.testing.assert_array_equal(gt.convert_IDs_to_IDs(l["raw"], l["lut"]), l["mapped"])
    assert l["lut"].max() == 19 and (np.unique(l["lut"]) == np.arange(20)).all()      # 35 ids -> 20 trainIds, 0 = void
    out = gt.convert_IDs_to_IDs_partial(np.array([[1, 2], [3, 1]]), {1: 9})
    assert out.tolist() == [[9, 2], [3, 9]]
    cm = {(255, 0, 0): 1, (0, 255, 0): 2}
    col = np.array([[[255, 0, 0], [0, 255, 0]], [[0, 255, 0], [9, 9, 9]]], np.uint8)
    assert gt.convert_between_IDs_and_colors(col, cm).tolist() == [[1, 2], [2, 0]]


@pytest.fixture()
def dataset(tmp_path):
    d = np.load(os.path.join(GOLD, "batchgen_contract.npz"))
    idir, gdir = tmp_path / "img" / "city", tmp_path / "gt" / "city"
    idir.mkdir(parents=True); gdir.mkdir(parents=True)
    for i in range(3):
        Image.fromarray(d["imgs"][i]).save(idir / ("city_%06d_leftImg8bit.png" % i))
        Image.fromarray(d["gts"][i]).save(gdir / ("city_%06d_gtFine_labelIds.png" % i))
    gen = BatchGenerator(image_dirs=[str(tmp_path / "img")], image_file_extension="png",
                         ground_truth_dirs=[str(tmp_path / "gt")], image_name_split_separator="leftImg8bit",
                         ground_truth_suffix="gtFine_labelIds", check_existence=True, num_classes=20)
    gen.image_paths.sort()
    return gen, d, tmp_path


def test_generate_matches_reference_contract(dataset):
    gen, d, _ = dataset
    assert gen.get_num_files() == int(d["num_files"]) == 3
    g = gen.generate(batch_size=2, convert_to_one_hot=True, shuffle=False)
    for b in range(3):                                  # full batch, short last batch, wrap-around
        x, y = next(g)
        assert x.dtype == np.uint8 and y.dtype == np.bool_
        np.testing.assert_array_equal(x, d["x%d" % b]); np.testing.assert_array_equal(y, d["y%d" % b])
    x, y = next(gen.generate(batch_size=3, convert_to_one_hot=False, shuffle=False))
    np.testing.assert_array_equal(x, d["x_ids"]); np.testing.assert_array_equal(y, d["y_ids"])
    assert y.dtype == np.uint8 and y.shape == (3, 8, 16)


def test_random_crop_places_like_the_reference(dataset):
    gen, d, _ = dataset
    np.random.seed(7)                                   # same seed as the fixture capture
    x, y = next(gen.generate(batch_size=3, convert_to_one_hot=False, random_crop=(12, 10), void_class_id=0, shuffle=False))
    np.testing.assert_array_equal(x, d["x_crop"]); np.testing.assert_array_equal(y, d["y_crop"])


def test_augmentations_keep_shapes_and_labels_consistent(dataset):
    gen, d, _ = dataset
    np.random.seed(0)
    g = gen.generate(batch_size=3, convert_to_one_hot=True, void_class_id=0, brightness=(0.5, 2.0, 1.0), flip=1.0,
                     translate=((1, 3), (1, 2), 1.0), scale=(0.6, 1.4, 1.0), resize=(16, 32), shuffle=False)
    x, y = next(g)
    assert x.shape == (3, 16, 32, 3) and y.shape == (3, 16, 32, 20) and (y.sum(-1) == 1).all()
    x2, y2 = next(gen.generate(batch_size=3, convert_to_one_hot=False, flip=1.0, shuffle=False))
    np.testing.assert_array_equal(x2, d["imgs"][:, :, ::-1]); np.testing.assert_array_equal(y2, d["gts"][:, :, ::-1])
    xg, _ = next(gen.generate(batch_size=1, convert_to_one_hot=False, gray=True, shuffle=False))
    assert xg.shape == (1, 8, 16, 1)


def test_opencv_restatement_against_known_answers():
    """cv2_compat.py (the OpenCV calls of data_generator/batch_generator.py:328-331, :341, :355, :367, :377, :387, :469-486 in OpenCV's own
    8-bit arithmetic) against tests/golden/cv2_vectors.npz: an independent pixel-by-pixel transcription of resize.cpp / color_hsv /
    color_yuv (tests/golden/make_cv2_vectors.py, which also asserts the hand-derived values) -- bit-exact."""
    from fcn8s_tensorflow_amd import cv2_compat as cv
    d = np.load(os.path.join(os.path.dirname(__file__), "golden", "cv2_vectors.npz"))
    for i, (h, w) in enumerate(d["sizes"]):
        np.testing.assert_array_equal(cv.resize_linear(d["img"], int(h), int(w)), d["linear_%d" % i])
        np.testing.assert_array_equal(cv.resize_nearest(d["lab"], int(h), int(w)), d["nearest_%d" % i])
    np.testing.assert_array_equal(cv.resize_linear(d["img2"], 6, 8), d["linear_half"])                 # exact 2x shrink: INTER_AREA's box mean
    np.testing.assert_array_equal(cv.resize_linear(d["img2"], 12, 8), d["linear_half_x_only"])
    np.testing.assert_array_equal(cv.rgb2hsv(d["px"]), d["hsv"])
    np.testing.assert_array_equal(cv.hsv2rgb(d["hsv_in"]), d["rgb_from_hsv"])
    for i, f in enumerate(d["factors"]):
        np.testing.assert_array_equal(cv.brightness(d["px"], float(f)), d["bright_%d" % i])
    np.testing.assert_array_equal(cv.rgb2gray(d["px"]), d["gray"])
    # hand-derived (make_cv2_vectors.py HAND): cv2.resize of the row [0, 100] to 4 pixels, the primaries' hues
    np.testing.assert_array_equal(cv.resize_linear(np.array([[0, 100]], np.uint8), 1, 4), [[0, 25, 75, 100]])
    np.testing.assert_array_equal(cv.rgb2hsv(np.array([[255, 0, 0], [0, 255, 0], [0, 0, 255], [128, 64, 32]], np.uint8)),
                                  [[0, 255, 255], [60, 255, 255], [120, 255, 255], [10, 191, 128]])
    np.testing.assert_array_equal(cv.resize_nearest(np.arange(5, dtype=np.uint8)[None], 1, 3), [[0, 1, 3]])          # floor(x * 5/3), not pixel centres
    # label ids: nearest never invents a class, bilinear of a constant image is that constant, the identity resize is a copy
    lab = d["lab"]
    assert set(np.unique(cv.resize_nearest(lab, 29, 5))) <= set(np.unique(lab))
    np.testing.assert_array_equal(cv.resize_linear(np.full((9, 7, 3), 201, np.uint8), 20, 13), np.full((20, 13, 3), 201, np.uint8))
    np.testing.assert_array_equal(cv.resize_linear(d["img"], 11, 14), d["img"])
    np.testing.assert_array_equal(cv.translate(lab, 3, -2, 9)[:-2, 3:], lab[2:, :-3])
    assert (cv.translate(lab, 3, -2, 9)[-2:] == 9).all() and (cv.translate(lab, 3, -2, 9)[:, :3] == 9).all()


def test_generator_resize_scale_brightness_gray_use_the_opencv_arithmetic(dataset):
    """generate() routes resize / scale / brightness / gray through cv2_compat, in the reference's order of operations and draws."""
    from fcn8s_tensorflow_amd import cv2_compat as cv
    gen, d, _ = dataset
    x, y = next(gen.generate(batch_size=3, convert_to_one_hot=False, resize=(13, 21), shuffle=False))
    for i in range(3):
        np.testing.assert_array_equal(x[i], cv.resize_linear(d["imgs"][i], 13, 21))
        np.testing.assert_array_equal(y[i], cv.resize_nearest(d["gts"][i], 13, 21))
    np.random.seed(5)
    xb, _ = next(gen.generate(batch_size=1, convert_to_one_hot=False, brightness=(0.5, 2.0, 1.0), shuffle=False))
    np.random.seed(5)
    np.random.uniform(0, 1)                             # p
    factor = np.random.uniform(0.5, 2.0)                # random_br, drawn inside _brightness (:476)
    np.testing.assert_array_equal(xb[0], cv.brightness(d["imgs"][0], factor))
    np.random.seed(6)
    xs, ys = next(gen.generate(batch_size=1, convert_to_one_hot=False, scale=(0.5, 0.9, 1.0), void_class_id=7, shuffle=False))
    np.random.seed(6)
    np.random.uniform(0, 1)
    f = np.random.uniform(0.5, 0.9)
    h, w = d["imgs"][0].shape[:2]
    sh, sw = int(h * f), int(w * f); yo, xo = abs(int((h - sh) / 2)), abs(int((w - sw) / 2))
    want = np.zeros_like(d["imgs"][0]); want[yo:yo + sh, xo:xo + sw] = cv.resize_linear(d["imgs"][0], sh, sw)
    wl = np.full_like(d["gts"][0], 7); wl[yo:yo + sh, xo:xo + sw] = cv.resize_nearest(d["gts"][0], sh, sw)
    np.testing.assert_array_equal(xs[0], want); np.testing.assert_array_equal(ys[0], wl)
    xg, _ = next(gen.generate(batch_size=1, convert_to_one_hot=False, gray=True, shuffle=False))
    np.testing.assert_array_equal(xg[0, ..., 0], cv.rgb2gray(d["imgs"][0]))


def test_errors_and_process_all(dataset, tmp_path):
    gen, d, root = dataset
    with pytest.raises(DataError):
        BatchGenerator(image_dirs=[str(root / "gt" / "nothing")])
    os.remove(root / "gt" / "city" / "city_000001_gtFine_labelIds.png")
    with pytest.raises(DataError):
        BatchGenerator(image_dirs=[str(root / "img")], ground_truth_dirs=[str(root / "gt")],
                       image_name_split_separator="leftImg8bit", ground_truth_suffix="gtFine_labelIds")
    nogt = BatchGenerator(image_dirs=[str(root / "img")])
    with pytest.raises(ValueError):
        next(nogt.generate(batch_size=1))               # one-hot conversion without ground truth
    imgs = next(nogt.generate(batch_size=2, convert_to_one_hot=False, shuffle=False))
    assert isinstance(imgs, np.ndarray) and imgs.shape == (2, 8, 16, 3)
    exp = BatchGenerator(image_dirs=[str(root / "img")], root_dir=str(root), export_dir=str(root / "out"))
    exp.process_all(resize=(4, 8), batch_size=2)
    assert sorted(os.listdir(root / "out" / "img" / "city")) == sorted(os.listdir(root / "img" / "city"))
    assert np.asarray(Image.open(root / "out" / "img" / "city" / "city_000000_leftImg8bit.png")).shape == (4, 8, 3)


def test_prefetch_wrapper_preserves_order_and_errors(dataset):
    from fcn8s_tensorflow_amd.batch_generator import prefetch
    gen, d, _ = dataset
    g = prefetch(gen.generate(batch_size=2, convert_to_one_hot=True, shuffle=False), depth=2)
    for b in range(3):
        x, y = next(g)
        np.testing.assert_array_equal(x, d["x%d" % b]); np.testing.assert_array_equal(y, d["y%d" % b])

    def boom():
        yield 1
        raise KeyError("x")
    p = prefetch(boom())
    assert next(p) == 1
    with pytest.raises(KeyError):
        next(p)


def _png_tree(tmp_path, n=5, h=16, w=24, C=20):
    from PIL import Image
    rng = np.random.default_rng(0)
    (tmp_path / 'img' / 'a').mkdir(parents=True); (tmp_path / 'gt' / 'a').mkdir(parents=True)
    for i in range(n):
        Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)).save(tmp_path / 'img' / 'a' / ('f%d_x.png' % i))
        Image.fromarray(rng.integers(0, C, (h, w), dtype=np.uint8)).save(tmp_path / 'gt' / 'a' / ('f%d_gt.png' % i))
    from fcn8s_tensorflow_amd.batch_generator import BatchGenerator
    return lambda: BatchGenerator([str(tmp_path / 'img')], 'png', [str(tmp_path / 'gt')], '_x', '_gt', True, C)


def test_worker_processes_and_class_id_batches_do_not_change_what_a_seeded_run_yields(tmp_path):
    """`workers` moves decoding / augmentation into processes, `next_ids()` skips the one-hot expansion: with the same seeds both
    give the batches the serial, one-hot generator gives (the random decisions are drawn in the parent, in the reference's order)."""
    import random
    make = _png_tree(tmp_path)
    kw = dict(batch_size=2, random_crop=(12, 20), brightness=(0.5, 2.0, 0.5), flip=0.5, translate=((0, 3), (0, 3), 0.5),
              scale=(0.7, 1.3, 0.5), void_class_id=0, shuffle=True)

    def run(workers, ids):
        random.seed(5); np.random.seed(5)
        g = make().generate(workers=workers, **kw)
        out = [g.next_ids() if ids else next(g) for _ in range(4)]         # wraps around once (5 files, batches of 2: 2, 2, 1, 2)
        g.close()
        return out
    serial = run(0, False)
    assert [b[0].shape[0] for b in serial] == [2, 2, 1, 2]
    assert serial[0][1].dtype == bool and serial[0][1].shape == (2, 12, 20, 20)
    for other, ids in ((run(2, False), False), (run(0, True), True), (run(2, True), True)):
        for (i0, g0), (i1, g1) in zip(serial, other):
            np.testing.assert_array_equal(i0, i1)
            np.testing.assert_array_equal(np.argmax(g0, -1).astype(np.uint8), g1 if ids else np.argmax(g1, -1).astype(np.uint8))
            if ids:
                assert g1.dtype == np.uint8 and g1.ndim == 3


def test_worker_pool_error_does_not_leave_replies_behind(tmp_path):
    """A sample that fails in a decode worker raises in the parent only after every reply of that batch has been collected: the
    next batch must not read the previous batch's acknowledgements as its own (it would return a buffer still being written)."""
    import random
    make = _png_tree(tmp_path)
    random.seed(1); np.random.seed(1)
    gen = make()
    g = gen.generate(batch_size=2, workers=2, shuffle=False)
    first = g.next_ids()
    victim = gen.image_paths[2]
    os.rename(victim, victim + ".gone")
    with pytest.raises(RuntimeError):
        g.next_ids()                                   # batch (2, 3): sample 2 cannot be read, sample 3's reply is still collected
    os.rename(victim + ".gone", victim)
    random.seed(1); np.random.seed(1)
    ref = make().generate(batch_size=2, workers=0, shuffle=False)
    want = [ref.next_ids() for _ in range(4)]
    got = g.next_ids()                                 # the generator carries on behind the failed batch: (4,), then wraps
    np.testing.assert_array_equal(got[0], want[2][0]); np.testing.assert_array_equal(got[1], want[2][1])
    got = g.next_ids()
    np.testing.assert_array_equal(got[0], want[3][0]); np.testing.assert_array_equal(got[1], want[3][1])
    np.testing.assert_array_equal(first[0], want[0][0])
    g.close()


def test_worker_pool_says_it_is_closed_after_a_worker_died(tmp_path):
    """A decode worker that dies takes the pool down (its other replies never come); whoever catches that error and pulls another batch
    gets a clear RuntimeError about the closed pool -- not a ZeroDivisionError from an empty process list."""
    import random
    make = _png_tree(tmp_path)
    random.seed(1); np.random.seed(1)
    g = make().generate(batch_size=2, workers=2, shuffle=False)
    g.next_ids()
    g.pool.procs[0].kill(); g.pool.procs[0].wait()
    with pytest.raises(RuntimeError, match="worker died"):
        g.next_ids()
    with pytest.raises(RuntimeError, match="shut down after a worker died"):
        g.next_ids()
    g.close()


def test_feeder_thread_stops_when_the_consumer_gives_up():
    """_Feeder.close() (called from a finally in the train / evaluate loops) ends the helper thread: it pulls no further batches from
    the user's generator and is not left inside Engine.stage while the model closes."""
    import itertools
    import time
    from fcn8s_tensorflow_amd.fcn8s import _Feeder
    pulled = []

    def gen():
        for i in itertools.count():
            pulled.append(i)
            yield np.zeros((1, 32, 32, 3), np.float32), np.zeros((1, 32, 32, 20), bool)      # not the stageable kind: engine unused
    f = _Feeder(None, gen(), 100)
    a, b = f.next()
    assert a.shape == (1, 32, 32, 3)
    f.close()
    assert not f.t.is_alive()
    n = len(pulled)
    assert n <= 4
    time.sleep(0.2)
    assert len(pulled) == n

    def bad():
        yield np.zeros((1, 32, 32, 3), np.float32), np.zeros((1, 32, 32, 20), bool)
        raise KeyError("boom")
    f = _Feeder(None, bad(), 5)
    f.next()
    with pytest.raises(KeyError):
        f.next()
    f.close()
    assert not f.t.is_alive()
