"""Pins the CPU oracle (oracle/fcn8s_oracle.py, torch-CPU) against
  (a) the explicit-loop C restatement of the TF op definitions (oracle/fcn8s_oracle.c),
  (b) hand-derived known answers,
  (c) golden vectors captured from the reference modules that import here
      (tests/golden/make_golden.py) and the reference's native confusion-matrix file.
Runs on CPU in seconds."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
import torch

from oracle import fcn8s_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")


@pytest.fixture(scope="module")
def clib():
    so = os.path.join(ROOT, "oracle", "libfcn8s_oracle.so")
    if not os.path.isfile(so):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "libfcn8s_oracle.so"])
    return C.CDLL(so)


def p(a):
    return a.ctypes.data_as(C.c_void_p)


def nchw(a):
    return torch.tensor(a).permute(0, 3, 1, 2)


def nhwc(t):
    return t.permute(0, 2, 3, 1).contiguous().numpy()


@pytest.mark.parametrize("K,Cin,Cout", [(3, 3, 5), (1, 6, 4), (7, 2, 3)])
def test_conv_same_matches_c_loops(clib, K, Cin, Cout):
    rng = np.random.default_rng(0)
    N, H, W = 2, 5, 6
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = rng.standard_normal((K, K, Cin, Cout)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    dy = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    y = np.empty((N, H, W, Cout), np.float32)
    clib.orc_conv2d_same(p(x), p(w), p(b), p(y), N, H, W, Cin, Cout, K, 1)
    xt = nchw(x).requires_grad_(True); wt = torch.tensor(w).requires_grad_(True); bt = torch.tensor(b).requires_grad_(True)
    yt = orc.conv2d_same_t(xt, wt, bt, relu=True)
    np.testing.assert_allclose(nhwc(yt.detach()), y, rtol=1e-5, atol=1e-5)
    # backward of the linear part
    dx = np.empty_like(x); dw = np.empty_like(w); db = np.empty_like(b)
    clib.orc_conv2d_same_bwd(p(x), p(w), p(dy), p(dx), p(dw), p(db), N, H, W, Cin, Cout, K)
    orc.conv2d_same_t(xt, wt, bt).backward(nchw(dy))
    np.testing.assert_allclose(nhwc(xt.grad), dx, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(wt.grad.numpy(), dw, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bt.grad.numpy(), db, rtol=1e-4, atol=1e-4)


def test_conv_known_answer():
    # 3x3 all-ones kernel on an all-ones 3x3 image counts the in-bounds taps (SAME zero padding)
    y = orc.conv2d_same_t(torch.ones(1, 1, 3, 3), torch.ones(3, 3, 1, 1))
    assert y[0, 0].tolist() == [[4, 6, 4], [6, 9, 6], [4, 6, 4]]
    # HWIO orientation is cross-correlation (no kernel flip): a kernel with a single 1 at (ky=0,kx=2) reads x[h-1, w+1]
    x = torch.arange(9.0).view(1, 1, 3, 3)
    w = torch.zeros(3, 3, 1, 1); w[0, 2] = 1
    assert orc.conv2d_same_t(x, w)[0, 0].tolist() == [[0, 0, 0], [1, 2, 0], [4, 5, 0]]


def test_maxpool_matches_c_loops(clib):
    rng = np.random.default_rng(1)
    N, H, W, Cc = 2, 6, 4, 3
    x = rng.standard_normal((N, H, W, Cc)).astype(np.float32)
    dy = rng.standard_normal((N, H // 2, W // 2, Cc)).astype(np.float32)
    y = np.empty((N, H // 2, W // 2, Cc), np.float32); dx = np.empty_like(x)
    clib.orc_maxpool2x2(p(x), p(y), N, H, W, Cc)
    clib.orc_maxpool2x2_bwd(p(x), p(dy), p(dx), N, H, W, Cc)
    xt = nchw(x).requires_grad_(True)
    yt = orc.maxpool2x2_t(xt)
    yt.backward(nchw(dy))
    np.testing.assert_array_equal(nhwc(yt.detach()), y)
    np.testing.assert_array_equal(nhwc(xt.grad), dx)


@pytest.mark.parametrize("K,S", [(4, 2), (16, 8)])
def test_conv_transpose_matches_c_loops(clib, K, S):
    rng = np.random.default_rng(2)
    N, Hi, Wi, Cin, Cout = 1, 3, 2, 3, 4
    x = rng.standard_normal((N, Hi, Wi, Cin)).astype(np.float32)
    w = rng.standard_normal((K, K, Cout, Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    dy = rng.standard_normal((N, Hi * S, Wi * S, Cout)).astype(np.float32)
    y = np.empty((N, Hi * S, Wi * S, Cout), np.float32)
    clib.orc_conv2d_transpose_same(p(x), p(w), p(b), p(y), N, Hi, Wi, Cin, Cout, K, S)
    xt = nchw(x).requires_grad_(True); wt = torch.tensor(w).requires_grad_(True); bt = torch.tensor(b).requires_grad_(True)
    yt = orc.conv2d_transpose_same_t(xt, wt, bt, S)
    assert tuple(yt.shape) == (N, Cout, Hi * S, Wi * S)           # out = in * stride
    np.testing.assert_allclose(nhwc(yt.detach()), y, rtol=1e-4, atol=1e-4)
    dx = np.empty_like(x); dw = np.empty_like(w); db = np.empty_like(b)
    clib.orc_conv2d_transpose_same_bwd(p(x), p(w), p(dy), p(dx), p(dw), p(db), N, Hi, Wi, Cin, Cout, K, S)
    yt.backward(nchw(dy))
    np.testing.assert_allclose(nhwc(xt.grad), dx, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(wt.grad.numpy(), dw, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(bt.grad.numpy(), db, rtol=1e-4, atol=1e-3)


def test_softmax_xent_matches_c_loops(clib):
    rng = np.random.default_rng(3)
    npix, Cc = 50, 20
    logits = (rng.standard_normal((1, 5, 10, Cc)) * 4).astype(np.float32)
    lab = rng.integers(0, Cc, (1, 5, 10)).astype(np.uint8)
    dl = np.empty((npix, Cc), np.float32)
    clib.orc_softmax_xent.restype = C.c_double
    loss_c = clib.orc_softmax_xent(p(logits), p(lab), p(dl), C.c_size_t(npix), Cc)
    lt = nchw(logits).requires_grad_(True)
    loss = orc.total_loss_t({k: torch.zeros(1) for k in orc.DECODER_KERNELS}, lt, torch.tensor(orc.one_hot(lab, Cc)).float(), 0.0)
    loss.backward()
    assert abs(float(loss.detach()) - loss_c) < 1e-5
    np.testing.assert_allclose(nhwc(lt.grad).reshape(npix, Cc), dl, atol=1e-7)
    # uniform logits: loss = ln C (hand-derived)
    z = torch.zeros(1, Cc, 2, 2)
    l0 = orc.total_loss_t({k: torch.zeros(1) for k in orc.DECODER_KERNELS}, z, torch.tensor(orc.one_hot(np.zeros((1, 2, 2), np.uint8), Cc)).float(), 0.0)
    assert abs(float(l0) - np.log(Cc)) < 1e-6


def test_l2_regulariser_is_half_sum_of_squares():
    P = {k: torch.full((2, 2), 3.0) for k in orc.DECODER_KERNELS}
    z = torch.zeros(1, 20, 1, 1)
    lab = torch.tensor(orc.one_hot(np.zeros((1, 1, 1), np.uint8), 20)).float()
    base = float(orc.total_loss_t(P, z, lab, 0.0))
    assert abs(float(orc.total_loss_t(P, z, lab, 0.5)) - base - 0.5 * 0.5 * 6 * 4 * 9.0) < 1e-5


def test_softmax_argmax_matches_c_loops(clib):
    rng = np.random.default_rng(4)
    npix, Cc = 64, 20
    logits = (rng.standard_normal((npix, Cc)) * 2).astype(np.float32)
    logits[:8, 3] = logits[:8, 9] = 10.0                           # exact ties -> lowest index
    sm = np.empty((npix, Cc), np.float32); am = np.empty(npix, np.int64)
    clib.orc_softmax_argmax(p(logits), p(sm), p(am), C.c_size_t(npix), Cc)
    sm_t = orc.softmax(logits)
    np.testing.assert_allclose(sm_t, sm, atol=1e-6)
    np.testing.assert_array_equal(np.argmax(sm_t, -1), am)
    assert (am[:8] == 3).all()


def test_tf_adam_matches_c_loops_and_hand_value(clib):
    rng = np.random.default_rng(5)
    n = 1000
    th = rng.standard_normal(n).astype(np.float32); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    thc, mc, vc = th.copy(), m.copy(), v.copy()
    for t in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        th, m, v = orc.tf_adam_step(th, g, m, v, t, 1e-3)
        clib.orc_tf_adam(p(thc), p(g), p(mc), p(vc), C.c_size_t(n), t, C.c_float(1e-3), C.c_float(0.9), C.c_float(0.999), C.c_float(1e-8))
    np.testing.assert_allclose(th, thc, atol=1e-6)
    # first step, g = 1: m = 0.1, v = 0.001, lr_t = lr*sqrt(0.001)/0.1 -> theta -= lr * 1/(1 + eps/sqrt(0.001))
    t1, _, _ = orc.tf_adam_step(np.zeros(1, np.float32), np.ones(1, np.float32), np.zeros(1, np.float32), np.zeros(1, np.float32), 1, 0.5)
    assert abs(float(t1[0]) + 0.5 / (1 + 1e-8 / np.sqrt(0.001))) < 1e-6
    # epsilon placement differs from torch.optim.Adam (eps inside the bias-corrected denominator)
    tt = torch.zeros(1, requires_grad=True); opt = torch.optim.Adam([tt], lr=0.5, eps=1e-2)
    tt.grad = torch.full((1,), 1e-3); opt.step()
    tf1, _, _ = orc.tf_adam_step(np.zeros(1, np.float32), np.full(1, 1e-3, np.float32), np.zeros(1, np.float32), np.zeros(1, np.float32), 1, 0.5, eps=1e-2)
    assert abs(float(tf1[0]) - float(tt.detach())) > 1e-3


# ---- golden vectors from the reference ---------------------------------------------------------
def test_golden_one_hot():
    d = np.load(os.path.join(GOLD, "onehot.npz"))
    got = orc.one_hot(d["ids"], 20)
    assert got.dtype == np.bool_ and got.shape == d["onehot"].shape
    np.testing.assert_array_equal(got, d["onehot"])
    np.testing.assert_array_equal(np.argmax(got, -1), d["back"])


def test_golden_confusion_matrix_reference_native():
    d = np.load(os.path.join(GOLD, "confmat.npz"))
    cm = 2 * orc.confusion_matrix(d["gt"], d["pred"], 20)          # the fixture accumulated the pair twice
    np.testing.assert_array_equal(cm, d["conf"])
    np.testing.assert_array_equal(orc.confusion_matrix(d["ka_gt"], d["ka_pred"], 3), d["ka_conf"])
    assert d["ka_conf"].tolist() == [[1, 0, 0], [0, 2, 1], [1, 0, 1]]   # SURVEY 8c known answer


def test_confusion_c_restatement_matches_reference_build(clib):
    ref_so = os.path.join(ROOT, "oracle", "_ref", "libaddToConfusionMatrix.so")
    if not os.path.isfile(ref_so):
        pytest.skip("oracle/_ref not built (reference not present)")
    ref = C.CDLL(ref_so)
    rng = np.random.default_rng(6)
    gt = rng.integers(0, 20, (31, 17), dtype=np.uint8); pr = rng.integers(0, 20, (31, 17), dtype=np.uint8)
    c_ref = np.zeros((20, 20), np.uint64)
    ref.addToConfusionMatrix(p(pr), p(gt), 17, 31, p(c_ref), 20)
    c_mine = np.zeros((20, 20), np.int64)
    clib.orc_confusion(p(gt.astype(np.int64)), p(pr.astype(np.int64)), C.c_size_t(gt.size), p(c_mine), 20)
    np.testing.assert_array_equal(c_mine, c_ref.astype(np.int64))
    np.testing.assert_array_equal(orc.confusion_matrix(gt, pr, 20), c_mine)


def test_mean_iou_semantics():
    cm = np.array([[5, 1, 0], [2, 3, 0], [0, 0, 0]])                # class 2 absent from GT and predictions
    iou0, iou1 = 5 / 8, 3 / 6
    assert abs(orc.mean_iou_from_confusion(cm) - (iou0 + iou1) / 2) < 1e-12          # valid-class mean
    assert abs(orc.mean_iou_from_confusion(cm, valid_only=False) - (iou0 + iou1) / 3) < 1e-12
    assert abs(orc.accuracy_from_confusion(cm) - 8 / 11) < 1e-12
    sm = orc.StreamingMetrics(3)
    sm.update(1.0, np.array([0, 1]), np.array([0, 0])); sm.update(3.0, np.array([1]), np.array([1]))
    loss, miou, acc = sm.values()
    assert loss == 2.0 and abs(acc - 2 / 3) < 1e-12                   # loss is a mean over batches, not samples


def test_full_graph_shapes_and_finite_difference():
    widths = (4, 4, 8, 8, 8, 16, 16)
    P = orc.init_params(4, widths, fc6_ksize=3, seed=0, decoder_std_scale=50.0, bias_std=0.1)
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (1, 32, 64, 3), dtype=np.uint8)
    lab = rng.integers(0, 4, (1, 32, 64), dtype=np.uint8)
    logits, acts = orc.forward(P, img, keep=True)
    assert logits.shape == (1, 32, 64, 4) and acts["pool3"].shape == (1, 4, 8, 8) and acts["fc7"].shape == (1, 1, 2, 16)
    loss, g, _ = orc.loss_and_grads(P, img, orc.one_hot(lab, 4).astype(np.float32), l2_rate=0.1, dtype=torch.float64)
    # central finite difference on one decoder weight and one encoder bias (float64)
    for name, idx in (("fc7_pool4_conv2d_trans/kernel", (1, 2, 0, 3)), ("conv3_2/biases", (5,))):
        eps = 1e-4
        Pp = {k: v.astype(np.float64).copy() for k, v in P.items()}; Pm = {k: v.astype(np.float64).copy() for k, v in P.items()}
        Pp[name][idx] += eps; Pm[name][idx] -= eps
        lp, _, _ = orc.loss_and_grads(Pp, img, orc.one_hot(lab, 4).astype(np.float64), l2_rate=0.1, dtype=torch.float64)
        lm, _, _ = orc.loss_and_grads(Pm, img, orc.one_hot(lab, 4).astype(np.float64), l2_rate=0.1, dtype=torch.float64)
        fd = (lp - lm) / (2 * eps)
        assert abs(fd - g[name][idx]) < 1e-5 * max(1.0, abs(fd)), name
    pred = orc.predict(P, img)
    assert pred.dtype == np.int64 and pred.shape == (1, 32, 64)


def test_param_table_matches_reference_names():
    specs = orc.param_specs(20)
    for n in ("conv3_3/filter", "conv3_3/biases", "conv4_3/filter", "fc6/weights", "fc6/biases", "fc7/weights", "fc7/biases",
              "pool3_1x1/kernel", "pool4_1x1/bias", "fc7_1x1/kernel", "fc7_conv2d_trans/kernel",
              "fc7_pool4_conv2d_trans/bias", "fc7_pool4_pool3_conv2d_trans/kernel"):   # fcn8s_tensorflow.py:331-350
        assert n in specs
    assert specs["fc6/weights"] == (7, 7, 512, 4096) and specs["fc7_pool4_pool3_conv2d_trans/kernel"] == (16, 16, 20, 20)
    assert sum(int(np.prod(s)) for s in specs.values()) == 134473144


def test_pool_routes_reproduce_the_unrouted_gradients_and_steer_ties():
    """`routes=` (the device's max-pool decisions fed back to the checker): the oracle's own routes reproduce its unrouted loss and
    gradients exactly; at an exact tie, choosing the other tied element is a different valid subgradient and moves the gradient."""
    widths = (4, 4, 8, 8, 8, 16, 16)
    P = orc.init_params(4, widths, fc6_ksize=3, seed=1, decoder_std_scale=30.0, bias_std=0.05)
    rng = np.random.default_rng(3)
    img = rng.integers(0, 256, (2, 32, 64, 3), dtype=np.uint8)
    lab = orc.one_hot(rng.integers(0, 4, (2, 32, 64), dtype=np.uint8), 4).astype(np.float32)
    _, acts = orc.forward(P, img, keep=True)
    routes, gaps = orc.pool_routes(acts)
    assert set(routes) == {"pool%d" % b for b in range(1, 6)}
    assert routes["pool1"].shape == (2, 16, 32, 4) and routes["pool1"].dtype == np.uint8 and routes["pool1"].max() <= 4
    on = acts["pool1"] > 0
    np.testing.assert_array_equal(routes["pool1"] == 4, ~on)              # 4 = the window's maximum is not > 0
    l0, g0, z0 = orc.loss_and_grads(P, img, lab, l2_rate=1e-3)
    l1, g1, z1 = orc.loss_and_grads(P, img, lab, l2_rate=1e-3, routes=routes)
    assert abs(l0 - l1) < 1e-6 * abs(l0)
    assert np.abs(z0 - z1).max() < 1e-5 * np.abs(z0).max()       # (same values pooled; the convs behind may pick another summation order)
    for k in g0:
        np.testing.assert_allclose(g1[k], g0[k], rtol=0, atol=2e-5 * np.abs(g0[k]).max())
    # known answer on one window: entries (3, 7, 7, 1) -> first maximum is element 1; element 2 is the tied alternative
    z = torch.tensor([[[[3.0, 7.0], [7.0, 1.0]]]], requires_grad=True)
    for r, want in ((1, [[0, 1], [0, 0]]), (2, [[0, 0], [1, 0]])):
        y = orc._pool_routed(z, torch.tensor([[[[r]]]]))
        assert float(y.detach()) == 7.0
        (gz,) = torch.autograd.grad(y.sum(), z)
        assert gz[0, 0].tolist() == want
    y = orc._pool_routed(z, torch.tensor([[[[4]]]]))
    assert float(y.detach()) == 0.0


def test_bf16_convs_mode_of_the_oracle():
    """`bf16_convs` (FCN8S_PREC_BF16_FWD's forward arithmetic for conv3_1 .. conv5_3): the forward VALUE of those layers is the fp32
    convolution of the bf16-rounded operands -- checked on one layer against an explicit rounding -- the backward pass is the fp32 one
    (straight-through), and blocks 1-2 are untouched."""
    widths = (4, 4, 8, 8, 8, 16, 16)
    P = orc.init_params(4, widths, fc6_ksize=3, seed=2, decoder_std_scale=30.0, bias_std=0.05)
    rng = np.random.default_rng(4)
    img = rng.integers(0, 256, (1, 32, 32, 3), dtype=np.uint8)
    _, a32 = orc.forward(P, img, keep=True)
    _, a16 = orc.forward(P, img, keep=True, bf16_convs=True)
    for k in ("conv1_1", "conv1_2", "conv2_1", "conv2_2", "pool2"):
        np.testing.assert_array_equal(a32[k], a16[k])
    d = np.abs(a16["conv3_1"] - a32["conv3_1"]).max() / np.abs(a32["conv3_1"]).max()
    assert 1e-5 < d < 2e-2, d                                                 # bf16 rounding is visible, and small
    x = torch.from_numpy(a32["pool2"]).permute(0, 3, 1, 2)
    w = torch.from_numpy(P["conv3_1/filter"]); b = torch.from_numpy(P["conv3_1/biases"])
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
    want = torch.relu(torch.nn.functional.conv2d(rb(x), rb(w).permute(3, 2, 0, 1), b, padding=1)).permute(0, 2, 3, 1).numpy()
    np.testing.assert_allclose(a16["conv3_1"], want, rtol=0, atol=1e-5 * np.abs(want).max())
    lab = orc.one_hot(rng.integers(0, 4, (1, 32, 32), dtype=np.uint8), 4).astype(np.float32)
    l16, g16, _ = orc.loss_and_grads(P, img, lab, bf16_convs=True, bf16_fc=True)
    l32, g32, _ = orc.loss_and_grads(P, img, lab)
    assert np.isfinite(l16) and abs(l16 - l32) < 0.1 * abs(l32)
    for k in g32:
        assert np.isfinite(g16[k]).all() and g16[k].shape == g32[k].shape


def test_float_ops_against_scipy():
    """A pin of the oracle's floating-point building blocks that is neither TensorFlow (not installable here) nor this repository's own restatements:
    SciPy.  SAME convolution = scipy.signal.correlate2d(mode='same') summed over input channels; the stride-s, k = 2s transposed convolution of
    tf.layers.conv2d_transpose(padding='same') = scipy.signal.convolve2d(mode='full') of the zero-stuffed input with the kernel, cropped by
    p = (k - s) / 2 on every side; log-softmax / softmax = scipy.special; all in float64, agreement to 1e-12."""
    import torch
    from scipy import signal, special
    rng = np.random.default_rng(3)
    # SAME conv, 3x3 and 7x7
    for K in (3, 7):
        x = rng.standard_normal((1, 9, 11, 3)); w = rng.standard_normal((K, K, 3, 2)); b = rng.standard_normal(2)
        got = orc.conv2d_same_t(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w), torch.tensor(b)).permute(0, 2, 3, 1).numpy()
        want = np.stack([sum(signal.correlate2d(x[0, :, :, ci], w[:, :, ci, co], mode="same") for ci in range(3)) + b[co] for co in range(2)], -1)[None]
        assert np.abs(got - want).max() < 1e-12
    # transposed conv (k, s) = (4, 2) and (16, 8): kernel [k, k, Cout, Cin]
    for K, S in ((4, 2), (16, 8)):
        x = rng.standard_normal((1, 5, 6, 2)); w = rng.standard_normal((K, K, 3, 2)); b = rng.standard_normal(3)
        got = orc.conv2d_transpose_same_t(torch.tensor(x).permute(0, 3, 1, 2), torch.tensor(w), torch.tensor(b), S).permute(0, 2, 3, 1).numpy()
        p = (K - S) // 2
        want = np.zeros((1, 5 * S, 6 * S, 3))
        for co in range(3):
            acc = 0.0
            for ci in range(2):
                up = np.zeros(((5 - 1) * S + 1, (6 - 1) * S + 1)); up[::S, ::S] = x[0, :, :, ci]        # zero-stuffed input
                acc = acc + signal.convolve2d(up, w[:, :, co, ci], mode="full")                       # y[i s + ky, j s + kx] += x[i, j] w[ky, kx]
            want[0, :, :, co] = acc[p:p + 5 * S, p:p + 6 * S] + b[co]
        assert got.shape == want.shape and np.abs(got - want).max() < 1e-12
    # softmax / cross entropy
    lg = rng.standard_normal((2, 3, 4, 5)) * 4
    assert np.abs(orc.softmax(lg) - special.softmax(lg, -1)).max() < 1e-12
    lab = rng.integers(0, 5, (2, 3, 4))
    P = {k: torch.zeros(1, dtype=torch.float64) for k in orc.DECODER_KERNELS}
    loss = float(orc.total_loss_t(P, torch.tensor(lg).permute(0, 3, 1, 2), torch.tensor(orc.one_hot(lab, 5).astype(np.float64)), 0.0))
    want = float(-np.take_along_axis(special.log_softmax(lg, -1), lab[..., None], -1).mean())
    assert abs(loss - want) < 1e-12
