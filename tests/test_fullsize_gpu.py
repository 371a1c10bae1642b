"""BASELINE.json's full-size configurations on the GPU.

config 1: one 256x256 image, forward + argmax -- the CPU oracle stands where the TF1 CPU path stood.
config 2: 1024x512 bs1 fp32 inference -- logits within 1e-3 of the CPU oracle, per-pixel argmax identical
          wherever the oracle's top-2 logit gap exceeds twice that tolerance.
config 3: 1024x512 bs16 training step -- too large for the CPU oracle in a test, so it is checked through
          size-independent properties: closed-form loss / bias gradient at zero decoder weights, batch
          linearity of the gradients (what data-parallel training relies on), and determinism of the forward.
config 5 (shape): 2048x1024, 4 images per GPU (global 32 over 8 GPUs) -- same properties at that size."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import fcn8s_oracle as orc  # noqa: E402  (checker only)


def _check_inference(e, P, img, tol=1e-3):
    e.set_params(P)
    pred = e.predict(img, argmax=True)
    n, h, w = img.shape[:3]
    logits = e.activation("logits", (n, h, w, 20))
    ref = orc.forward(P, img)
    scale = max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(logits - ref).max()) < tol * scale
    srt = np.sort(ref, -1)
    safe = (srt[..., -1] - srt[..., -2]) > 2 * tol * scale
    ref_arg = np.argmax(orc.softmax(ref), -1)
    assert pred.dtype == np.int64 and pred.shape == (n, h, w)
    assert (pred[safe] == ref_arg[safe]).all(), int((pred[safe] != ref_arg[safe]).sum())
    return safe.mean(), int((pred != ref_arg).sum())


def test_config1_single_256x256_image_forward_argmax():
    from fcn8s_tensorflow_amd.engine import Engine
    rng = np.random.default_rng(7)                                               # c1 = one 256x256 image, seed 7
    img = rng.integers(0, 256, (1, 256, 256, 3), dtype=np.uint8)
    e = Engine(20)
    safe, differ = _check_inference(e, orc.init_params(20, seed=0, decoder_std_scale=30.0, bias_std=0.05), img)
    assert safe > 0.95 and differ <= 4, differ              # of 65 536 pixels (measured 0-1: profiles/parity_r04.json, c1)
    # the softmax output (what predict(argmax=False) returns, fcn8s_tensorflow.py:268) against the oracle's
    # (decoder scaled so that the logits are O(1): a saturated softmax would turn logit round-off into 0/1 flips)
    P = orc.init_params(20, seed=1, decoder_std_scale=5.0, bias_std=0.05)
    e.set_params(P)
    sm = e.predict(img, argmax=False)
    ref = orc.forward(P, img)
    assert 0.05 < float(np.abs(ref).max()) < 50.0
    assert sm.shape == (1, 256, 256, 20) and np.abs(sm.sum(-1) - 1).max() < 1e-5
    assert np.abs(sm - orc.softmax(ref)).max() < 1e-3
    e.close()


def test_config2_inference_1024x512_bs1_vs_oracle():
    from fcn8s_tensorflow_amd.engine import Engine
    P = orc.init_params(20, seed=0, decoder_std_scale=30.0, bias_std=0.05)      # lively decoder so the margins are real
    img, _ = orc.synthetic_batch(1, 512, 1024)
    e = Engine(20)
    e.set_params(P)
    pred = e.predict(img, argmax=True)
    logits = e.activation("logits", (1, 512, 1024, 20))
    ref = orc.forward(P, img)
    scale = max(1.0, float(np.abs(ref).max()))
    assert float(np.abs(logits - ref).max()) < 1e-3 * scale
    # argmax must agree wherever the oracle's top-2 logit gap exceeds twice the logit tolerance
    # (closer pairs are legitimately ambiguous under fp32 summation-order differences)
    srt = np.sort(ref, -1)
    safe = (srt[..., -1] - srt[..., -2]) > 2e-3 * scale
    assert safe.mean() > 0.95, safe.mean()
    ref_arg = np.argmax(orc.softmax(ref), -1)
    assert (pred[safe] == ref_arg[safe]).all(), int((pred[safe] != ref_arg[safe]).sum())
    # ... and the ambiguous pixels are a handful: at most 16 of the 524 288 (measured 2-7 over decoders and boxes, every one of them
    # a pixel whose two best logits are closer than the logit error itself; profiles/parity_r04.json holds the count)
    n_diff = int((pred != ref_arg).sum())
    print("config 2: %d of %d pixels differ from the oracle's argmax, all at top-2 margins <= 2e-3 x logit scale" % (n_diff, pred.size))
    assert n_diff <= 16, n_diff
    # with the reference's own decoder init (sigma 1e-3 / 1e-2, fcn8s_tensorflow.py:159-160) the logits are tiny;
    # check them in relative terms as well
    P2 = orc.init_params(20, seed=0)
    e.set_params(P2)
    e.predict(img)
    l2 = e.activation("logits", (1, 512, 1024, 20))
    r2 = orc.forward(P2, img)
    assert float(np.abs(l2 - r2).max()) < 1e-3 * max(float(np.abs(r2).max()), 1e-30) + 1e-12
    # north_star's bound as stated: ABSOLUTE |logit difference| < 1e-3, with a decoder whose logits are O(1-10)
    # (the x30 decoder above gives logits in the hundreds, where 1e-3 relative is a 0.1 absolute bar)
    P3 = orc.init_params(20, seed=1, decoder_std_scale=5.0, bias_std=0.05)
    e.set_params(P3)
    pred3 = e.predict(img, argmax=True)
    l3 = e.activation("logits", (1, 512, 1024, 20))
    r3 = orc.forward(P3, img)
    amax = float(np.abs(r3).max())
    err = float(np.abs(l3 - r3).max())
    print("config 2: max |logit| %.3f, max abs logit error %.3e" % (amax, err))
    assert 0.5 < amax < 50.0, amax
    assert err < 1e-3, err
    srt = np.sort(r3, -1)
    safe3 = (srt[..., -1] - srt[..., -2]) > 2e-3
    assert (pred3[safe3] == np.argmax(orc.softmax(r3), -1)[safe3]).all()
    e.close()


@pytest.mark.parametrize("mode,kw,rel_bound,abs_bound_x5,max_mismatch,margin", [
    # (precision, what the oracle rounds, logit error / scale, absolute logit error with the O(1-10) decoder, differing pixels allowed, safe margin / scale)
    ("f32x3", {}, 1e-4, 1e-3, 16, 2e-3),                    # fp32-accurate by construction: the fp32 bounds (measured 7e-6, 1.1e-4, 3-5 pixels)
    ("f32x2", {}, 4e-4, 4e-3, 160, 2e-3),                   # 16 significand bits: measured 1.2e-4, 1.9e-3 ABSOLUTE (outside north_star's 1e-3: a reduced-precision mode), 47-54 pixels
    ("bf16_fwd", {"bf16_fc": True, "bf16_convs": True}, 3e-2, 0.5, 10500, 6e-2),      # 8-bit forward operands against the same-rounding oracle: measured 8e-3 .. 1e-2, 0.16, 0.7-0.8 % of the pixels
    ("bf16_fwd_x2", {"bf16_fc": True, "bf16_convs": True}, 3e-2, 0.5, 10500, 6e-2)])
def test_config2_precision_modes_1024x512_bs1_vs_oracle(mode, kw, rel_bound, abs_bound_x5, max_mismatch, margin):
    """Config 2's size in every optional arithmetic (round 3 gated these modes at 2 x 32x64 and 256x256 only; the figures were in
    profiles/parity_r03.json but no test held them): logits against the oracle -- for the bf16-forward modes the oracle that rounds the
    same operands of the same layers -- relative to the logit scale with the x30 decoder and ABSOLUTE with the O(1-10) decoder, the
    argmax identical wherever the oracle's top-2 margin exceeds `margin` x scale, and the number of differing pixels bounded."""
    from fcn8s_tensorflow_amd.engine import Engine
    img, _ = orc.synthetic_batch(1, 512, 1024)
    e = Engine(20, precision=mode)
    for pname, P in (("x30", orc.init_params(20, seed=0, decoder_std_scale=30.0, bias_std=0.05)), ("x5", orc.init_params(20, seed=1, decoder_std_scale=5.0, bias_std=0.05))):
        e.set_params(P)
        pred = e.predict(img, argmax=True)
        logits = e.activation("logits", (1, 512, 1024, 20))
        ref = orc.forward(P, img, **kw)
        scale = max(1.0, float(np.abs(ref).max()))
        err = float(np.abs(logits - ref).max())
        srt = np.sort(ref, -1)
        safe = (srt[..., -1] - srt[..., -2]) > margin * scale
        ref_arg = np.argmax(orc.softmax(ref), -1)
        n_diff = int((pred != ref_arg).sum())
        print("config 2 [%s, decoder %s]: max |logit| %.1f, error %.3e (%.2e of scale), %d of %d pixels differ, %d above the margin"
              % (mode, pname, scale, err, err / scale, n_diff, pred.size, int((pred != ref_arg)[safe].sum())))
        assert err < rel_bound * scale, (pname, err / scale)
        if pname == "x5":
            assert 0.5 < scale < 50.0 and err < abs_bound_x5, err
        assert safe.mean() > 0.1 and (pred[safe] == ref_arg[safe]).all(), (float(safe.mean()), int((pred != ref_arg)[safe].sum()))
        assert n_diff <= max_mismatch, n_diff
    e.close()


@pytest.mark.parametrize("variant,options,bound", [("F(6x6) for all twelve 3x3 layers, F(4x4,4x4) for fc6 (the default)", {}, 1e-4)])
def test_config3_gradients_1024x512_bs1_vs_oracle(variant, options, bound):
    """Every one of the 42 gradient tensors of a full-width training step at 1024x512 against the oracle (autograd over the CPU
    restatement): F(6x6,3x3) data / weight gradients at 14 706 tiles per image, fc6's F(4x4,4x4) weight gradient at depth 2048,
    split-K atomics, conv1_1's VALU weight gradient at 512x1024 -- none of which the small cases reach at this size.  The oracle
    differentiates along the ReLU / max-pool decisions the device took, after the test has checked that those differ from the oracle's
    own only at fp32 coin flips (units within 1e-5 of zero, window maxima within 2e-5 of the layer's largest activation of each other).
    Bound: 1e-4 of each tensor's largest gradient -- a tenth of SURVEY section 7 step 5's 1e-3; measured 3.4-4.5e-5 (worst: the last
    transposed conv's bias; conv1_1/filter, the end of the backward chain, 1.2e-5), the same for F(4x4) and for direct convolution
    (profiles/parity_r03.json).  Without the alignment the very same gradients sit 1.9e-3 from the oracle's -- and the fp32 oracle
    0.94e-3 from its own float64 run: that is 54 of 16 million pool windows (and 69 of 98 million ReLU units) taking the other branch
    of a tie, not arithmetic error."""
    from fcn8s_tensorflow_amd.engine import Engine
    P = orc.init_params(20, seed=4, decoder_std_scale=30.0, bias_std=0.05)
    img, lab = orc.synthetic_batch(1, 512, 1024)
    e = Engine(20, options=options)
    e.set_params(P)
    loss = e.forward_backward(img, lab, keep_prob=1.0, l2_rate=1e-3)
    g = e.get_grads()
    br = e.relu_branches((1, 512, 1024))
    rt = e.pool_routes((1, 512, 1024))
    acts_dev = {k: e.activation(k, v.shape, missing_ok=True) for k, v in br.items()}      # (None for a conv fused with its successor: no tensor exists)
    e.close()
    _, acts = orc.forward(P, img, keep=True)
    n_relu = n_route = 0
    for k, on in br.items():
        d = on != (acts[k] > 0)
        n_relu += int(d.sum())
        if d.any():
            big = np.abs(acts[k][d]) if acts_dev[k] is None else np.maximum(np.abs(acts_dev[k][d]), np.abs(acts[k][d]))
            assert big.max() < 1e-5 * np.abs(acts[k]).max(), k
    own, gaps = orc.pool_routes(acts)
    for k in rt:
        d = rt[k] != own[k]
        n_route += int(d.sum())
        tie = d & ((rt[k] == 4) == (own[k] == 4))
        if tie.any():
            assert gaps[k][tie].max() <= 2e-5, (k, float(gaps[k][tie].max()))
        onoff = d & ~tie
        if onoff.any():
            assert np.abs(acts[k][onoff]).max() < 1e-5 * np.abs(acts[k]).max(), k
    del acts, acts_dev, own, gaps
    loss_ref, g_ref, _ = orc.loss_and_grads(P, img, orc.one_hot(lab, 20).astype(np.float32), l2_rate=1e-3, branches=br, routes=rt)
    assert abs(loss - loss_ref) < 1e-4 * max(1.0, abs(loss_ref)), (loss, loss_ref)
    assert len(g_ref) == 42
    errs = {k: float(np.abs(np.asarray(g[k], np.float64) - g_ref[k]).max() / (np.abs(g_ref[k]).max() + 1e-30)) for k in g_ref}
    ranked = sorted(errs.items(), key=lambda kv: -kv[1])
    print("config 3 gradients at 1024x512 [%s]: %d ReLU units and %d pool routes differ from the oracle's (all coin flips); error / max per tensor, "
          "worst five:" % (variant, n_relu, n_route), ", ".join("%s %.2e" % kv for kv in ranked[:5]), "; median %.2e" % float(np.median(list(errs.values()))))
    for k, err in ranked:
        assert err < bound, (k, err)


def test_config3_training_step_1024x512_bs16_properties():
    import torch
    from fcn8s_tensorflow_amd.engine import Engine
    N, H, W, C = 16, 512, 1024, 20
    e = Engine(C, seed=3)
    e.init_params(seed=0)
    img, lab = orc.synthetic_batch(N, H, W)
    imgd, labd = torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda()

    # (1) all decoder kernels and biases zero -> logits == 0 -> loss = ln C and
    #     d loss / d (last bias)[c] = 1/C - count_c / npix   (closed form, any size)
    zero = {k: np.zeros(s[0], np.float32) for k, s in e.specs.items() if "1x1" in k or "trans" in k}
    e.set_params(zero)
    loss = e.forward_backward(imgd, labd, keep_prob=1.0)
    assert abs(loss - np.log(C)) < 1e-5
    gb = e.grad_view("fc7_pool4_pool3_conv2d_trans/bias").cpu().numpy()
    want = 1.0 / C - np.bincount(lab.ravel(), minlength=C) / lab.size
    assert np.abs(gb - want).max() < 1e-6
    # every encoder gradient is exactly zero (the decoder passes nothing back)
    assert float(e.grad_view("conv1_1/filter").abs().max()) == 0.0 and float(e.grad_view("fc6/weights").abs().max()) == 0.0

    # (2) batch linearity: grad(batch of 16) == mean of grad(first 8), grad(last 8)  (keep_prob 1)
    e.init_params(seed=1)
    e.forward_backward(imgd, labd, keep_prob=1.0)
    g_full = e.flat_grads.clone()
    e.forward_backward(imgd[:8], labd[:8], keep_prob=1.0)
    g_a = e.flat_grads.clone()
    e.forward_backward(imgd[8:], labd[8:], keep_prob=1.0)
    g_mean = 0.5 * (g_a + e.flat_grads)
    for name in ("conv1_1/filter", "conv3_2/filter", "conv5_3/biases", "fc6/weights", "fc7/weights", "pool4_1x1/kernel",
                 "fc7_pool4_pool3_conv2d_trans/kernel"):
        shape, off = e.specs[name]
        n = int(np.prod(shape))
        a, b = g_full[off:off + n], g_mean[off:off + n]
        # fp32 round-off only (summation order, Winograd transforms): an order of magnitude inside the 2e-3 gradient tolerance
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-12, (name, float((a - b).abs().max()) / float(b.abs().max()))

    # (3) forward determinism (no atomics on the forward path): identical logits bit for bit
    p1 = e.predict(imgd[:2], argmax=False).clone()
    p2 = e.predict(imgd[:2], argmax=False)
    assert torch.equal(p1, p2)

    # (4) full-size directional-derivative check of the whole backward pass through a 4-image batch: a CENTRAL difference
    #     along g, (L(theta + eps g) - L(theta - eps g)) / (2 eps |g|^2), equals 1 up to the third-order term, so the step can be
    #     large enough for the fp32 loss resolution (2.4e-7) not to matter
    e.forward_backward(imgd[:4], labd[:4], keep_prob=1.0)
    g = e.flat_grads.clone()
    theta = e.flat_params.clone()                       # torch view of the library's parameter buffer
    norm2 = float((g.double() ** 2).sum())
    ratios = []
    for target in (2e-5, 4e-5, 6e-5):                   # intended first-order loss change (at 1e-4 the third-order term is already 3 %)
        eps = target / norm2
        e.flat_params.copy_(theta - eps * g)
        lm = e.forward_backward(imgd[:4], labd[:4], keep_prob=1.0)
        e.flat_params.copy_(theta + eps * g)
        lp = e.forward_backward(imgd[:4], labd[:4], keep_prob=1.0)
        ratios.append((lp - lm) / (2 * target))
    e.flat_params.copy_(theta)
    print("config 3 directional derivative ratios:", ratios)
    assert any(0.97 < r < 1.03 for r in ratios), ratios
    e.close()


def test_config3_bs16_1024x512_each_image_vs_oracle():
    """Config 3 at its OWN batch size against the oracle (until now bs1 vs oracle + bs16 self-consistency).
    One forward / backward pass of the 16-image batch at 1024x512 on the device; then
      (i)  the logits of images 0, 7 and 15 of that batch against `orc.forward` of each image alone (1e-3 of the logit scale, argmax
           identical above the 2e-3 margin, at most 16 differing pixels per image) -- image k > 0 sits in other tiles of every Winograd
           image, in other row blocks of every GEMM and behind other pool routing bytes than image 0 does;
      (ii) all 42 gradient tensors of the bs16 step against the MEAN of the sixteen per-image oracle gradients (the loss is a mean over
           N.H.W, fcn8s_tensorflow.py:253; the L2 term is the same in every per-image loss), each per-image oracle pass differentiating
           along the ReLU / pool decisions the device took for THAT image, and recording (oracle `stats=`) that those differ from its own
           only at fp32 coin flips: ReLU units within 1e-5 of zero, window maxima within 2e-5 of each other (of the layer's largest).
    Bound 1e-4 of each tensor's largest entry, as for bs1."""
    import torch
    from fcn8s_tensorflow_amd.engine import Engine
    N, H, W, C = 16, 512, 1024, 20
    P = orc.init_params(C, seed=4, decoder_std_scale=30.0, bias_std=0.05)
    img, lab = orc.synthetic_batch(N, H, W)
    e = Engine(C)
    e.set_params(P)
    loss = e.forward_backward(torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda(), keep_prob=1.0, l2_rate=1e-3)
    g = {k: np.asarray(v, np.float64) for k, v in e.get_grads().items()}
    logits = e.activation("logits", (N, H, W, C))
    pred = np.argmax(logits, -1)                       # (argmax of the fp32 softmax is tested on its own; here the batch position is the subject)
    br = e.relu_branches((N, H, W))
    rt = e.pool_routes((N, H, W))
    e.close()
    torch.set_num_threads(min(32, torch.get_num_threads()))
    # (i) logits of three images of the batch
    for k in (0, 7, 15):
        ref = orc.forward(P, img[k:k + 1])[0]
        scale = max(1.0, float(np.abs(ref).max()))
        err = float(np.abs(logits[k] - ref).max())
        srt = np.sort(ref, -1)
        safe = (srt[..., -1] - srt[..., -2]) > 2e-3 * scale
        ref_arg = np.argmax(ref, -1)
        n_diff = int((pred[k] != ref_arg).sum())
        print("config 3, image %d of 16: logit error %.2e of scale %.0f, %d of %d pixels differ (%d above the margin)"
              % (k, err / scale, scale, n_diff, ref_arg.size, int((pred[k] != ref_arg)[safe].sum())))
        assert err < 1e-3 * scale, (k, err / scale)
        assert safe.mean() > 0.95 and (pred[k][safe] == ref_arg[safe]).all(), k
        assert n_diff <= 16, (k, n_diff)
    del logits, pred
    # (ii) gradients: mean of the per-image oracle gradients along the device's decisions
    acc, loss_sum, n_relu, n_route, worst_relu, worst_gap = None, 0.0, 0, 0, 0.0, 0.0
    for k in range(N):
        st = {}
        lk, gk, _ = orc.loss_and_grads(P, img[k:k + 1], orc.one_hot(lab[k:k + 1], C).astype(np.float32), l2_rate=1e-3,
                                       branches={n: v[k:k + 1] for n, v in br.items()}, routes={n: v[k:k + 1] for n, v in rt.items()}, stats=st)
        for name, (cnt, total, dist) in st.items():
            if name in rt:
                n_route += cnt; worst_gap = max(worst_gap, dist)
                assert dist <= 2e-5, (k, name, cnt, dist)
            else:
                n_relu += cnt; worst_relu = max(worst_relu, dist)
                assert dist < 1e-5, (k, name, cnt, dist)
        loss_sum += lk
        if acc is None:
            acc = {n: np.asarray(v, np.float64) for n, v in gk.items()}
        else:
            for n, v in gk.items():
                acc[n] += v
    del br, rt
    assert abs(loss - loss_sum / N) < 1e-4 * max(1.0, abs(loss_sum / N)), (loss, loss_sum / N)
    assert len(acc) == 42
    errs = {n: float(np.abs(g[n] - acc[n] / N).max() / (np.abs(acc[n] / N).max() + 1e-30)) for n in acc}
    ranked = sorted(errs.items(), key=lambda kv: -kv[1])
    print("config 3 gradients at 16 x 1024x512 vs the mean of 16 per-image oracle gradients: %d ReLU units (largest %.1e) and %d pool routes (largest gap %.1e) "
          "differ from the oracle's own, all coin flips; error / max per tensor, worst five:" % (n_relu, worst_relu, n_route, worst_gap),
          ", ".join("%s %.2e" % kv for kv in ranked[:5]), "; median %.2e" % float(np.median(list(errs.values()))))
    for n, err in ranked:
        assert err < 1e-4, (n, err)


@pytest.mark.parametrize("variant,options,k_logit,k_grad", [
    # (what runs, library options, allowed logit-error ratio, allowed gradient-error ratio -- device vs float64 over fp32 oracle vs float64)
    ("F(6x6) for all twelve 3x3 layers, F(4x4,4x4) for fc6 (the default)", {}, 2.0, 2.5),
    ("direct convolution everywhere (summation order is the only difference from the fp32 oracle)", {"winograd_min_cin": 0, "winograd_fc6": 0}, 2.0, 1.5)])
def test_device_is_as_close_to_float64_as_the_fp32_cpu_path(variant, options, k_logit, k_grad):
    """The one parity statement that needs neither the alignment of ReLU / pool decisions nor trust in an fp32 oracle (VERDICT round 4
    item 6): the oracle run in FLOAT64 at 1024x512 is the truth of the graph the reference defines (fcn8s_tensorflow.py:154-259), the
    fp32 oracle stands where the reference's fp32 CPU path stood, and the device must be about as close to the truth as that path is --
      logits (O(1-10) decoder):  max |device - f64|  <=  k_logit x max |fp32 oracle - f64|   and  <= 1e-3 absolute (north_star);
      argmax (fcn8s_tensorflow.py:268-269): device == f64 wherever the f64 top-2 margin exceeds 2e-3, and the device differs from the f64
              argmax on at most as many pixels as the fp32 oracle does, + 8;
      gradients (x30 decoder, all 42 tensors, UNALIGNED -- every side takes its own ReLU / pool decisions):
              worst tensor and median tensor of max |device - f64| / max |f64|  <=  k_grad x the same two figures of the fp32 oracle.
    Measured (profiles/parity_r05.json): F(6x6) logits 9.0e-5 against the fp32 oracle's 5.4e-5 (1.65x), 5 differing pixels against 5;
    gradients 1.57e-3 against 0.94e-3 worst tensor (1.67x), 3.6e-4 against 1.5e-4 median (2.35x) -- the unaligned distance is a few dozen
    pool windows whose two maxima agree to round-off, DESIGN section 2, and either side has its own such windows.  Direct convolution:
    logits 8.1e-5 (1.49x: the summation order of a 128-row-tile GEMM against oneDNN's), 3 pixels, gradients 0.95e-3 (1.01x), median 0.82x."""
    import torch
    from fcn8s_tensorflow_amd.engine import Engine
    torch.set_num_threads(min(32, torch.get_num_threads()))
    H, W, C = 512, 1024, 20
    img, lab = orc.synthetic_batch(1, H, W)
    onehot = orc.one_hot(lab, C).astype(np.float32)
    e = Engine(C, options=options)
    # logits / argmax
    P = orc.init_params(C, seed=1, decoder_std_scale=5.0, bias_std=0.05)
    e.set_params(P)
    pred = e.predict(img, argmax=True)
    lg = e.activation("logits", (1, H, W, C)).astype(np.float64)
    l64 = orc.forward(P, img, dtype=torch.float64)
    l32 = orc.forward(P, img).astype(np.float64)
    e_dev, e_o32 = float(np.abs(lg - l64).max()), float(np.abs(l32 - l64).max())
    a64 = np.argmax(orc.softmax(l64), -1)
    a32 = np.argmax(orc.softmax(l32.astype(np.float32)), -1)
    srt = np.sort(l64, -1)
    safe = (srt[..., -1] - srt[..., -2]) > 2e-3
    m_dev, m_o32 = int((pred != a64).sum()), int((a32 != a64).sum())
    print("[%s] logits vs float64: device %.3e, fp32 oracle %.3e (ratio %.2f), max |logit| %.1f; argmax differs from float64's on %d pixels (fp32 oracle: %d), %d above the margin"
          % (variant, e_dev, e_o32, e_dev / e_o32, float(np.abs(l64).max()), m_dev, m_o32, int((pred != a64)[safe].sum())))
    assert 0.5 < float(np.abs(l64).max()) < 50.0
    assert e_dev < 1e-3 and e_dev <= k_logit * e_o32, (e_dev, e_o32)
    assert safe.mean() > 0.95 and (pred[safe] == a64[safe]).all()
    assert m_dev <= m_o32 + 8, (m_dev, m_o32)
    # gradients, unaligned
    Pg = orc.init_params(C, seed=4, decoder_std_scale=30.0, bias_std=0.05)
    e.set_params(Pg)
    loss = e.forward_backward(img, lab, keep_prob=1.0, l2_rate=1e-3)
    g = e.get_grads()
    e.close()
    loss64, g64, _ = orc.loss_and_grads(Pg, img, onehot, l2_rate=1e-3, dtype=torch.float64)
    loss32, g32, _ = orc.loss_and_grads(Pg, img, onehot, l2_rate=1e-3)
    rel = lambda a, b: float(np.abs(np.asarray(a, np.float64) - b).max() / (np.abs(b).max() + 1e-300))
    d_dev = {k: rel(g[k], g64[k]) for k in g64}
    d_o32 = {k: rel(g32[k], g64[k]) for k in g64}
    w_dev, w_o32 = max(d_dev.values()), max(d_o32.values())
    med_dev, med_o32 = float(np.median(list(d_dev.values()))), float(np.median(list(d_o32.values())))
    print("[%s] gradients vs float64 (42 tensors, every side on its own ReLU / pool decisions): worst tensor device %.2e (%s), fp32 oracle %.2e (%s), ratio %.2f; "
          "median device %.2e, fp32 oracle %.2e, ratio %.2f; loss device %.7f fp32 oracle %.7f float64 %.7f"
          % (variant, w_dev, max(d_dev, key=d_dev.get), w_o32, max(d_o32, key=d_o32.get), w_dev / w_o32, med_dev, med_o32, med_dev / med_o32, loss, loss32, loss64))
    assert abs(loss - loss64) <= max(2.0 * abs(loss32 - loss64), 1e-5 * abs(loss64)), (loss, loss32, loss64)
    assert w_dev <= k_grad * w_o32, (w_dev, w_o32)
    assert med_dev <= max(k_grad, 3.0) * med_o32, (med_dev, med_o32)


def test_config3_deterministic_mode_is_bit_reproducible_at_bs16():
    """Option `deterministic`: every reduction the default path splits over blocks and joins with fp32 atomics -- the
    Winograd-domain and plain weight gradients, conv1_1's and the score heads' weight gradients, the last bias gradient from the loss kernel --
    writes one partial slab per split and adds the slabs in split order.  At BASELINE config 3's size (16 x 1024x512, TF-Adam, keep_prob 0.5,
    L2 on): two FRESH engines end four training steps with bit-identical parameters, Adam slots included in what the steps computed; the
    gradients equal the default path's to summation-order round-off; and the default path itself is shown not to be reproducible (or is, by
    luck -- printed, not asserted)."""
    import torch
    from fcn8s_tensorflow_amd.engine import Engine
    from fcn8s_tensorflow_amd import _lib as L
    N, H, W, C = 16, 512, 1024, 20
    img, lab = orc.synthetic_batch(N, H, W)
    imgd, labd = torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda()
    P = orc.init_params(C, seed=4, decoder_std_scale=30.0, bias_std=0.05)

    def four_steps(det):
        e = Engine(C, seed=3, options={"deterministic": det})
        e.set_params(P)
        losses = [e.train_step(imgd, labd, 1e-4, keep_prob=0.5, l2_rate=1e-3, optimizer=L.OPT_TF_ADAM)[0] for _ in range(4)]
        params = e.flat_params.clone()
        e.forward_backward(imgd, labd, keep_prob=1.0, l2_rate=1e-3)
        grads = e.flat_grads.clone()
        specs = dict(e.specs)
        e.close()
        return losses, params, grads, specs

    a = four_steps(1)
    b = four_steps(1)
    assert a[0] == b[0], (a[0], b[0])
    assert torch.equal(a[1], b[1]) and torch.equal(a[2], b[2])
    c = four_steps(0)
    d = four_steps(0)
    print("deterministic mode: 4 TF-Adam steps at 16 x 1024x512 on two fresh engines -> identical losses %s and bit-identical parameters / gradients; "
          "default mode: parameters of two runs %s" % (a[0], "identical as well (by luck)" if torch.equal(c[1], d[1]) else
          "differ in %d of %d elements (largest difference %.2e)" % (int((c[1] != d[1]).sum()), c[1].numel(), float((c[1] - d[1]).abs().max()))))
    worst = ("", 0.0)
    for name, (shape, off) in a[3].items():
        n = int(np.prod(shape))
        x, y = a[2][off:off + n], c[2][off:off + n]
        err = float((x - y).abs().max()) / (float(y.abs().max()) + 1e-30)
        if err > worst[1]:
            worst = (name, err)
    print("deterministic vs default gradients after the same 4 steps: worst tensor %s %.2e of its largest entry" % worst)
    assert np.allclose(a[0], c[0], rtol=3e-3), (a[0], c[0])          # (four TF-Adam steps from a x30 decoder: a chaotic trajectory, the modes share it to a few 1e-4)


def test_config3_fused_forward_transforms_at_bs16_equal_the_two_kernel_form():
    """The default engine fuses each inner conv's output transform with the next conv's input transform (`fuse_out_in` = 1).  At BASELINE's
    size (16 x 1024x512), against an engine with the fusion off: the kernel really
    runs (five launches per pass), loss and softmax are bit-identical, every gradient tensor agrees to the weight-gradient atomics' order."""
    import torch
    from fcn8s_tensorflow_amd.engine import Engine
    N, H, W, C = 16, 512, 1024, 20
    img, lab = orc.synthetic_batch(N, H, W)
    imgd, labd = torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda()
    got = []
    for fuse in (1, 0):
        e = Engine(C, seed=3, options={"fuse_out_in": fuse})
        e.init_params(seed=1)
        e.profile(2); e.profile_reset()
        loss = e.forward_backward(imgd, labd, keep_prob=1.0)
        prof = e.profile_results()
        e.profile(0)
        n_fused = sum(int(v["launches"]) for k, v in prof.items() if k.startswith("kernel:") and "wino_out_in_kernel" in k)
        assert n_fused == (5 if fuse else 0), n_fused            # conv2_1, conv3_1, conv3_2, conv4_1, conv4_2 (conv5_x: six tile rows per image, ranges of three: left on two kernels)
        got.append((loss, e.flat_grads.clone(), e.predict(imgd[:2], argmax=False).clone(), dict(e.specs)))
        e.close()
    assert got[0][0] == got[1][0]
    assert torch.equal(got[0][2], got[1][2])
    worst = ("", 0.0)
    for name, (shape, off) in got[0][3].items():
        n = int(np.prod(shape))
        a, b = got[0][1][off:off + n], got[1][1][off:off + n]
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
        if err > worst[1]:
            worst = (name, err)
    print("fused vs two-kernel forward transforms at 16 x 1024x512: loss %.7f both, worst gradient difference %s %.2e" % (got[0][0], worst[0], worst[1]))
    assert worst[1] < 1e-4, worst


def test_config5_shape_2048x1024_bs4_properties():
    import torch
    from fcn8s_tensorflow_amd.engine import Engine
    N, H, W, C = 4, 1024, 2048, 20
    e = Engine(C, seed=5)
    img, lab = orc.synthetic_batch(N, H, W)
    imgd, labd = torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda()
    e.init_params(seed=0)
    zero = {k: np.zeros(s[0], np.float32) for k, s in e.specs.items() if "1x1" in k or "trans" in k}
    e.set_params(zero)
    loss = e.forward_backward(imgd, labd, keep_prob=1.0)
    assert abs(loss - np.log(C)) < 1e-5
    gb = e.grad_view("fc7_pool4_pool3_conv2d_trans/bias").cpu().numpy()
    assert np.abs(gb - (1.0 / C - np.bincount(lab.ravel(), minlength=C) / lab.size)).max() < 1e-6

    e.init_params(seed=2)
    e.forward_backward(imgd, labd, keep_prob=1.0)
    g_full = e.flat_grads.clone()
    e.forward_backward(imgd[:2], labd[:2], keep_prob=1.0)
    g_a = e.flat_grads.clone()
    e.forward_backward(imgd[2:], labd[2:], keep_prob=1.0)
    g_mean = 0.5 * (g_a + e.flat_grads)
    for name in ("conv1_2/filter", "conv4_1/filter", "fc6/weights", "fc7/biases", "fc7_1x1/kernel", "fc7_pool4_conv2d_trans/kernel"):
        shape, off = e.specs[name]
        n = int(np.prod(shape))
        a, b = g_full[off:off + n], g_mean[off:off + n]
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-12, name

    # a quarter of one image equals the same crop run on its own wherever the receptive field stays inside it:
    # not true for a conv net with SAME padding in general, so instead check translation of the batch axis --
    # image k of the batch gives the same prediction as image k alone
    full = e.predict(imgd, argmax=True)
    one = e.predict(imgd[3:4], argmax=True)
    assert float((torch.as_tensor(full[3:4]) != torch.as_tensor(one)).float().mean()) < 1e-4

    # config 5's arithmetic at config 5's size: forward fc6 / fc7 with bf16 operands (fp32 accumulate)
    sm32 = e.predict(imgd[:1], argmax=False).clone()
    e.set_precision('bf16_fc')
    sm16 = e.predict(imgd[:1], argmax=False)
    d = float((sm16 - sm32).abs().max())
    assert 0.0 < d < 2e-2, d                                  # operand rounding is visible and small
    e.forward_backward(imgd, labd, keep_prob=1.0)
    g_full = e.flat_grads.clone()
    e.forward_backward(imgd[:2], labd[:2], keep_prob=1.0)
    g_a = e.flat_grads.clone()
    e.forward_backward(imgd[2:], labd[2:], keep_prob=1.0)
    g_mean = 0.5 * (g_a + e.flat_grads)
    for name in ("conv5_3/filter", "fc6/weights", "fc7/weights", "fc7_1x1/kernel"):
        shape, off = e.specs[name]
        n = int(np.prod(shape))
        a, b = g_full[off:off + n], g_mean[off:off + n]
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-12, name
    loss, step = e.train_step(imgd, labd, 1e-4, keep_prob=0.5)
    assert np.isfinite(loss) and step == 1
    e.close()


@pytest.mark.parametrize("mode", ["bf16_fwd", "bf16_fwd_x2"])
def test_config5_bf16_fwd_2048x1024_bs4(mode):
    """BASELINE.json configs[4] in its own arithmetic at its own size: 2048x1024, 4 images per GPU, forward convolutions conv3_1 .. conv5_3,
    fc6 and fc7 on the bf16 MFMA with fp32 accumulation (`bf16_fwd`; `bf16_fwd_x2` = the same forward with the remaining GEMMs on two bf16
    pieces per operand).  The CPU oracle cannot run the whole graph at this size inside a test, so:
      (i)   each bf16 layer of image 3 of the batch is compared with the oracle's convolution of the bf16-rounded operands applied to the
            DEVICE's own input of that layer (identical rounding; only the fp32 summation order differs): conv3_1, conv4_2, conv5_3, fc6,
            fc7 at <= 1e-4 of the layer's largest output;
      (ii)  closed-form loss / last-bias gradient at zero decoder weights, and batch linearity of the gradients;
      (iii) image k of the batch predicts what image k predicts alone."""
    import torch
    from fcn8s_tensorflow_amd.engine import Engine
    N, H, W, C = 4, 1024, 2048, 20
    e = Engine(C, seed=5, precision=mode)
    img, lab = orc.synthetic_batch(N, H, W)
    imgd, labd = torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda()
    P = orc.init_params(C, seed=6, decoder_std_scale=6.0, bias_std=0.05)
    e.set_params(P)
    e.profile(2); e.profile_reset()
    loss = e.forward_backward(imgd, labd, keep_prob=1.0)
    prof = e.profile_results()
    e.profile(0)
    assert np.isfinite(loss)
    k256 = [v for k, v in prof.items() if k.startswith("kernel:") and "conv_bf16_256_kernel" in k]
    assert k256 and sum(int(v["launches"]) for v in k256) >= 11, {k: v["launches"] for k, v in prof.items() if k.startswith("kernel:")}
    mine, other = ("_x3_kernel", "_x2_kernel") if mode == "bf16_fwd" else ("_x2_kernel", "_x3_kernel")
    assert any(mine in k for k in prof) and not any(k.startswith("kernel:") and ("gemm_glds_kernel<" in k or "wgrad_glds_kernel<" in k or other in k) for k in prof)

    # (i) per-layer arithmetic, image 3 of the batch
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
    wd = e.widths
    K = 3
    shapes = {"pool2": (N, H // 4, W // 4, wd[1]), "conv3_1": (N, H // 4, W // 4, wd[2]), "conv4_1": (N, H // 8, W // 8, wd[3]), "conv4_2": (N, H // 8, W // 8, wd[3]),
              "conv5_2": (N, H // 16, W // 16, wd[4]), "conv5_3": (N, H // 16, W // 16, wd[4]), "pool5": (N, H // 32, W // 32, wd[4]),
              "fc6": (N, H // 32, W // 32, wd[5]), "fc7": (N, H // 32, W // 32, wd[6])}
    torch.set_num_threads(min(32, torch.get_num_threads()))
    worst = {}
    for src, dst, wname, bname in (("pool2", "conv3_1", "conv3_1/filter", "conv3_1/biases"), ("conv4_1", "conv4_2", "conv4_2/filter", "conv4_2/biases"),
                                   ("conv5_2", "conv5_3", "conv5_3/filter", "conv5_3/biases"), ("pool5", "fc6", "fc6/weights", "fc6/biases"),
                                   ("fc6", "fc7", "fc7/weights", "fc7/biases")):
        x = torch.from_numpy(e.activation(src, shapes[src])[K:K + 1]).permute(0, 3, 1, 2)
        wk = torch.from_numpy(P[wname]); k = wk.shape[0]
        want = torch.relu(torch.nn.functional.conv2d(rb(x), rb(wk).permute(3, 2, 0, 1), torch.from_numpy(P[bname]), padding=(k - 1) // 2)).permute(0, 2, 3, 1).numpy()
        got = e.activation(dst, shapes[dst])[K:K + 1]
        assert np.abs(want).max() > 0
        worst[dst] = float(np.abs(got - want).max() / np.abs(want).max())
        del x, want, got
    print("config 5 [%s] 2048x1024 x 4: per-layer error against the same-rounding convolution of the device's own input:" % mode,
          ", ".join("%s %.2e" % kv for kv in worst.items()))
    for k, v in worst.items():
        assert v < 1e-4, (k, v)

    # (iii) image k of the batch equals image k alone -- up to what bf16 operand rounding does to fp32 summation-order differences: a batch of
    #       one takes other tile shapes (and F(4x4) for conv1_2 / conv2_x is chosen per launch size), an activation that differs in its last
    #       fp32 bit can round to the other bf16 neighbour (a 2^-8 step), so the logits agree to 7e-3 of their scale (measured), not to
    #       fp32 round-off; the argmax must agree wherever the top-2 margin exceeds twice the bound
    full = np.asarray(torch.as_tensor(e.predict(imgd, argmax=True)).cpu())[K]
    lg_full = e.activation("logits", (N, H, W, C))[K].copy()
    one = np.asarray(torch.as_tensor(e.predict(imgd[K:K + 1], argmax=True)).cpu())[0]
    lg_one = e.activation("logits", (1, H, W, C))[0]
    scale = max(1.0, float(np.abs(lg_one).max()))
    d = float(np.abs(lg_full - lg_one).max()) / scale
    srt = np.sort(lg_one, -1)
    safe = (srt[..., -1] - srt[..., -2]) > 4e-2 * scale
    print("config 5 [%s]: image %d in the batch vs alone: logits differ by %.2e of their scale, argmax on %d of %d pixels (%d of them above the margin)"
          % (mode, K, d, int((full != one).sum()), one.size, int((full != one)[safe].sum())))
    assert d < 2e-2, d                                        # measured 7.2e-3 .. 7.4e-3
    assert safe.mean() > 0.1 and (full[safe] == one[safe]).all()
    del lg_full, lg_one, srt

    # (ii) closed forms and batch linearity
    zero = {k: np.zeros(s[0], np.float32) for k, s in e.specs.items() if "1x1" in k or "trans" in k}
    e.set_params(zero)
    loss = e.forward_backward(imgd, labd, keep_prob=1.0)
    assert abs(loss - np.log(C)) < 1e-5
    gb = e.grad_view("fc7_pool4_pool3_conv2d_trans/bias").cpu().numpy()
    assert np.abs(gb - (1.0 / C - np.bincount(lab.ravel(), minlength=C) / lab.size)).max() < 1e-6
    assert float(e.grad_view("conv1_1/filter").abs().max()) == 0.0 and float(e.grad_view("fc6/weights").abs().max()) == 0.0
    e.set_params(P)
    e.forward_backward(imgd, labd, keep_prob=1.0)
    g_full = e.flat_grads.clone()
    e.forward_backward(imgd[:2], labd[:2], keep_prob=1.0)
    g_a = e.flat_grads.clone()
    e.forward_backward(imgd[2:], labd[2:], keep_prob=1.0)
    g_mean = 0.5 * (g_a + e.flat_grads)
    for name in ("conv1_2/filter", "conv3_1/filter", "conv4_2/filter", "conv5_3/filter", "fc6/weights", "fc7/weights", "fc7/biases", "fc7_1x1/kernel",
                 "fc7_pool4_pool3_conv2d_trans/kernel"):
        shape, off = e.specs[name]
        n = int(np.prod(shape))
        a, b = g_full[off:off + n], g_mean[off:off + n]
        assert float(b.abs().max()) > 0
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-12, (name, float((a - b).abs().max()) / float(b.abs().max()))
    loss, step = e.train_step(imgd, labd, 1e-4, keep_prob=0.5)
    assert np.isfinite(loss) and step == 1
    e.close()


def test_config5_bf16_train_2048x1024_bs4():
    """BASELINE.json configs[4]'s shape (2048x1024, 4 images per GPU) in FCN8S_PREC_BF16_TRAIN: bf16-rounded operands in the forward pass AND in both
    gradients of conv1_2 .. conv5_3, fc6, fc7 (direct convolutions; no Winograd transform runs).  The CPU oracle cannot run this size inside a
    test, so, as for bf16_fwd: (i) the forward arithmetic of five layers of image 3 of the batch against the same-rounding convolution of the
    device's own layer input (1e-4); (ii) closed-form loss / last-bias gradient at zero decoder weights; batch linearity of the gradients --
    halving the batch doubles every dY exactly, and rounding to bf16 commutes with a power of two, so the bf16 gradients are as linear in the
    batch as the fp32 ones (1e-3); (iii) image 3 of the batch predicts what image 3 predicts alone."""
    import torch
    from fcn8s_tensorflow_amd.engine import Engine
    N, H, W, C = 4, 1024, 2048, 20
    e = Engine(C, seed=5, precision="bf16_train")
    img, lab = orc.synthetic_batch(N, H, W)
    imgd, labd = torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda()
    P = orc.init_params(C, seed=6, decoder_std_scale=6.0, bias_std=0.05)
    e.set_params(P)
    e.profile(2); e.profile_reset()
    loss_default = e.forward_backward(imgd, labd, keep_prob=1.0)
    prof = e.profile_results()
    e.profile(0)
    lg_default = e.activation("logits", (N, H, W, C))[3].copy()
    with pytest.raises(Exception, match="bf16_acts"):           # (the default keeps a conv -> conv activation only as the consumer's padded bf16 copy)
        e.activation("conv3_1", (N, H >> 2, W >> 2, e.widths[2]))
    # the same pass with every activation kept as an fp32 tensor too (what (i) looks at): the same logits bit for bit
    e.set_option("bf16_acts", 0)
    loss = e.forward_backward(imgd, labd, keep_prob=1.0)
    np.testing.assert_array_equal(e.activation("logits", (N, H, W, C))[3], lg_default)
    assert abs(loss - loss_default) <= 1e-6 * abs(loss)
    del lg_default
    assert np.isfinite(loss)
    kernels = {k[7:]: int(v["launches"]) for k, v in prof.items() if k.startswith("kernel:")}
    assert sum(v for k, v in kernels.items() if "conv_bf16_" in k) == 28 and sum(v for k, v in kernels.items() if "wgrad_bf16" in k) == 14, kernels
    assert not any("wino" in k or "gemm_glds_nt" in k or "_x2_" in k or "_x3_" in k for k in kernels), kernels      # (gemm_glds / wgrad_glds: the last transposed conv, exact fp32)
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
    wd = e.widths
    K = 3
    torch.set_num_threads(min(32, torch.get_num_threads()))
    worst = {}
    for src, dst, d, cs, cd, wname, bname in (("pool1", "conv2_1", 1, wd[0], wd[1], "conv2_1/filter", "conv2_1/biases"), ("conv3_1", "conv3_2", 2, wd[2], wd[2], "conv3_2/filter", "conv3_2/biases"),
                                                ("conv5_2", "conv5_3", 4, wd[4], wd[4], "conv5_3/filter", "conv5_3/biases"), ("pool5", "fc6", 5, wd[4], wd[5], "fc6/weights", "fc6/biases"),
                                                ("fc6", "fc7", 5, wd[5], wd[6], "fc7/weights", "fc7/biases")):
        x = torch.from_numpy(e.activation(src, (N, H >> d, W >> d, cs))[K:K + 1]).permute(0, 3, 1, 2)
        wk = torch.from_numpy(P[wname]); k = wk.shape[0]
        want = torch.relu(torch.nn.functional.conv2d(rb(x), rb(wk).permute(3, 2, 0, 1), torch.from_numpy(P[bname]), padding=(k - 1) // 2)).permute(0, 2, 3, 1).numpy()
        got = e.activation(dst, (N, H >> d, W >> d, cd))[K:K + 1]
        assert np.abs(want).max() > 0
        worst[dst] = float(np.abs(got - want).max() / np.abs(want).max())
        del x, want, got
    print("config 5 [bf16_train] 2048x1024 x 4: per-layer forward error against the same-rounding convolution of the device's own input:", ", ".join("%s %.2e" % kv for kv in worst.items()))
    for k, v in worst.items():
        assert v < 1e-4, (k, v)
    e.set_option("bf16_acts", 1)
    full = np.asarray(torch.as_tensor(e.predict(imgd, argmax=True)).cpu())[K]
    lg_full = e.activation("logits", (N, H, W, C))[K].copy()
    one = np.asarray(torch.as_tensor(e.predict(imgd[K:K + 1], argmax=True)).cpu())[0]
    lg_one = e.activation("logits", (1, H, W, C))[0]
    scale = max(1.0, float(np.abs(lg_one).max()))
    d = float(np.abs(lg_full - lg_one).max()) / scale
    srt = np.sort(lg_one, -1)
    safe = (srt[..., -1] - srt[..., -2]) > 4e-2 * scale
    print("config 5 [bf16_train]: image %d in the batch vs alone: logits differ by %.2e of their scale, argmax on %d of %d pixels (%d above the margin)"
          % (K, d, int((full != one).sum()), one.size, int((full != one)[safe].sum())))
    assert d < 2e-2, d                                  # (direct convolutions: the row tiles of image 3 differ between the two launches, nothing else)
    assert safe.mean() > 0.1 and (full[safe] == one[safe]).all()
    del lg_full, lg_one, srt
    zero = {k: np.zeros(s[0], np.float32) for k, s in e.specs.items() if "1x1" in k or "trans" in k}
    e.set_params(zero)
    loss = e.forward_backward(imgd, labd, keep_prob=1.0)
    assert abs(loss - np.log(C)) < 1e-5
    gb = e.grad_view("fc7_pool4_pool3_conv2d_trans/bias").cpu().numpy()
    assert np.abs(gb - (1.0 / C - np.bincount(lab.ravel(), minlength=C) / lab.size)).max() < 1e-6
    assert float(e.grad_view("conv1_1/filter").abs().max()) == 0.0 and float(e.grad_view("fc6/weights").abs().max()) == 0.0
    e.set_params(P)
    e.forward_backward(imgd, labd, keep_prob=1.0)
    g_full = e.flat_grads.clone()
    e.forward_backward(imgd[:2], labd[:2], keep_prob=1.0)
    g_a = e.flat_grads.clone()
    e.forward_backward(imgd[2:], labd[2:], keep_prob=1.0)
    g_mean = 0.5 * (g_a + e.flat_grads)
    for name in ("conv1_1/filter", "conv1_2/filter", "conv2_1/filter", "conv3_1/filter", "conv4_2/filter", "conv5_3/filter", "fc6/weights", "fc7/weights", "fc7/biases", "fc7_1x1/kernel",
                 "fc7_pool4_pool3_conv2d_trans/kernel"):
        shape, off = e.specs[name]
        n = int(np.prod(shape))
        a, b = g_full[off:off + n], g_mean[off:off + n]
        assert float(b.abs().max()) > 0
        assert float((a - b).abs().max()) <= 1e-3 * float(b.abs().max()) + 1e-12, (name, float((a - b).abs().max()) / float(b.abs().max()))
    loss, step = e.train_step(imgd, labd, 1e-4, keep_prob=0.5)
    assert np.isfinite(loss) and step == 1
    e.close()


def test_config5_bf16_train_backward_2048x1024_bs4_layer_by_layer():
    """The BACKWARD pass of bf16_train at BASELINE.json configs[4]'s shape, layer by layer (the forward pass had such a check, the
    backward pass only batch linearity and closed forms, which a gradient that is wrong CONSISTENTLY at this size -- row splits, offsets near the guard rows,
    the 64 x 128 nine-tap form, fc6's split-K slabs -- would pass).  One forward / backward pass of the 4-image batch with every fp32 tensor kept
    (`bf16_acts` = 0, `bf16_fuse_pool` = 0: plain pools) and every layer's output gradient captured (`keep_output_gradients`); then, for conv1_2, conv2_2, conv4_2, fc6 and fc7:
      * dW of the layer (all four images) against torch's `conv2d_weight` of the device's OWN bf16-rounded layer input and bf16-rounded dY (fp32 on the CPU);
      * dX of the layer for image 3 -- which the device hands to the previous layer as ITS dY, behind that layer's ReLU (and, for fc6, behind pool5 and conv5_3's
        ReLU) -- against torch's `conv2d_input` of the same rounded dY and rounded weights;
    bound 1e-4 of the tensor's largest entry (fp32 summation order is all that differs).  And the default configuration (`bf16_acts` = 1: conv -> conv tensors
    only as bf16 copies) gives the same 42 gradient tensors as `bf16_acts` = 0 to 1e-5 (same values into every product; the split sums meet in atomics)."""
    import torch
    import torch.nn.functional as F
    from fcn8s_tensorflow_amd.engine import Engine
    N, H, W, C = 4, 1024, 2048, 20
    K = 3                                                   # the image whose data gradients are checked
    e = Engine(C, seed=5, precision="bf16_train")
    img, lab = orc.synthetic_batch(N, H, W)
    imgd, labd = torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda()
    P = orc.init_params(C, seed=6, decoder_std_scale=30.0, bias_std=0.05)
    e.set_params(P)
    e.forward_backward(imgd, labd, keep_prob=1.0)
    g_default = e.flat_grads.clone()
    e.set_option("bf16_acts", 0)
    e.forward_backward(imgd, labd, keep_prob=1.0)
    g_kept = e.flat_grads.clone()
    worst = ("", 0.0)
    for name, (shape, off) in e.specs.items():
        n = int(np.prod(shape))
        a, b = g_default[off:off + n], g_kept[off:off + n]
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
        if err > worst[1]:
            worst = (name, err)
    print("config 5 [bf16_train] backward: default configuration vs every fp32 tensor kept: worst gradient difference %s %.2e" % worst)
    assert worst[1] < 1e-5, worst
    del g_default, g_kept
    # (plain pools for the layer-by-layer part: the routed pools pick their maxima among the bf16 values and write the last conv's dY only as a bf16 copy -- another,
    #  equally valid subgradient at rounding ties, and no fp32 tensor to look at)
    e.set_option("bf16_fuse_pool", 0); e.set_option("keep_output_gradients", 1)
    e.forward_backward(imgd, labd, keep_prob=1.0)
    torch.set_num_threads(min(64, torch.get_num_threads()))
    rb = lambda t: t.to(torch.bfloat16).to(torch.float32)
    nchw = lambda a: torch.from_numpy(a).permute(0, 3, 1, 2)
    wd = e.widths
    grads = e.get_grads()
    report = {}
    #        layer     its input  level (stride 2^d)  Cin    Cout   weights          the layer before it (None: pool5 -> conv5_3)
    for layer, src, d, cin, cout, wname, prev in (("conv1_2", "conv1_1", 0, wd[0], wd[0], "conv1_2/filter", "conv1_1"),
                                                  ("conv2_2", "conv2_1", 1, wd[1], wd[1], "conv2_2/filter", "conv2_1"),
                                                  ("conv4_2", "conv4_1", 3, wd[3], wd[3], "conv4_2/filter", "conv4_1"),
                                                  ("fc6", "pool5", 5, wd[4], wd[5], "fc6/weights", None),
                                                  ("fc7", "fc6", 5, wd[5], wd[6], "fc7/weights", "fc6")):
        h, w = H >> d, W >> d
        wk = torch.from_numpy(P[wname]); k = wk.shape[0]; pad = (k - 1) // 2
        wt = rb(wk).permute(3, 2, 0, 1).contiguous()                       # [Cout][Cin][k][k]
        x = e.activation(src, (N, h, w, cin))
        dy = e.activation("dy:" + layer, (N, h, w, cout))
        assert np.abs(dy).max() > 0
        xr, dyr = rb(nchw(x)).contiguous(), rb(nchw(dy)).contiguous()
        dw_ref = torch.nn.grad.conv2d_weight(xr, wt.shape, dyr, padding=pad).permute(2, 3, 1, 0).numpy()
        dw_err = float(np.abs(grads[wname] - dw_ref).max() / np.abs(dw_ref).max())
        # dX of image K, as the previous layer received it
        dx_ref = torch.nn.grad.conv2d_input((1, cin, h, w), wt, dyr[K:K + 1], padding=pad)
        if prev is not None:
            got = e.activation("dy:" + prev, (N, h, w, cin))[K]
            want = (dx_ref[0].permute(1, 2, 0).numpy()) * (x[K] > 0)
        else:
            # pool5 between conv5_3 and fc6: route the reference d(pool5) through torch's max-pool backward of the device's own conv5_3, then conv5_3's ReLU
            c53 = torch.from_numpy(e.activation("conv5_3", (N, 2 * h, 2 * w, cin))[K:K + 1]).permute(0, 3, 1, 2).contiguous().requires_grad_(True)
            F.max_pool2d(c53, 2).backward(dx_ref)
            want = (c53.grad[0].permute(1, 2, 0).numpy()) * (c53.detach()[0].permute(1, 2, 0).numpy() > 0)
            got = e.activation("dy:conv5_3", (N, 2 * h, 2 * w, cin))[K]
        assert np.abs(want).max() > 0
        dx_err = float(np.abs(got - want).max() / np.abs(want).max())
        report[layer] = (dw_err, dx_err)
        del x, dy, xr, dyr, dw_ref, dx_ref, got, want
    print("config 5 [bf16_train] backward at 2048x1024 x 4 against torch on the device's own rounded operands (dW all images, dX image %d): " % K
          + ", ".join("%s dW %.2e dX %.2e" % (k, v[0], v[1]) for k, v in report.items()))
    for k, v in report.items():
        assert v[0] < 1e-4 and v[1] < 1e-4, (k, v)
    e.close()


def test_batch_of_64_is_its_four_images_sixteen_times():
    """Size independence at the top of the range (1024x512 x 64 images: 160 GB of the 288 GB, Winograd images of 4 G elements -- past
    every 32-bit index): a batch made of four images repeated sixteen times must give the loss, all 42 gradient tensors (the loss is a
    mean over the batch) and, image by image, the logits of the four-image batch.  A wrapped index anywhere shows up in the late images."""
    import torch
    from fcn8s_tensorflow_amd.engine import Engine
    R, REP, H, W, C = 4, 16, 512, 1024, 20
    P = orc.init_params(C, seed=4, decoder_std_scale=30.0, bias_std=0.05)
    img, lab = orc.synthetic_batch(R, H, W)
    imgd, labd = torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda()
    e = Engine(C)
    e.set_params(P)
    loss4 = e.forward_backward(imgd, labd, keep_prob=1.0, l2_rate=1e-3)
    g4 = e.flat_grads.clone()
    logits4 = e.activation("logits", (R, H, W, C))
    scale = float(np.abs(logits4).max())
    big_i, big_l = imgd.repeat(REP, 1, 1, 1).contiguous(), labd.repeat(REP, 1, 1).contiguous()
    loss64 = e.forward_backward(big_i, big_l, keep_prob=1.0, l2_rate=1e-3)
    assert abs(loss64 - loss4) < 1e-5 * max(1.0, abs(loss4)), (loss64, loss4)
    g64 = e.flat_grads
    worst = ("", 0.0)
    for name, (shape, off) in e.specs.items():
        n = int(np.prod(shape))
        a, b = g64[off:off + n], g4[off:off + n]
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
        if err > worst[1]:
            worst = (name, err)
    print("batch 64 vs its 4 images: loss %.7f / %.7f, worst gradient difference %s %.2e of the tensor's largest entry" % (loss64, loss4, worst[0], worst[1]))
    assert worst[1] < 1e-4, worst              # fp32 summation order only (16x more rows per reduction)
    logits64 = e.activation("logits", (R * REP, H, W, C))
    for i in (0, 5, 37, 62, 63):
        d = float(np.abs(logits64[i] - logits4[i % R]).max())
        assert d <= 1e-5 * scale, (i, d, scale)
    e.close()


def test_bf16_train_batch_of_64_is_its_four_images_sixteen_times():
    """bf16_train past 4 GiB per padded copy (64 x 1024x512: conv1_2's padded bf16 input is 64 x 514 x 1026 x 64 x 2 bytes = 4.3 GB, each of its two 32-channel
    planes 2.2 GB).  Round 5 refused this batch (32-bit byte offsets from the start of a copy); the convolution kernels now carry the tile's position in their
    64-bit base and the weight-gradient kernels' per-lane offsets span two planes.  A batch made of four images repeated sixteen times must give the loss of
    the four-image batch, its logits image by image, and -- the loss being a mean over the batch, and 1/16 a power of two that bf16 rounding commutes with --
    its gradients up to fp32 summation order (16x more rows per weight-gradient reduction).  A wrapped offset anywhere shows up in the late images."""
    import torch
    from fcn8s_tensorflow_amd.engine import Engine
    R, REP, H, W, C = 4, 16, 512, 1024, 20
    P = orc.init_params(C, seed=4, decoder_std_scale=30.0, bias_std=0.05)
    img, lab = orc.synthetic_batch(R, H, W)
    imgd, labd = torch.from_numpy(img).cuda(), torch.from_numpy(lab).cuda()
    e = Engine(C, precision="bf16_train")
    e.set_params(P)
    loss4 = e.forward_backward(imgd, labd, keep_prob=1.0, l2_rate=1e-3)
    g4 = e.flat_grads.clone()
    logits4 = e.activation("logits", (R, H, W, C))
    scale = float(np.abs(logits4).max())
    big_i, big_l = imgd.repeat(REP, 1, 1, 1).contiguous(), labd.repeat(REP, 1, 1).contiguous()
    loss64 = e.forward_backward(big_i, big_l, keep_prob=1.0, l2_rate=1e-3)
    assert abs(loss64 - loss4) < 1e-5 * max(1.0, abs(loss4)), (loss64, loss4)
    g64 = e.flat_grads
    worst = ("", 0.0)
    for name, (shape, off) in e.specs.items():
        n = int(np.prod(shape))
        a, b = g64[off:off + n], g4[off:off + n]
        err = float((a - b).abs().max()) / (float(b.abs().max()) + 1e-30)
        if err > worst[1]:
            worst = (name, err)
    print("bf16_train, batch 64 vs its 4 images: loss %.7f / %.7f, worst gradient difference %s %.2e of the tensor's largest entry" % (loss64, loss4, worst[0], worst[1]))
    assert worst[1] < 1e-3, worst
    logits64 = e.activation("logits", (R * REP, H, W, C))
    dmax = 0.0
    for i in (0, 5, 37, 62, 63):
        d = float(np.abs(logits64[i] - logits4[i % R]).max())
        dmax = max(dmax, d)
        assert d <= 1e-5 * scale, (i, d, scale)            # (measured 6.4e-7: the forward kernels add a dot product's terms in the same order whatever the batch)
    print("bf16_train, batch 64: logits of images 0, 5, 37, 62, 63 differ from the four-image batch's by at most %.2e of their scale" % (dmax / scale))
    loss, step = e.train_step(big_i, big_l, 1e-4, keep_prob=0.5)
    assert np.isfinite(loss) and step == 1
    e.close()
