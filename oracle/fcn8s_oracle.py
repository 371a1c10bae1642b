"""CPU oracle for the FCN-8s hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product.  Only ``tests/``,
``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
it.  Nothing under ``fcn8s_tensorflow_amd/`` imports it, and the product path
raises if the HIP library is missing instead of falling back to this code.

PARITY UNPINNED (for the floating-point graph): the reference
(`/root/reference/fcn8s_tensorflow.py`) computes this path with TensorFlow 1.x
kernels (unpinned; tutorial output shows 1.3.0, `fcn8s_tutorial.ipynb:311`)
and an un-vendored VGG-16 SavedModel (`README.md:42`).  Neither is available
offline and the reference holds no tests or golden vectors for the path, so
this restatement is pinned only by (a) the explicit-loop C restatement in
``oracle/fcn8s_oracle.c`` (known-answer tests in ``tests/test_oracle.py``),
(a') SciPy's correlate2d / convolve2d / special.log_softmax for the float building
blocks (``test_float_ops_against_scipy``: third-party code, not TensorFlow) and
(b) golden vectors captured from the reference modules that *do* import here
(``tests/golden/make_golden.py``: one-hot / ID-LUT helpers, the native
confusion-matrix C file, the BatchGenerator contract).

What follows which reference line:

* encoder topology         -> fcn8s_tensorflow.py:127-152 (+ SavedModel tensor
                              names :140-144, variable names :343-350)
* decoder                  -> fcn8s_tensorflow.py:154-237
* loss / Adam              -> fcn8s_tensorflow.py:239-259
* softmax / argmax         -> fcn8s_tensorflow.py:261-271
* streaming metrics        -> fcn8s_tensorflow.py:273-322
* feed values (keep_prob)  -> fcn8s_tensorflow.py:554-572, 685-689, 764-770

All tensors at this interface are NHWC numpy arrays; weights use the TF
layouts (conv: HWIO ``[kh,kw,Cin,Cout]``; conv2d_transpose:
``[kh,kw,Cout,Cin]``).  torch-CPU is used as the arithmetic engine (NCHW
internally) and ``torch.autograd`` provides the backward pass.
"""
from __future__ import annotations

import math
from collections import OrderedDict

import numpy as np
import torch
import torch.nn.functional as F

# [INFERRED] VGG mean in BGR order, subtracted inside the SavedModel's
# "Processing" scope (image fed as RGB at fcn8s_tensorflow.py:558).
VGG_MEAN_BGR = (103.939, 116.779, 123.68)

DEFAULT_WIDTHS = (64, 128, 256, 512, 512, 4096, 4096)  # c1..c5, fc6, fc7
CONVS_PER_BLOCK = (2, 2, 3, 3, 3)
POOL3_SCALE = 0.0001   # fcn8s_tensorflow.py:171
POOL4_SCALE = 0.01     # fcn8s_tensorflow.py:182

DECODER_KERNELS = ("pool3_1x1/kernel", "pool4_1x1/kernel", "fc7_1x1/kernel",
                   "fc7_conv2d_trans/kernel", "fc7_pool4_conv2d_trans/kernel",
                   "fc7_pool4_pool3_conv2d_trans/kernel")


def param_specs(num_classes=20, widths=DEFAULT_WIDTHS, fc6_ksize=7):
    """Ordered (name -> shape) table using the reference's variable names
    (fcn8s_tensorflow.py:331-350 and the VGG-16 naming they imply)."""
    specs = OrderedDict()
    cin = 3
    for blk, (nconv, cout) in enumerate(zip(CONVS_PER_BLOCK, widths[:5]), start=1):
        for i in range(1, nconv + 1):
            specs["conv%d_%d/filter" % (blk, i)] = (3, 3, cin, cout)
            specs["conv%d_%d/biases" % (blk, i)] = (cout,)
            cin = cout
    c3, c4, c5, f6, f7 = widths[2], widths[3], widths[4], widths[5], widths[6]
    C = num_classes
    specs["fc6/weights"] = (fc6_ksize, fc6_ksize, c5, f6)
    specs["fc6/biases"] = (f6,)
    specs["fc7/weights"] = (1, 1, f6, f7)
    specs["fc7/biases"] = (f7,)
    specs["pool3_1x1/kernel"] = (1, 1, c3, C)
    specs["pool3_1x1/bias"] = (C,)
    specs["pool4_1x1/kernel"] = (1, 1, c4, C)
    specs["pool4_1x1/bias"] = (C,)
    specs["fc7_1x1/kernel"] = (1, 1, f7, C)
    specs["fc7_1x1/bias"] = (C,)
    specs["fc7_conv2d_trans/kernel"] = (4, 4, C, C)
    specs["fc7_conv2d_trans/bias"] = (C,)
    specs["fc7_pool4_conv2d_trans/kernel"] = (4, 4, C, C)
    specs["fc7_pool4_conv2d_trans/bias"] = (C,)
    specs["fc7_pool4_pool3_conv2d_trans/kernel"] = (16, 16, C, C)
    specs["fc7_pool4_pool3_conv2d_trans/bias"] = (C,)
    return specs


def _trunc_normal(rng, shape, std):
    # tf.truncated_normal_initializer: redraw samples beyond 2 sigma.
    out = rng.standard_normal(shape)
    bad = np.abs(out) > 2.0
    while bad.any():
        out[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(out) > 2.0
    return (out * std).astype(np.float32)


def init_params(num_classes=20, widths=DEFAULT_WIDTHS, fc6_ksize=7, seed=0,
                decoder_std_scale=1.0, bias_std=0.0):
    """Synthetic weights: VGG convs He-normal, biases 0,
    decoder truncated-normal sigma 1e-3 (1x1) / 1e-2 (tconv)
    (fcn8s_tensorflow.py:159-160).  `decoder_std_scale` / `bias_std` let tests
    use livelier weights so that every path carries signal."""
    rng = np.random.default_rng(seed)
    params = OrderedDict()
    for name, shape in param_specs(num_classes, widths, fc6_ksize).items():
        if len(shape) == 1:
            params[name] = (rng.standard_normal(shape) * bias_std).astype(np.float32)
        elif name.endswith("/filter") or name.endswith("/weights"):
            fan_in = shape[0] * shape[1] * shape[2]
            params[name] = (rng.standard_normal(shape) * math.sqrt(2.0 / fan_in)).astype(np.float32)
        elif "trans" in name:
            params[name] = _trunc_normal(rng, shape, 0.01 * decoder_std_scale)
        else:
            params[name] = _trunc_normal(rng, shape, 0.001 * decoder_std_scale)
    return params


# ----------------------------------------------------------------------------
# Single ops (TF semantics restated; NHWC numpy in / out)
# ----------------------------------------------------------------------------

def _t(x, dtype):
    return torch.as_tensor(np.ascontiguousarray(x)).to(dtype)


def _nchw(x):
    return x.permute(0, 3, 1, 2)


def _nhwc(x):
    return x.permute(0, 2, 3, 1)


def preprocess_t(images_t):
    """uint8/float RGB NHWC -> float BGR minus VGG mean (K0, [INFERRED])."""
    mean = torch.tensor(VGG_MEAN_BGR, dtype=images_t.dtype)
    return images_t[..., [2, 1, 0]] - mean


def conv2d_same_t(x, w_hwio, b=None, relu=False):
    """tf.nn.conv2d(stride 1, SAME) + bias_add (+ relu); x NCHW torch."""
    k = w_hwio.shape[0]
    y = F.conv2d(x, w_hwio.permute(3, 2, 0, 1), b, padding=(k - 1) // 2)
    return F.relu(y) if relu else y


def maxpool2x2_t(x):
    """tf.nn.max_pool(2x2, stride 2, SAME); inputs here always have even H, W."""
    return F.max_pool2d(x, 2, 2, ceil_mode=True)


def conv2d_transpose_same_t(x, w_kkoi, b, stride):
    """tf.layers.conv2d_transpose(padding='same'): kernel [kh,kw,Cout,Cin],
    y[n,i*s+ky-p,j*s+kx-p,co] += x[n,i,j,ci]*W[ky,kx,co,ci], p=(k-s)//2,
    out = in*s (fcn8s_tensorflow.py:204-233; mapping validated in SURVEY 8c)."""
    k = w_kkoi.shape[0]
    return F.conv_transpose2d(x, w_kkoi.permute(3, 2, 0, 1), b, stride=stride,
                              padding=(k - stride) // 2)


def dropout_t(x, keep_prob, mask):
    """tf.nn.dropout: x / keep_prob * mask (mask in {0,1})."""
    if mask is None:
        if keep_prob != 1.0:
            raise ValueError("keep_prob != 1 needs an explicit mask (TF's RNG stream is not reproducible)")
        return x
    return x * (mask / keep_prob)


# ----------------------------------------------------------------------------
# The graph
# ----------------------------------------------------------------------------

def _bf16_round(t):
    """Round-to-nearest-even to bfloat16 and back (what v_cvt_pk_bf16_f32 does to an MFMA operand)."""
    return t.detach().to(torch.bfloat16).to(t.dtype)


def _note_relu(stats, name, z, on):
    """Alignment record for a parity test: how many units of `name` take another ReLU branch in `on` than this restatement would, and how
    far from zero the largest of them sits (relative to the layer's largest pre-activation): (count, units, largest |z| / max |z|)."""
    with torch.no_grad():
        d = (z > 0) != (on > 0)
        n = int(d.sum())
        stats[name] = (n, z.numel(), float(z[d].abs().max() / z.abs().max().clamp_min(1e-30)) if n else 0.0)


def _relu_branch(z, name, branches, stats=None):
    """ReLU -- or, when `branches` holds a 0/1 tensor for `name`, multiplication by that record of which units are on.
    Gradient parity is only defined where both sides take the same ReLU branches: a pre-activation that one side computes as
    -1e-7 of the layer's range and the other as +1e-7 (both inside fp32 round-off) switches one element of dZ on.  A parity test
    feeds the branches the device took (see `branches_from_activations`) and checks separately that they differ from this
    restatement's own only at units within round-off of zero."""
    if branches is not None and name in branches:
        if stats is not None:
            _note_relu(stats, name, z, branches[name])
        return z * branches[name]
    return F.relu(z)


def _fc_conv(x, w, b, bf16, name=None, branches=None, stats=None):
    """fc6 / fc7: SAME conv + bias + ReLU.  bf16: the forward VALUE is the contraction of the bf16-rounded operands
    (fp32 products and sums); the backward pass is the fp32 conv gradient taken with the unrounded operands -- the
    mode only changes the forward arithmetic (BASELINE config 5: "bf16 fwd / fp32 accum")."""
    z = conv2d_same_t(x, w, b)
    if bf16:
        z = z + (conv2d_same_t(_bf16_round(x), _bf16_round(w), b.detach()) - z).detach()
    return _relu_branch(z, name, branches, stats)


def _pool_routed(z, route, stats=None, name=None):
    """2x2/2 max-pool of relu(z) taken along recorded routes instead of this restatement's own argmax: route (N,C,h/2,w/2) int64 holds
    the window element (2*row + col) the gradient goes to, 4 = ReLU off.  max-pool's argmax is discontinuous where two window entries
    agree to round-off; a parity test feeds the routes the device took and checks separately (pool_routes) that they differ from this
    restatement's own only at such ties."""
    n, c, h, w = z.shape
    win = z.reshape(n, c, h // 2, 2, w // 2, 2).permute(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
    on = (route < 4).to(z.dtype)
    taken = torch.gather(win, -1, route.clamp(max=3).unsqueeze(-1)).squeeze(-1)
    if stats is not None:
        # (count of windows routed differently from this restatement's own rule -- first maximum, off unless > 0 --, windows, and the largest
        # distance from a tie among them: top - taken entry where both are on, |top| where only one side is, over the layer's largest |z|)
        with torch.no_grad():
            top, own = win.max(-1)
            own = torch.where(top > 0, own, torch.full_like(own, 4))
            d = own != route
            n = int(d.sum())
            both = d & (own < 4) & (route < 4)
            gap = torch.where(both, top - taken, top.abs())[d]
            stats[name] = (n, route.numel(), float(gap.max() / z.abs().max().clamp_min(1e-30)) if n else 0.0)
    return taken * on


def _conv_bf16_value(x, w, b):
    """SAME conv + bias whose forward VALUE is the contraction of the bf16-rounded operands (fp32 products and sums) and whose backward
    pass is the fp32 conv gradient at the unrounded operands -- the arithmetic of the library's bf16 modes (straight-through rounding)."""
    z = conv2d_same_t(x, w, b)
    return z + (conv2d_same_t(_bf16_round(x), _bf16_round(w), b.detach()) - z).detach()


class _ConvBf16Train(torch.autograd.Function):
    """SAME conv + bias in the arithmetic of the library's bf16_train mode (FCN8S_PREC_BF16_TRAIN, include/fcn8s_hip.h): BOTH operands of each of the
    three products are rounded to bfloat16 (round to nearest even), products and sums are fp32 --
        forward          y  = conv(bf16 x, bf16 w) + b
        data gradient    dx = conv^T(bf16 dy, bf16 w)
        weight gradient  dw = corr(bf16 x, bf16 dy)          bias gradient  db = sum dy   (exact)
    -- i.e. mixed-precision training on fp32 master weights (the rounding of x and w is straight-through)."""

    @staticmethod
    def forward(ctx, x, w_hwio, b):
        xr, wr = _bf16_round(x), _bf16_round(w_hwio).permute(3, 2, 0, 1).contiguous()
        k = w_hwio.shape[0]
        ctx.save_for_backward(xr, wr)
        ctx.k = k
        return F.conv2d(xr, wr, b, padding=(k - 1) // 2)

    @staticmethod
    def backward(ctx, dy):
        xr, wr = ctx.saved_tensors
        dyr, p = _bf16_round(dy), (ctx.k - 1) // 2
        dx = torch.nn.grad.conv2d_input(xr.shape, wr, dyr, padding=p)
        dw = torch.nn.grad.conv2d_weight(xr, wr.shape, dyr, padding=p)
        return dx, dw.permute(2, 3, 1, 0).contiguous(), dy.sum((0, 2, 3))


def _conv_bf16_train(x, w, b):
    return _ConvBf16Train.apply(x, w, b)


def forward_t(P, images_t, keep_prob=1.0, masks=None, keep=False, bf16_fc=False, branches=None, routes=None, bf16_convs=False, stats=None, bf16_train=False):
    """Forward pass on torch tensors.  P: name -> torch tensor (TF layouts).
    images_t: NHWC RGB float.  masks: optional (mask6, mask7) NHWC tensors.
    bf16_fc: BASELINE config 5 -- both operands of the fc6 / fc7 contractions rounded to bfloat16, fp32 accumulate.
    bf16_convs: the same for the forward convolutions conv3_1 .. conv5_3 (FCN8S_PREC_BF16_FWD; implies nothing about fc6 / fc7).
    bf16_train: FCN8S_PREC_BF16_TRAIN -- conv1_2 .. conv5_3, fc6 and fc7 with bf16-rounded operands in the forward pass AND in both gradients
    (_ConvBf16Train); overrides bf16_fc / bf16_convs.
    branches: optional name -> 0/1 NCHW tensor ("conv1_1" ... for the convs that feed another conv, "pool1".."pool5" for each block's
    last conv + pool, "fc6", "fc7"): the ReLU branches to take instead of this restatement's own (see _relu_branch).
    routes: optional "pool1".."pool5" -> int64 (N,C,h/2,w/2) tensor of max-pool routes (see _pool_routed); takes the place of that
    block's pool branch record (route 4 = off).
    stats: optional dict that receives, per name in `branches` / `routes`, how the imposed decisions differ from this restatement's own
    (see _note_relu / _pool_routed): what a test needs to assert that they differ at fp32 coin flips only.
    Returns logits NCHW (and the activation dict when keep=True)."""
    acts = OrderedDict()
    x = _nchw(preprocess_t(images_t))
    pools = {}
    for blk, nconv in enumerate(CONVS_PER_BLOCK, start=1):
        for i in range(1, nconv + 1):
            n = "conv%d_%d" % (blk, i)
            routed = routes is not None and ("pool%d" % blk) in routes
            pooled_branch = i == nconv and (routed or (branches is not None and ("pool%d" % blk) in branches))
            conv = _conv_bf16_train if (bf16_train and not (blk == 1 and i == 1)) else (_conv_bf16_value if (bf16_convs and blk >= 3) else conv2d_same_t)
            z = conv(x, P[n + "/filter"], P[n + "/biases"])
            # (a block's last conv: max(relu(z)) = relu(max(z)), so its branch record lives on the pooled tensor)
            x = z if pooled_branch else _relu_branch(z, n, branches, stats)
            if keep:
                acts[n] = F.relu(z) if pooled_branch else x
        if routed:
            x = _pool_routed(x, routes["pool%d" % blk], stats, "pool%d" % blk)
        else:
            x = maxpool2x2_t(x)
            if branches is not None and ("pool%d" % blk) in branches:
                if stats is not None:
                    _note_relu(stats, "pool%d" % blk, x, branches["pool%d" % blk])
                x = x * branches["pool%d" % blk]
        pools[blk] = x
        if keep:
            acts["pool%d" % blk] = x
    m6 = m7 = None
    if masks is not None:
        m6, m7 = (_nchw(m) for m in masks)
    if bf16_train:
        x = _relu_branch(_conv_bf16_train(x, P["fc6/weights"], P["fc6/biases"]), "fc6", branches, stats)
    else:
        x = _fc_conv(x, P["fc6/weights"], P["fc6/biases"], bf16_fc, "fc6", branches, stats)
    x = dropout_t(x, keep_prob, m6)
    if keep:
        acts["fc6"] = x
    if bf16_train:
        x = _relu_branch(_conv_bf16_train(x, P["fc7/weights"], P["fc7/biases"]), "fc7", branches, stats)
    else:
        x = _fc_conv(x, P["fc7/weights"], P["fc7/biases"], bf16_fc, "fc7", branches, stats)
    x = dropout_t(x, keep_prob, m7)
    if keep:
        acts["fc7"] = x
    # decoder (fcn8s_tensorflow.py:171-233)
    p3 = conv2d_same_t(pools[3] * POOL3_SCALE, P["pool3_1x1/kernel"], P["pool3_1x1/bias"])
    p4 = conv2d_same_t(pools[4] * POOL4_SCALE, P["pool4_1x1/kernel"], P["pool4_1x1/bias"])
    s7 = conv2d_same_t(x, P["fc7_1x1/kernel"], P["fc7_1x1/bias"])
    u1 = conv2d_transpose_same_t(s7, P["fc7_conv2d_trans/kernel"], P["fc7_conv2d_trans/bias"], 2)
    a4 = u1 + p4
    u2 = conv2d_transpose_same_t(a4, P["fc7_pool4_conv2d_trans/kernel"], P["fc7_pool4_conv2d_trans/bias"], 2)
    a3 = u2 + p3
    logits = conv2d_transpose_same_t(a3, P["fc7_pool4_pool3_conv2d_trans/kernel"],
                                     P["fc7_pool4_pool3_conv2d_trans/bias"], 8)
    if keep:
        acts.update(p3=p3, p4=p4, s7=s7, a4=a4, a3=a3, logits=logits)
        return logits, acts
    return logits


def total_loss_t(P, logits_nchw, labels_onehot_t, l2_rate):
    """approximation_loss + regularization_loss (fcn8s_tensorflow.py:250-254).
    softmax_cross_entropy_with_logits with dense labels: -sum_c l_c*logp_c,
    reduce_mean over N*H*W.  l2_regularizer(rate)(w) = rate * sum(w^2)/2."""
    logp = F.log_softmax(_nhwc(logits_nchw), dim=-1)
    ce = -(labels_onehot_t * logp).sum(-1).mean()
    reg = sum((P[n] ** 2).sum() for n in DECODER_KERNELS) * (0.5 * l2_rate)
    return ce + reg


def _params_t(params, dtype, requires_grad=False):
    return OrderedDict((k, _t(v, dtype).requires_grad_(requires_grad)) for k, v in params.items())


def forward(params, images, keep_prob=1.0, masks=None, dtype=torch.float32, keep=False, bf16_fc=False, bf16_convs=False, bf16_train=False):
    """numpy front-end.  Returns logits NHWC (and activations NHWC if keep)."""
    with torch.no_grad():
        P = _params_t(params, dtype)
        mt = None if masks is None else tuple(_t(m, dtype) for m in masks)
        out = forward_t(P, _t(images, dtype), keep_prob, mt, keep, bf16_fc, bf16_convs=bf16_convs, bf16_train=bf16_train)
        if keep:
            logits, acts = out
            return _nhwc(logits).contiguous().numpy(), {k: _nhwc(v).contiguous().numpy() for k, v in acts.items()}
        return _nhwc(out).contiguous().numpy()


def branch_layers():
    """Names whose ReLU branches `branches=` can carry: convs that feed another conv, each block's pool, fc6, fc7."""
    names = []
    for blk, nconv in enumerate(CONVS_PER_BLOCK, start=1):
        names += ["conv%d_%d" % (blk, i) for i in range(1, nconv)] + ["pool%d" % blk]
    return names + ["fc6", "fc7"]


def pool_routes(acts):
    """This restatement's own max-pool routes, from the NHWC activations of forward(..., keep=True): "pool1".."pool5" -> uint8
    (N,h/2,w/2,C), the window element (2*row + col) holding the FIRST maximum of the block's last (post-ReLU) conv output, 4 where that
    maximum is not > 0 -- the rule of TF's MaxPoolGrad on a ReLU output, and the encoding of fcn8s_get_pool_routing.  Second result:
    per block, the gap between the largest and second-largest window entry relative to the largest activation of that layer (0 = exact
    tie) -- the scale on which a convolution's round-off lives (its error is proportional to the magnitudes summed, not to the result)."""
    routes, gaps = OrderedDict(), OrderedDict()
    for blk, nconv in enumerate(CONVS_PER_BLOCK, start=1):
        x = np.asarray(acts["conv%d_%d" % (blk, nconv)])
        n, h, w, c = x.shape
        win = x.reshape(n, h // 2, 2, w // 2, 2, c).transpose(0, 1, 3, 5, 2, 4).reshape(n, h // 2, w // 2, c, 4)
        r = np.argmax(win, -1).astype(np.uint8)                      # numpy: first maximum
        top = win.max(-1)
        r[~(top > 0)] = 4
        srt = np.sort(win, -1)
        gaps["pool%d" % blk] = (srt[..., 3] - srt[..., 2]) / max(float(np.abs(x).max()), 1e-30)
        routes["pool%d" % blk] = r
    return routes, gaps


def loss_and_grads(params, images, labels_onehot, l2_rate=0.0, keep_prob=1.0, masks=None,
                   dtype=torch.float32, bf16_fc=False, branches=None, routes=None, bf16_convs=False, stats=None, bf16_train=False):
    """total_loss and d(total_loss)/d(every variable) -- what
    AdamOptimizer.minimize differentiates (var_list=None, :257).
    stats: optional dict filled with the alignment record of `branches` / `routes` (forward_t).
    branches: optional name -> NHWC bool/0-1 array of post-ReLU activations that are on (activation > 0), see forward_t.
    routes: optional "pool<b>" -> NHWC uint8 array of max-pool routes (the encoding of pool_routes), see forward_t."""
    P = _params_t(params, dtype, requires_grad=True)
    mt = None if masks is None else tuple(_t(m, dtype) for m in masks)
    bt = None if branches is None else {k: _nchw(_t(np.asarray(v) > 0, dtype)) for k, v in branches.items()}
    rt = None if routes is None else {k: torch.from_numpy(np.ascontiguousarray(np.asarray(v).transpose(0, 3, 1, 2))).to(torch.int64) for k, v in routes.items()}
    logits = forward_t(P, _t(images, dtype), keep_prob, mt, bf16_fc=bf16_fc, branches=bt, routes=rt, bf16_convs=bf16_convs, stats=stats, bf16_train=bf16_train)
    loss = total_loss_t(P, logits, _t(labels_onehot, dtype), l2_rate)
    grads = torch.autograd.grad(loss, list(P.values()))
    return (float(loss.detach()),
            OrderedDict((k, g.numpy()) for k, g in zip(P.keys(), grads)),
            _nhwc(logits.detach()).contiguous().numpy())


def softmax(logits):
    """tf.nn.softmax over the last axis (fcn8s_tensorflow.py:268)."""
    return F.softmax(torch.as_tensor(logits), dim=-1).numpy()


def predict(params, images, argmax=True, dtype=torch.float32):
    """FCN8s.predict (fcn8s_tensorflow.py:743-770): keep_prob 1.0; argmax of the
    *softmax output*, int64, lowest index wins ties."""
    sm = softmax(forward(params, images, 1.0, None, dtype))
    if not argmax:
        return sm
    return np.argmax(sm, axis=-1).astype(np.int64)   # np.argmax returns the first maximum


# ----------------------------------------------------------------------------
# Optimizers
# ----------------------------------------------------------------------------

def tf_adam_step(theta, g, m, v, t, lr, beta1=0.9, beta2=0.999, eps=1e-8):
    """tf.train.AdamOptimizer update (fcn8s_tensorflow.py:256): note epsilon sits
    outside the bias correction (differs from torch.optim.Adam).  `t` is the
    1-based step being applied.  float32 arithmetic throughout."""
    f = np.float32
    lr_t = f(lr) * f(math.sqrt(1.0 - beta2 ** t)) / f(1.0 - beta1 ** t)
    m = f(beta1) * m + f(1.0 - beta1) * g
    v = f(beta2) * v + f(1.0 - beta2) * g * g
    theta = theta - lr_t * m / (np.sqrt(v) + f(eps))
    return theta.astype(np.float32), m.astype(np.float32), v.astype(np.float32)


def sgd_momentum_step(theta, g, buf, lr, momentum=0.9):
    """tf.train.MomentumOptimizer: accum = momentum*accum + g; theta -= lr*accum
    (BASELINE.json config 3 names SGD+momentum; not used by the reference)."""
    f = np.float32
    buf = f(momentum) * buf + g
    return (theta - f(lr) * buf).astype(np.float32), buf.astype(np.float32)


# ----------------------------------------------------------------------------
# Streaming metrics (fcn8s_tensorflow.py:273-322)
# ----------------------------------------------------------------------------

def confusion_matrix(label_ids, pred_ids, num_classes):
    """conf[gt, pred] += 1 -- same arithmetic as
    cityscapesscripts/evaluation/addToConfusionMatrix_impl.c:10-16 and as the
    accumulator inside tf.metrics.mean_iou."""
    idx = label_ids.astype(np.int64).ravel() * num_classes + pred_ids.astype(np.int64).ravel()
    return np.bincount(idx, minlength=num_classes * num_classes).reshape(num_classes, num_classes).astype(np.int64)


def mean_iou_from_confusion(cm, valid_only=True):
    """tf.metrics.mean_iou: iou_c = diag / (row + col - diag), zero denominators
    replaced by 1; averaged over classes with non-zero denominator (later TF1.x)
    or over all classes (`valid_only=False`, TF 1.3 behaviour)."""
    cm = cm.astype(np.float64)
    diag = np.diag(cm)
    denom = cm.sum(0) + cm.sum(1) - diag
    iou = diag / np.where(denom > 0, denom, 1.0)
    if valid_only:
        nvalid = (denom > 0).sum()
        return float(iou.sum() / nvalid) if nvalid > 0 else 0.0
    return float(iou.mean())


def accuracy_from_confusion(cm):
    tot = cm.sum()
    return float(np.trace(cm) / tot) if tot > 0 else 0.0


class StreamingMetrics:
    """loss = mean over *batches* of total_loss (tf.metrics.mean, :284);
    mean_iou / accuracy from the accumulated confusion matrix (:291-301)."""

    def __init__(self, num_classes):
        self.C = num_classes
        self.reset()

    def reset(self):
        self.cm = np.zeros((self.C, self.C), np.int64)
        self.loss_sum = 0.0
        self.loss_cnt = 0

    def update(self, loss, label_ids, pred_ids):
        self.loss_sum += float(loss)
        self.loss_cnt += 1
        self.cm += confusion_matrix(label_ids, pred_ids, self.C)

    def values(self):
        return (self.loss_sum / max(self.loss_cnt, 1), mean_iou_from_confusion(self.cm),
                accuracy_from_confusion(self.cm))


def eval_step(params, images, labels_onehot, l2_rate=0.0, dtype=torch.float32):
    """One metric_update_ops run (fcn8s_tensorflow.py:685-689): keep_prob 1.0."""
    with torch.no_grad():
        P = _params_t(params, dtype)
        logits = forward_t(P, _t(images, dtype), 1.0, None)
        loss = float(total_loss_t(P, logits, _t(labels_onehot, dtype), l2_rate))
        sm = F.softmax(_nhwc(logits), dim=-1).numpy()
    pred = np.argmax(sm, -1).astype(np.int64)
    lab = np.argmax(labels_onehot, -1).astype(np.int64)     # labels_argmax, :280
    return loss, lab, pred


def one_hot(label_ids, num_classes):
    """helpers/ground_truth_conversion_utils.py:84-88: np.eye(C, dtype=bool)[image]."""
    return np.eye(num_classes, dtype=bool)[label_ids]


# ----------------------------------------------------------------------------
# Synthetic workload (BASELINE.md section 3 / SURVEY 8d)
# ----------------------------------------------------------------------------

def synthetic_batch(n, h, w, num_classes=20, rank=0):
    rng = np.random.default_rng(1234 + rank)
    images = rng.integers(0, 256, size=(n, h, w, 3), dtype=np.uint8)
    labels = rng.integers(0, num_classes, size=(n, h, w), dtype=np.uint8)
    return images, labels
