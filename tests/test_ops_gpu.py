"""Parity of every HIP kernel (called through the C ABI's op-level entry points)
against the CPU oracle on the same seeded inputs.  Tolerances are stated per test;
fp32 accumulation-order differences only."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import fcn8s_oracle as orc  # noqa: E402  (checker only)


def _lib():
    import os
    from fcn8s_tensorflow_amd import _lib
    # test_conv_ops_in_f32x3_mode_hold_the_fp32_tolerances re-runs this module with the op-level entry points switched to the split-bf16 arithmetic
    if os.environ.get("FCN8S_TEST_OP_F32X3") == "1":
        _lib.check(_lib.lib.fcn8s_set_option(None, b"op_f32x3", 1))
    return _lib


def dev(a):
    return torch.as_tensor(np.ascontiguousarray(a)).cuda()


def ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def rel_err(a, b):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


CONV_CASES = [
    # N, H, W, Cin, Cout, K
    (2, 8, 16, 4, 64, 3),
    (1, 32, 32, 64, 64, 3),
    (2, 16, 8, 64, 128, 3),
    (1, 8, 8, 128, 256, 3),
    (1, 6, 10, 32, 64, 7),
    (2, 4, 4, 256, 128, 1),
    (1, 16, 16, 256, 20, 1),
    (1, 3, 5, 8, 20, 3),
    (2, 4, 16, 64, 128, 7),     # 7x7 with W % 16 == 0, C % 64 == 0: the one-filter-row-per-block wgrad kernel
    (2, 8, 32, 128, 64, 3),     # 3x3, all-nine-taps wgrad kernel, two ci tiles
    # 1x1 score heads (skinny.hip): ragged row counts, the 8-wave K split, the K % 32 tail, and the 4-class build
    (1, 5, 7, 256, 20, 1),
    (2, 8, 8, 4096, 20, 1),
    (3, 11, 3, 512, 20, 1),
    (1, 9, 5, 96, 4, 1),
    (1, 4, 8, 2112, 20, 1),
    (1, 64, 128, 256, 20, 1),
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,K", CONV_CASES)
def test_conv2d_fwd_bwd(N, H, W, Cin, Cout, K):
    L = _lib()
    rng = np.random.default_rng(1)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((K, K, Cin, Cout)) / np.sqrt(K * K * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    dy = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    xt = torch.tensor(x).permute(0, 3, 1, 2).double().requires_grad_(True)
    wt = torch.tensor(w).double().requires_grad_(True)
    bt = torch.tensor(b).double().requires_grad_(True)
    yt = orc.conv2d_same_t(xt, wt, bt, relu=False)
    yt.backward(torch.tensor(dy).permute(0, 3, 1, 2).double())
    y_ref = yt.detach().permute(0, 2, 3, 1).numpy()
    dx_ref = xt.grad.permute(0, 2, 3, 1).numpy(); dw_ref = wt.grad.numpy(); db_ref = bt.grad.numpy()

    dx_, dw_, db_, y_ = torch.empty(N, H, W, Cin).cuda(), torch.empty(K, K, Cin, Cout).cuda(), torch.empty(Cout).cuda(), torch.empty(N, H, W, Cout).cuda()
    xd, wd, bd, dyd = dev(x), dev(w), dev(b), dev(dy)
    L.check(L.lib.fcn8s_op_conv2d(None, ptr(xd), ptr(wd), ptr(bd), ptr(y_), N, H, W, Cin, Cout, K, 0))
    L.check(L.lib.fcn8s_op_conv2d_bwd(None, ptr(xd), ptr(wd), ptr(dyd), ptr(dx_), ptr(dw_), ptr(db_), N, H, W, Cin, Cout, K))
    torch.cuda.synchronize()
    assert rel_err(y_.cpu().numpy(), y_ref) < 2e-5
    assert rel_err(dx_.cpu().numpy(), dx_ref) < 2e-5
    assert rel_err(dw_.cpu().numpy(), dw_ref) < 2e-5
    assert rel_err(db_.cpu().numpy(), db_ref) < 2e-5
    # ReLU epilogue
    L.check(L.lib.fcn8s_op_conv2d(None, ptr(xd), ptr(wd), ptr(bd), ptr(y_), N, H, W, Cin, Cout, K, 1))
    assert rel_err(y_.cpu().numpy(), np.maximum(y_ref, 0)) < 2e-5


@pytest.mark.parametrize("N,H,W,Cin,Cout,tile", [(1, 4, 6, 16, 32, 2), (2, 8, 16, 64, 128, 2), (1, 16, 32, 256, 256, 2), (3, 2, 2, 32, 64, 2),
                                                  (1, 4, 8, 16, 32, 4), (2, 8, 16, 64, 128, 4), (1, 16, 32, 256, 256, 4), (2, 4, 4, 512, 64, 4),
                                                  (1, 6, 12, 16, 32, 6), (2, 8, 16, 64, 128, 6), (1, 32, 64, 256, 256, 6), (3, 2, 4, 32, 64, 6), (1, 34, 22, 64, 64, 6)])
def test_conv3x3_winograd(N, H, W, Cin, Cout, tile, K=3):
    """Winograd F(2x2,3x3) path (filter/input transforms, 16 batched MFMA GEMMs, output transform + bias + ReLU)."""
    L = _lib()
    rng = np.random.default_rng(11)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((K, K, Cin, Cout)) / np.sqrt(K * K * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    ref = orc.conv2d_same_t(torch.tensor(x).permute(0, 3, 1, 2).double(), torch.tensor(w).double(), torch.tensor(b).double(), relu=True)
    ref = ref.permute(0, 2, 3, 1).numpy()
    xd, wd, bd = dev(x), dev(w), dev(b)
    y_ = torch.empty(N, H, W, Cout).cuda()
    L.check(L.lib.fcn8s_op_conv2d_winograd(None, ptr(xd), ptr(wd), ptr(bd), ptr(y_), N, H, W, Cin, Cout, K, 1, tile))
    torch.cuda.synchronize()
    tol = 1e-5 if tile == 2 else 1e-4          # fp32 F(4x4,3x3) / F(6x6,3x3) carry ~1e-5 / 2e-5 of the output range (winograd.hip header)
    assert rel_err(y_.cpu().numpy(), ref) < tol
    yd = torch.empty(N, H, W, Cout).cuda()
    L.check(L.lib.fcn8s_op_conv2d(None, ptr(xd), ptr(wd), ptr(bd), ptr(yd), N, H, W, Cin, Cout, K, 1))
    torch.cuda.synchronize()
    assert rel_err(y_.cpu().numpy(), yd.cpu().numpy()) < tol       # and against the direct kernel


WINO_BWD_CASES = [
    # N, H, W, Cin, Cout, tile, pooled, mask_mode
    (1, 34, 22, 64, 64, 6, 0, 0),        # ragged: partial edge tiles in both directions
    (1, 34, 22, 64, 64, 6, 1, 2),        # ... with the pool routing and the one-bit ReLU record of the input (conv1_1 -> conv1_2)
    (1, 96, 160, 64, 128, 6, 1, 1),
    (2, 24, 36, 128, 128, 6, 0, 1),
    (1, 46, 70, 128, 256, 6, 1, 0),
    (1, 30, 54, 256, 256, 6, 1, 1),
    (1, 18, 42, 256, 512, 6, 0, 2),
    (1, 12, 24, 512, 512, 6, 1, 1),
    (2, 12, 18, 192, 192, 6, 0, 1),      # widths the transposed-B GEMM does not take (N = 192): second, transposed filter bank
    (1, 12, 18, 64, 192, 6, 1, 0),
    (1, 16, 32, 64, 64, 4, 1, 1),        # F(4x4): the fused kernel that writes both backward operands from one read of dy
    (2, 8, 16, 128, 256, 4, 0, 0),
    (1, 8, 8, 512, 512, 4, 1, 1),
    (1, 8, 12, 64, 128, 2, 0, 1),        # F(2x2) fallback
]


@pytest.mark.parametrize("N,H,W,Cin,Cout,tile,pooled,mask_mode", WINO_BWD_CASES)
def test_conv3x3_winograd_backward(N, H, W, Cin, Cout, tile, pooled, mask_mode):
    """The 3x3 layers' real training path at op level: forward that keeps V / U / pool argmax bytes / ReLU bits, then the
    Winograd-domain weight gradient, the bias gradient from dM's (1,1) slab and the data gradient -- for tile 6 the adjoint of the forward
    algorithm on the transposed-B GEMM + overlap-add gather -- against float64 autograd.  Tolerance 1e-4 of each tensor's largest value
    (the fp32 F(6x6,3x3) transforms carry ~2e-5 of the range per pass, tools/winograd_matrices.py)."""
    L = _lib()
    rng = np.random.default_rng(21)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((3, 3, Cin, Cout)) / np.sqrt(9 * Cin)).astype(np.float32)
    b = (0.3 * rng.standard_normal(Cout)).astype(np.float32)
    skip = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    xt = torch.tensor(x).permute(0, 3, 1, 2).double().requires_grad_(True)
    wt = torch.tensor(w).double().requires_grad_(True)
    bt = torch.tensor(b).double().requires_grad_(True)
    z = orc.conv2d_same_t(xt, wt, bt, relu=False)
    yt = torch.relu(z)
    if pooled:
        out = orc.maxpool2x2_t(yt)
        dy = rng.standard_normal((N, H // 2, W // 2, Cout)).astype(np.float32)
        # max-pool routing is discontinuous: a window whose two largest entries agree to the fp32 Winograd round-off (1e-5 of the range) may
        # legitimately send its gradient to the other pixel (at 8x8x512 one such window moved a whole 3x3x512 patch of dx by 1e-2).  Those
        # windows get no upstream gradient here, so that the tolerance below measures arithmetic, not coin flips.
        ynp = yt.detach().permute(0, 2, 3, 1).numpy()
        srt = np.sort(ynp.reshape(N, H // 2, 2, W // 2, 2, Cout).transpose(0, 1, 3, 5, 2, 4).reshape(N, H // 2, W // 2, Cout, 4), -1)
        near_tie = (srt[..., 3] > 0) & (srt[..., 3] - srt[..., 2] < 1e-4 * np.abs(ynp).max())
        assert near_tie.mean() < 0.01
        dy[near_tie] = 0.0
        out.backward(torch.tensor(dy).permute(0, 3, 1, 2).double())
    else:
        dy = rng.standard_normal((N, H, W, Cout)).astype(np.float32)      # gradient w.r.t. the pre-activation (the ReLU mask is the consumer's job)
        z.backward(torch.tensor(dy).permute(0, 3, 1, 2).double())
        out = None
    dx_ref = xt.grad.permute(0, 2, 3, 1).numpy() + skip
    if mask_mode:
        dx_ref = dx_ref * (x > 0)
    dw_ref, db_ref = wt.grad.numpy(), bt.grad.numpy()
    xd, wd, bd, dyd, sk = dev(x), dev(w), dev(b), dev(dy), dev(skip)
    y_ = None if pooled else torch.empty(N, H, W, Cout).cuda()
    pool_ = torch.empty(N, H // 2, W // 2, Cout).cuda() if pooled else None
    dx_, dw_, db_ = torch.empty(N, H, W, Cin).cuda(), torch.empty(3, 3, Cin, Cout).cuda(), torch.empty(Cout).cuda()
    L.check(L.lib.fcn8s_op_conv3x3_winograd_fwd_bwd(None, ptr(xd), ptr(wd), ptr(bd), ptr(dyd), ptr(sk), ptr(y_), ptr(pool_), ptr(dx_), ptr(dw_), ptr(db_),
                                                    N, H, W, Cin, Cout, tile, pooled, mask_mode))
    torch.cuda.synchronize()
    tol = 2e-5 if tile == 2 else 1e-4
    if pooled:
        assert rel_err(pool_.cpu().numpy(), out.detach().permute(0, 2, 3, 1).numpy()) < tol
    else:
        assert rel_err(y_.cpu().numpy(), yt.detach().permute(0, 2, 3, 1).numpy()) < tol
    assert rel_err(dx_.cpu().numpy(), dx_ref) < tol, rel_err(dx_.cpu().numpy(), dx_ref)
    assert rel_err(dw_.cpu().numpy(), dw_ref) < tol, rel_err(dw_.cpu().numpy(), dw_ref)
    assert rel_err(db_.cpu().numpy(), db_ref) < tol, rel_err(db_.cpu().numpy(), db_ref)
    # exact zeros where the mask says so (integer work: bit-exact)
    if mask_mode:
        assert (dx_.cpu().numpy()[x <= 0] == 0).all()


@pytest.mark.parametrize("N,H,W,Cin,Cout", [(1, 4, 4, 16, 32), (2, 8, 12, 64, 128), (1, 16, 32, 32, 64)])
def test_conv7x7_winograd_subfilter_decomposition(N, H, W, Cin, Cout):
    """fc6's path: the 7x7 filter as a 3x3 grid of 3x3 sub-filters summed in the F(4x4,3x3) Winograd domain."""
    test_conv3x3_winograd(N, H, W, Cin, Cout, 4, K=7)


@pytest.mark.parametrize("N,H,W,Cin,Cout,K", [(1, 4, 8, 32, 128, 7), (2, 8, 16, 64, 256, 7), (3, 5, 7, 128, 128, 1), (1, 16, 32, 512, 384, 1),
                                               (2, 3, 3, 32, 128, 3)])
def test_conv_bf16_mfma(N, H, W, Cin, Cout, K):
    """FCN8S_PREC_BF16_FC's kernel (config 5): operands rounded to bfloat16 (nearest-even), fp32 products and sums.
    Against a float64 conv of the ROUNDED operands the only difference is fp32 summation order (1e-5); against the
    unrounded conv it is the bf16 operand rounding, 2^-9 per operand (stated, not a parity bar)."""
    L = _lib()
    rng = np.random.default_rng(5)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((K, K, Cin, Cout)) / np.sqrt(K * K * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    xr = torch.tensor(x).to(torch.bfloat16).double().permute(0, 3, 1, 2)
    wr = torch.tensor(w).to(torch.bfloat16).double()
    ref = orc.conv2d_same_t(xr, wr, torch.tensor(b).double(), relu=True).permute(0, 2, 3, 1).numpy()
    exact = orc.conv2d_same_t(torch.tensor(x).double().permute(0, 3, 1, 2), torch.tensor(w).double(), torch.tensor(b).double(), relu=True)
    exact = exact.permute(0, 2, 3, 1).numpy()
    xd, wd, bd = dev(x), dev(w), dev(b)
    y_ = torch.empty(N, H, W, Cout).cuda()
    L.check(L.lib.fcn8s_op_conv2d_bf16(None, ptr(xd), ptr(wd), ptr(bd), ptr(y_), N, H, W, Cin, Cout, K, 1))
    torch.cuda.synchronize()
    y = y_.cpu().numpy()
    assert rel_err(y, ref) < 1e-5
    assert rel_err(y, exact) < 2e-2 and rel_err(y, exact) > 1e-5      # it really ran in bf16
    with pytest.raises(ValueError):
        L.check(L.lib.fcn8s_op_conv2d_bf16(None, ptr(xd), ptr(wd), ptr(bd), ptr(y_), N, H, W, Cin, Cout + 4, K, 1))


@pytest.mark.parametrize("N,H,W,Cin,Cout,K", [
    (2, 16, 32, 64, 64, 3),        # the 64-column tile of the convolution kernel, the 64 x 64 tile of the weight gradient (conv1_2's shape class)
    (1, 20, 24, 64, 128, 3),       # 480 rows: a partial last row tile; 128-column tile forward, 64-column tile for the data gradient (conv2_1)
    (1, 16, 16, 128, 64, 3),
    (2, 8, 8, 256, 256, 3),        # 256-column tiles, 128 x 128 weight-gradient tiles
    (1, 8, 16, 512, 256, 3),
    (1, 64, 64, 64, 64, 3),        # 4356 padded rows: the weight gradient's row range is split over several blocks
    (3, 4, 8, 64, 128, 7),         # fc6's kernel size: guard rows of 3 * Wp + 3 + 32
    (2, 4, 4, 128, 256, 1),        # fc7: a plain GEMM, no padding
    (1, 8, 16, 256, 512, 7),       # fc6's shape class: the data gradient (one 256 x 256 tile, 784 K-tiles) splits K into eight slabs added in order
])
def test_conv_bf16_train_kernels(N, H, W, Cin, Cout, K):
    """The three products of FCN8S_PREC_BF16_TRAIN on the kernels that mode runs (fcn8s_op_conv2d_bf16_train -> conv_bf16_256_kernel<64|128|256> for
    the forward pass and the data gradient, wgrad_bf16_kernel<64|128> for the weight gradient): against float64 evaluations of the SAME bf16-rounded
    operands the only difference is fp32 summation order (1e-5 of the result's largest entry); the bias gradient is the exact fp32 column sum.
    Against the unrounded float64 results the rounding must be visible (the kernels really ran in bf16)."""
    L = _lib()
    rng = np.random.default_rng(11)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((K, K, Cin, Cout)) / np.sqrt(K * K * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    dy = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    mask = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    rb = lambda a: torch.tensor(a).to(torch.bfloat16).double()
    nchw = lambda t: t.permute(0, 3, 1, 2)
    nhwc = lambda t: t.permute(0, 2, 3, 1).numpy()
    pad = (K - 1) // 2
    xr, wr, dyr = nchw(rb(x)), rb(w).permute(3, 2, 0, 1), nchw(rb(dy))
    y_ref = nhwc(torch.relu(torch.nn.functional.conv2d(xr, wr, torch.tensor(b).double(), padding=pad)))
    dx_ref = nhwc(torch.nn.grad.conv2d_input(xr.shape, wr, dyr, padding=pad)) * (mask > 0)
    dw_ref = torch.nn.grad.conv2d_weight(xr, wr.shape, dyr, padding=pad).permute(2, 3, 1, 0).numpy()
    db_ref = dy.astype(np.float64).sum((0, 1, 2))
    dw_exact = torch.nn.grad.conv2d_weight(nchw(torch.tensor(x).double()), wr.shape, nchw(torch.tensor(dy).double()), padding=pad).permute(2, 3, 1, 0).numpy()
    xd, wd, bd, dyd, md = dev(x), dev(w), dev(b), dev(dy), dev(mask)
    y_, dx_ = torch.empty(N, H, W, Cout).cuda(), torch.empty(N, H, W, Cin).cuda()
    dw_, db_ = torch.full((K, K, Cin, Cout), 7.0).cuda(), torch.full((Cout,), 7.0).cuda()      # (garbage: the gradients are assigned, not accumulated)
    L.check(L.lib.fcn8s_op_conv2d_bf16_train(None, ptr(xd), ptr(wd), ptr(bd), ptr(y_), 1, ptr(dyd), ptr(md), ptr(dx_), ptr(dw_), ptr(db_), N, H, W, Cin, Cout, K))
    torch.cuda.synchronize()
    assert rel_err(y_.cpu().numpy(), y_ref) < 1e-5
    assert rel_err(dx_.cpu().numpy(), dx_ref) < 1e-5
    assert rel_err(dw_.cpu().numpy(), dw_ref) < 1e-5
    assert rel_err(db_.cpu().numpy(), db_ref) < 1e-5
    assert 1e-5 < rel_err(dw_.cpu().numpy(), dw_exact) < 3e-2
    # the data gradient without a mask (fc6's case: an identity epilogue -- few output tiles behind a long reduction split K into slabs added in split order)
    dx2 = torch.full((N, H, W, Cin), 7.0).cuda()
    L.check(L.lib.fcn8s_op_conv2d_bf16_train(None, None, ptr(wd), None, None, 0, ptr(dyd), None, ptr(dx2), None, None, N, H, W, Cin, Cout, K))
    torch.cuda.synchronize()
    assert rel_err(dx2.cpu().numpy(), nhwc(torch.nn.grad.conv2d_input(xr.shape, wr, dyr, padding=pad))) < 1e-5
    # deterministic mode: the split weight gradient through slabs, bit-identical from run to run and equal to the default to round-off
    L.check(L.lib.fcn8s_set_option(None, b"op_deterministic", 1))
    try:
        outs = []
        for _ in range(2):
            t = torch.empty(K, K, Cin, Cout).cuda()
            L.check(L.lib.fcn8s_op_conv2d_bf16_train(None, ptr(xd), None, None, None, 0, ptr(dyd), None, None, ptr(t), None, N, H, W, Cin, Cout, K))
            torch.cuda.synchronize()
            outs.append(t.cpu().numpy())
    finally:
        L.check(L.lib.fcn8s_set_option(None, b"op_deterministic", 0))
    np.testing.assert_array_equal(outs[0], outs[1])
    assert rel_err(outs[0], dw_ref) < 1e-5
    # the padded copies as [rows][C] instead of channel-chunk planes (the layout the other bf16 modes hand these kernels): the same results
    L.check(L.lib.fcn8s_set_option(None, b"op_bf16_planes", 0))
    try:
        y2, dx3, dw2 = torch.empty(N, H, W, Cout).cuda(), torch.empty(N, H, W, Cin).cuda(), torch.empty(K, K, Cin, Cout).cuda()
        L.check(L.lib.fcn8s_op_conv2d_bf16_train(None, ptr(xd), ptr(wd), ptr(bd), ptr(y2), 1, ptr(dyd), ptr(md), ptr(dx3), ptr(dw2), None, N, H, W, Cin, Cout, K))
        torch.cuda.synchronize()
    finally:
        L.check(L.lib.fcn8s_set_option(None, b"op_bf16_planes", 1))
    np.testing.assert_array_equal(y2.cpu().numpy(), y_.cpu().numpy())
    np.testing.assert_array_equal(dx3.cpu().numpy(), dx_.cpu().numpy())
    assert rel_err(dw2.cpu().numpy(), dw_ref) < 1e-5
    with pytest.raises(ValueError):
        L.check(L.lib.fcn8s_op_conv2d_bf16_train(None, ptr(xd), ptr(wd), ptr(bd), ptr(y_), 1, None, None, None, None, None, N, H, W, Cin + 32, Cout, K))


@pytest.mark.parametrize("N,H,W,Cin,Cout", [
    (1, 16, 16, 64, 256),          # two channel chunks (18 steps: the ring of five B slots wraps), a partial last row tile (324 padded positions), two column tiles
    (2, 24, 40, 256, 256),         # several row tiles, eight channel chunks, tiles that straddle the two images
    (1, 16, 32, 512, 512),         # four column tiles
    (1, 30, 34, 128, 128),         # one column tile; five row tiles, the last one partial
    (2, 8, 8, 256, 384),           # 384 = 3 x 128
    (1, 8, 8, 64, 128),            # 100 padded positions: a single, mostly empty row tile
])
def test_conv_bf16_train_rows_forms(N, H, W, Cin, Cout):
    """The two forms of the flat-position 3 x 3 kernel (conv_bf16_rows_kernel<64>: blocks of 256 positions x 64 channels, the default; <128>: 128 x 128, op option
    `op_bf16_rows_bn`) on shapes with partial last row tiles, tiles that straddle images and several column tiles: forward (bias, ReLU) and masked data
    gradient against float64 evaluations of the same bf16-rounded operands (1e-5: fp32 summation order), and BIT-identical to each other -- both add the products
    of a dot product in the same order (K-tile by K-tile, tap by tap, 16 channels at a time).  [Round 6 ran a third form with fat waves, 128 positions x 64
    channels per wave, through this test: bit-identical, 12-50 % slower, not shipped -- tools/labs/conv_bf16_taps_lab.hip, profiles/r06_bf16_fat_tile_lab.txt.]"""
    L = _lib()
    K = 3
    rng = np.random.default_rng(12)
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((K, K, Cin, Cout)) / np.sqrt(K * K * Cin)).astype(np.float32)
    b = rng.standard_normal(Cout).astype(np.float32)
    dy = rng.standard_normal((N, H, W, Cout)).astype(np.float32)
    mask = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    rb = lambda a: torch.tensor(a).to(torch.bfloat16).double()
    nchw = lambda t: t.permute(0, 3, 1, 2)
    nhwc = lambda t: t.permute(0, 2, 3, 1).numpy()
    xr, wr, dyr = nchw(rb(x)), rb(w).permute(3, 2, 0, 1), nchw(rb(dy))
    y_ref = nhwc(torch.relu(torch.nn.functional.conv2d(xr, wr, torch.tensor(b).double(), padding=1)))
    dx_ref = nhwc(torch.nn.grad.conv2d_input(xr.shape, wr, dyr, padding=1)) * (mask > 0)
    xd, wd, bd, dyd, md = dev(x), dev(w), dev(b), dev(dy), dev(mask)
    out = {}
    for form in (128, 64):
        L.check(L.lib.fcn8s_set_option(None, b"op_bf16_rows_bn", form))
        try:
            y_, dx_ = torch.full((N, H, W, Cout), 7.0).cuda(), torch.full((N, H, W, Cin), 7.0).cuda()
            L.check(L.lib.fcn8s_op_conv2d_bf16_train(None, ptr(xd), ptr(wd), ptr(bd), ptr(y_), 1, ptr(dyd), ptr(md), ptr(dx_), None, None, N, H, W, Cin, Cout, K))
            torch.cuda.synchronize()
        finally:
            L.check(L.lib.fcn8s_set_option(None, b"op_bf16_rows_bn", 0))
        out[form] = (y_.cpu().numpy(), dx_.cpu().numpy())
        assert rel_err(out[form][0], y_ref) < 1e-5, form
        assert rel_err(out[form][1], dx_ref) < 1e-5, form
    # (a forward launch with few blocks behind >= 24 K-tiles -- Cin >= 256 at these sizes -- splits K into slabs added in slab order, and the two forms cut the
    #  K range differently: the same terms in another order)
    if Cin >= 256:
        assert rel_err(out[128][0], out[64][0]) < 2e-6
    else:
        np.testing.assert_array_equal(out[128][0], out[64][0])
    np.testing.assert_array_equal(out[128][1], out[64][1])


@pytest.mark.parametrize("N,H,W,C", [(2, 8, 8, 64), (1, 4, 6, 8), (1, 32, 64, 128)])
def test_maxpool(N, H, W, C):
    L = _lib()
    rng = np.random.default_rng(2)
    x = np.maximum(rng.standard_normal((N, H, W, C)), 0).astype(np.float32)   # post-ReLU, many exact zeros (ties)
    dy = rng.standard_normal((N, H // 2, W // 2, C)).astype(np.float32)
    xt = torch.tensor(x).permute(0, 3, 1, 2).requires_grad_(True)
    yt = orc.maxpool2x2_t(xt)
    xd, dyd = dev(x), dev(dy)
    y_ = torch.empty(N, H // 2, W // 2, C).cuda(); dx_ = torch.empty(N, H, W, C).cuda()
    L.check(L.lib.fcn8s_op_maxpool2x2(None, ptr(xd), ptr(y_), N, H, W, C))
    np.testing.assert_array_equal(y_.cpu().numpy(), yt.detach().permute(0, 2, 3, 1).numpy())
    # backward with the fused ReLU mask == autograd through relu(pre)->pool when x = relu(pre)
    pre = torch.tensor(x).permute(0, 3, 1, 2).clone()
    pre[pre == 0] = -1.0
    pre.requires_grad_(True)
    orc.maxpool2x2_t(torch.relu(pre)).backward(torch.tensor(dy).permute(0, 3, 1, 2))
    L.check(L.lib.fcn8s_op_maxpool2x2_bwd(None, ptr(xd), ptr(dyd), ptr(dx_), N, H, W, C, 1))
    np.testing.assert_array_equal(dx_.cpu().numpy(), pre.grad.permute(0, 2, 3, 1).numpy())


@pytest.mark.parametrize("N,Hi,Wi,C,K,S", [(2, 3, 5, 20, 4, 2), (1, 4, 4, 20, 16, 8), (1, 2, 3, 8, 4, 2), (2, 8, 16, 20, 16, 8)])
def test_conv2d_transpose_fwd_bwd(N, Hi, Wi, C, K, S):
    L = _lib()
    rng = np.random.default_rng(3)
    x = rng.standard_normal((N, Hi, Wi, C)).astype(np.float32)
    w = (rng.standard_normal((K, K, C, C)) * 0.1).astype(np.float32)
    b = rng.standard_normal(C).astype(np.float32)
    add = rng.standard_normal((N, Hi * S, Wi * S, C)).astype(np.float32)
    dy = rng.standard_normal((N, Hi * S, Wi * S, C)).astype(np.float32)
    xt = torch.tensor(x).permute(0, 3, 1, 2).double().requires_grad_(True)
    wt = torch.tensor(w).double().requires_grad_(True)
    bt = torch.tensor(b).double().requires_grad_(True)
    yt = orc.conv2d_transpose_same_t(xt, wt, bt, S) + torch.tensor(add).permute(0, 3, 1, 2).double()
    yt.backward(torch.tensor(dy).permute(0, 3, 1, 2).double())
    xd, wd, bd, ad, dyd = dev(x), dev(w), dev(b), dev(add), dev(dy)
    y_ = torch.empty(N, Hi * S, Wi * S, C).cuda()
    dx_, dw_, db_ = torch.empty(N, Hi, Wi, C).cuda(), torch.empty(K, K, C, C).cuda(), torch.empty(C).cuda()
    L.check(L.lib.fcn8s_op_conv2d_transpose(None, ptr(xd), ptr(wd), ptr(bd), ptr(ad), ptr(y_), N, Hi, Wi, C, K, S))
    L.check(L.lib.fcn8s_op_conv2d_transpose_bwd(None, ptr(xd), ptr(wd), ptr(dyd), ptr(dx_), ptr(dw_), ptr(db_), N, Hi, Wi, C, K, S))
    torch.cuda.synchronize()
    assert rel_err(y_.cpu().numpy(), yt.detach().permute(0, 2, 3, 1).numpy()) < 2e-5
    assert rel_err(dx_.cpu().numpy(), xt.grad.permute(0, 2, 3, 1).numpy()) < 2e-5
    assert rel_err(dw_.cpu().numpy(), wt.grad.numpy()) < 2e-5
    assert rel_err(db_.cpu().numpy(), bt.grad.numpy()) < 2e-5


@pytest.mark.parametrize("npix,Cc", [(1000, 20), (4096, 20), (333, 4), (257, 12)])
def test_softmax_xent_and_argmax(npix, Cc):
    L = _lib()
    rng = np.random.default_rng(4)
    logits = (rng.standard_normal((npix, Cc)) * 3).astype(np.float32)
    labels = rng.integers(0, Cc, npix).astype(np.uint8)
    lt = torch.tensor(logits).double().requires_grad_(True)
    loss = torch.nn.functional.cross_entropy(lt, torch.tensor(labels.astype(np.int64)))
    loss.backward()
    ld, lab = dev(logits), dev(labels)
    dl = torch.empty(npix, Cc).cuda(); lo = torch.zeros(1).cuda()
    L.check(L.lib.fcn8s_op_softmax_xent(None, ptr(ld), ptr(lab), ptr(dl), ptr(lo), npix, Cc))
    assert abs(float(lo.cpu()) - float(loss.detach())) < 1e-5 * max(1.0, abs(float(loss.detach())))
    assert np.abs(dl.cpu().numpy() - lt.grad.numpy()).max() < 1e-6
    sm = torch.empty(npix, Cc).cuda(); am = torch.empty(npix, dtype=torch.int64).cuda()
    L.check(L.lib.fcn8s_op_softmax_argmax(None, ptr(ld), ptr(sm), ptr(am), npix, Cc))
    sm_ref = orc.softmax(logits)
    assert np.abs(sm.cpu().numpy() - sm_ref).max() < 1e-6
    np.testing.assert_array_equal(am.cpu().numpy(), np.argmax(sm_ref, -1))


def test_argmax_ties_lowest_index():
    L = _lib()
    logits = np.zeros((64, 20), np.float32)
    logits[:, 7] = 1.0; logits[:, 11] = 1.0       # exact tie -> lowest index (7)
    am = torch.empty(64, dtype=torch.int64).cuda()
    ld = dev(logits)
    L.check(L.lib.fcn8s_op_softmax_argmax(None, ptr(ld), None, ptr(am), 64, 20))
    torch.cuda.synchronize()
    assert (am.cpu().numpy() == 7).all()


def test_confusion_matrix():
    L = _lib()
    rng = np.random.default_rng(5)
    n, Cc = 100000, 20
    lab = rng.integers(0, Cc, n).astype(np.uint8); pred = rng.integers(0, Cc, n).astype(np.int64)
    conf = torch.zeros(Cc * Cc, dtype=torch.int64).cuda()
    labd, predd = dev(lab), dev(pred)
    for _ in range(2):
        L.check(L.lib.fcn8s_op_confusion(None, ptr(labd), ptr(predd), n, ptr(conf), Cc))
    torch.cuda.synchronize()
    np.testing.assert_array_equal(conf.cpu().numpy().reshape(Cc, Cc), 2 * orc.confusion_matrix(lab, pred, Cc))


def test_optimizers():
    L = _lib()
    rng = np.random.default_rng(6)
    n = 100003
    th = rng.standard_normal(n).astype(np.float32); m = np.zeros(n, np.float32); v = np.zeros(n, np.float32)
    thd, md, vd = dev(th), dev(m), dev(v)
    for t in range(1, 4):
        g = rng.standard_normal(n).astype(np.float32)
        th, m, v = orc.tf_adam_step(th, g, m, v, t, 1e-3)
        gd = dev(g)
        L.check(L.lib.fcn8s_op_tf_adam(None, ptr(thd), ptr(gd), ptr(md), ptr(vd), n, t, 1e-3, 0.9, 0.999, 1e-8, 1.0))
        torch.cuda.synchronize()
    assert np.abs(thd.cpu().numpy() - th).max() < 2e-6
    assert np.abs(md.cpu().numpy() - m).max() < 1e-6 and np.abs(vd.cpu().numpy() - v).max() < 1e-6
    buf = np.zeros(n, np.float32); bd = dev(buf); th2 = th.copy(); th2d = dev(th2)
    for _ in range(3):
        g = rng.standard_normal(n).astype(np.float32)
        th2, buf = orc.sgd_momentum_step(th2, g, buf, 1e-2)
        gd = dev(g)
        L.check(L.lib.fcn8s_op_sgd_momentum(None, ptr(th2d), ptr(gd), ptr(bd), n, 1e-2, 0.9, 1.0))
        torch.cuda.synchronize()
    assert np.abs(th2d.cpu().numpy() - th2).max() < 2e-6


def test_preprocess():
    L = _lib()
    rng = np.random.default_rng(7)
    img = rng.integers(0, 256, (2, 4, 8, 3), dtype=np.uint8)
    out = torch.empty(2, 4, 8, 4).cuda()
    imgd = dev(img)
    L.check(L.lib.fcn8s_op_preprocess(None, ptr(imgd), 0, ptr(out), 2 * 4 * 8))
    torch.cuda.synchronize()
    ref = orc.preprocess_t(torch.tensor(img).float()).numpy()
    o = out.cpu().numpy()
    np.testing.assert_allclose(o[..., :3], ref, rtol=0, atol=1e-5)
    assert (o[..., 3] == 0).all()


def test_split_piece_count_sets_the_product_accuracy():
    """The three arithmetics of the LDS-DMA GEMMs on one 1x1 conv (a plain [2048 x 512] x [512 x 256] GEMM) against a float64 product of the
    same operands: the f32 MFMA and the three-piece split (f32x3) agree with it to fp32 round-off; the two-piece split (f32x2, 16 significand
    bits of each operand) sits one to two orders above, well under TF32's 2^-11, and equals -- to fp32 summation order -- the float64
    product of the operands cut to their two pieces (that is what the mode is defined as)."""
    L = _lib()
    rng = np.random.default_rng(12)
    N, H, W, Cin, Cout = 2, 32, 32, 512, 256
    x = rng.standard_normal((N, H, W, Cin)).astype(np.float32)
    w = (rng.standard_normal((1, 1, Cin, Cout)) / np.sqrt(Cin)).astype(np.float32)
    xd, wd = dev(x), dev(w)
    want = x.reshape(-1, Cin).astype(np.float64) @ w.reshape(Cin, Cout).astype(np.float64)

    def two_pieces(a):
        t = torch.from_numpy(a)
        hi = t.to(torch.bfloat16).to(torch.float32)
        lo = (t - hi).to(torch.bfloat16).to(torch.float32)
        return (hi.double() + lo.double()).numpy()
    x2, w2 = two_pieces(x).reshape(-1, Cin), two_pieces(w).reshape(Cin, Cout)
    hi = lambda a: torch.from_numpy(a).to(torch.bfloat16).double().numpy()
    # a b ~ hi hi + hi lo + lo hi (the lo lo term is dropped)
    want2 = x2 @ w2 - (x2 - hi(x).reshape(-1, Cin)) @ (w2 - hi(w).reshape(Cin, Cout))
    prev = C.c_int64()
    L.check(L.lib.fcn8s_get_option(None, b"op_split_pieces", C.byref(prev)))
    err = {}
    try:
        for pieces in (0, 3, 2):
            L.check(L.lib.fcn8s_set_option(None, b"op_split_pieces", pieces))
            y = torch.empty(N, H, W, Cout).cuda()
            L.check(L.lib.fcn8s_op_conv2d(None, ptr(xd), ptr(wd), None, ptr(y), N, H, W, Cin, Cout, 1, 0))
            torch.cuda.synchronize()
            got = y.cpu().numpy().reshape(-1, Cout)
            err[pieces] = rel_err(got, want)
            if pieces == 2:
                err["2 vs its definition"] = rel_err(got, want2)
        assert L.lib.fcn8s_set_option(None, b"op_split_pieces", 1) != 0          # 0, 2 or 3 only
    finally:
        L.check(L.lib.fcn8s_set_option(None, b"op_split_pieces", int(prev.value)))
    print("max error / max |y| against float64: f32 MFMA %.1e, three pieces %.1e, two pieces %.1e (vs the two-piece product in float64 %.1e)"
          % (err[0], err[3], err[2], err["2 vs its definition"]))
    assert err[0] < 2e-6 and err[3] < 2e-6, err
    assert 1e-6 < err[2] < 3e-5, err                   # 2^-17 per operand, random signs over K = 512
    assert err["2 vs its definition"] < 2e-6, err


def test_conv_ops_in_f32x3_mode_hold_the_fp32_tolerances():
    """fcn8s_set_option(NULL, "op_f32x3", 1) routes every LDS-DMA GEMM of the op-level entry points through the split-bf16 kernels (the
    model-level switch is fcn8s_set_precision): the convolution cases above must hold their fp32 tolerances (2e-5) unchanged."""
    import os, subprocess, sys
    env = dict(os.environ, FCN8S_TEST_OP_F32X3="1")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-k", "conv and not f32x3"],
                       env=env, capture_output=True, text=True, cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
