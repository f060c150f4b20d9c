"""T2: every HIP kernel against a plain-PyTorch fp32 reference of the same op.

Random (never zero / symmetric) data, asymmetric operands, odd and ragged shapes.
bf16 kernels are compared with an fp32 reference computed from the SAME
bf16-rounded inputs, so the tolerance only has to cover fp32-accumulate + one
output rounding.
"""
import math

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _C():
    from distributedpytorch_amd.ops import _ext
    return _ext.load()


def _rand_bf16(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


def _close(got, ref, rtol, atol, what="", max_bad_frac=0.0):
    """``max_bad_frac``: fraction of elements allowed outside tolerance - used only where a
    ReLU mask is taken from a bf16-rounded activation, so an element within one rounding of
    zero may legitimately fall on the other side of the mask than in the fp32 reference."""
    got, ref = got.float().cpu(), ref.float().cpu()
    err = (got - ref).abs()
    tol = atol + rtol * ref.abs()
    bad = (err > tol).sum().item()
    assert bad <= max_bad_frac * err.numel(), f"{what}: {bad}/{err.numel()} out of tolerance, max err {err.max().item():.4g}, ref max {ref.abs().max().item():.4g}"


CONV_CASES = [
    # N, H, W, C, K, R, stride, pad
    (2, 14, 14, 64, 128, 1, 1, 0),
    (2, 14, 14, 64, 64, 3, 1, 1),     # K<=64 path (256x64 tiles)
    (3, 15, 13, 32, 136, 3, 1, 1),    # odd spatial, K not a tile multiple
    (2, 16, 16, 128, 256, 3, 2, 1),   # stride 2
    (2, 16, 16, 256, 512, 1, 2, 0),   # 1x1 stride 2 (downsample)
    (2, 32, 32, 8, 64, 7, 2, 3),      # stem shape: C=8 (3 padded), 7x7/2
    (1, 7, 7, 512, 512, 3, 1, 1),     # small M, deep K
    (5, 1, 1, 320, 56, 1, 1, 0),      # Linear-shaped, ragged rows
]


def _ref_conv(x, w, stride, pad):
    return F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=stride, padding=pad)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_fprop(case):
    N, H, W, C, K, R, s, p = case
    x = _rand_bf16(N, H, W, C, seed=1)
    w = _rand_bf16(K, R, R, C, scale=1.0 / math.sqrt(R * R * C), seed=2)
    (y,) = _C().conv_fprop(x, w, None, s, p, 0, False, None)
    ref = _ref_conv(x, w, s, p).permute(0, 2, 3, 1)
    assert y.shape == ref.shape
    _close(y, ref, 1e-2, 2e-2, "fprop")


def test_conv_fprop_bias_relu_and_stats():
    N, H, W, C, K = 4, 9, 9, 64, 192
    x = _rand_bf16(N, H, W, C, seed=3)
    w = _rand_bf16(K, 3, 3, C, scale=0.05, seed=4)
    b = torch.randn(K, device=DEV)
    (y,) = _C().conv_fprop(x, w, b, 1, 1, 1, False, None)
    ref = F.relu(_ref_conv(x, w, 1, 1) + b.view(1, -1, 1, 1)).permute(0, 2, 3, 1)
    _close(y, ref, 1e-2, 2e-2, "bias+relu")
    y2, stats = _C().conv_fprop(x, w, None, 1, 1, 0, True, None)
    yf = y2.float().reshape(-1, K)
    s = stats.sum(0)
    _close(s[0], yf.sum(0), 1e-3, 5e-2, "stats sum")
    _close(s[1], (yf * yf).sum(0), 1e-3, 5e-2, "stats sumsq")
    # persistent accumulator: conv adds into it, bn_finalize(reset) re-zeroes it
    buf = torch.zeros(32, 2, K, device=DEV)
    _C().conv_fprop(x, w, None, 1, 1, 0, True, buf)
    _close(buf.sum(0)[0], yf.sum(0), 1e-3, 5e-2, "stats_buf sum")
    g, bt = torch.ones(K, device=DEV), torch.zeros(K, device=DEV)
    mean, invstd, scale, shift = _C().bn_finalize(buf, yf.shape[0], g, bt, None, None, 0.1, 1e-5, True)
    _close(mean, yf.mean(0), 1e-3, 1e-3, "mean from fused stats")
    _close(invstd, 1.0 / torch.sqrt(yf.var(0, unbiased=False) + 1e-5), 1e-3, 1e-3, "invstd from fused stats")
    assert float(buf.abs().max()) == 0.0


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_dgrad(case):
    N, H, W, C, K, R, s, p = case
    Ho, Wo = (H + 2 * p - R) // s + 1, (W + 2 * p - R) // s + 1
    dy = _rand_bf16(N, Ho, Wo, K, seed=5)
    w = _rand_bf16(K, R, R, C, scale=1.0 / math.sqrt(R * R * K), seed=6)
    wt = _C().transpose_filter(w)
    assert torch.equal(wt, w.permute(3, 1, 2, 0).contiguous())
    dx = _C().conv_dgrad(dy, wt, H, W, s, p, None, None)
    xr = torch.zeros(N, C, H, W, requires_grad=True)
    yr = F.conv2d(xr, w.float().cpu().permute(0, 3, 1, 2), stride=s, padding=p)
    yr.backward(dy.float().cpu().permute(0, 3, 1, 2))
    _close(dx, xr.grad.permute(0, 2, 3, 1), 1e-2, 2e-2, "dgrad")
    # fused residual-gradient add in the epilogue
    add = _rand_bf16(N, H, W, C, seed=55)
    dx2 = _C().conv_dgrad(dy, wt, H, W, s, p, add, None)
    _close(dx2, xr.grad.permute(0, 2, 3, 1) + add.float().cpu(), 1e-2, 3e-2, "dgrad+addend")
    # ... under a ReLU bit mask (1 bit per element, 8 per byte, LSB = lowest channel)
    bits = torch.rand(N, H, W, C, generator=torch.Generator().manual_seed(56)) > 0.5
    packed = (bits.view(-1, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(1).to(torch.uint8).to(DEV)
    dx3 = _C().conv_dgrad(dy, wt, H, W, s, p, add, packed)
    _close(dx3, xr.grad.permute(0, 2, 3, 1) + add.float().cpu() * bits, 1e-2, 3e-2, "dgrad+masked addend")
    mm = _C().relu_mask_mul(add, packed)
    assert torch.equal(mm.float().cpu(), add.float().cpu() * bits)


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv_wgrad(case):
    N, H, W, C, K, R, s, p = case
    Ho, Wo = (H + 2 * p - R) // s + 1, (W + 2 * p - R) // s + 1
    x = _rand_bf16(N, H, W, C, seed=7)
    dy = _rand_bf16(N, Ho, Wo, K, scale=0.1, seed=8)
    dw = torch.full((K, R, R, C), 0.5, device=DEV)          # accumulates INTO dw
    _C().conv_wgrad(x, dy, dw, s, p)
    wr = torch.zeros(K, C, R, R, requires_grad=True)
    yr = F.conv2d(x.float().cpu().permute(0, 3, 1, 2), wr, stride=s, padding=p)
    yr.backward(dy.float().cpu().permute(0, 3, 1, 2))
    ref = wr.grad.permute(0, 2, 3, 1) + 0.5
    _close(dw, ref, 5e-3, 2e-2, "wgrad")


def test_conv_wgrad_split_k_large():
    """Enough pixels to split the reduction over many workgroups (atomics path)."""
    N, H, W, C, K = 8, 56, 56, 64, 64
    x = _rand_bf16(N, H, W, C, seed=9)
    dy = _rand_bf16(N, H, W, K, scale=0.05, seed=10)
    dw = torch.zeros(K, 1, 1, C, device=DEV)
    _C().conv_wgrad(x, dy, dw, 1, 0)
    ref = dy.float().reshape(-1, K).t() @ x.float().reshape(-1, C)
    _close(dw.view(K, C), ref, 5e-3, 5e-2, "wgrad split-k")


@pytest.mark.parametrize("case", [(3, 56, 56, 64, 64), (2, 56, 56, 128, 64), (1, 56, 56, 64, 192), (5, 2, 56, 64, 64)])
def test_conv_wgrad_halo_tiled_3x3_on_56_wide_maps(case):
    """csrc/conv_wgrad_halo.hip (conv_wgrad routes 3 x 3 / stride 1 / pad 1 problems on 56-wide maps with 64-channel
    slabs to it): image borders, several slabs, fewer tiles than workgroups, accumulation INTO dw."""
    N, H, W, C, K = case
    x = _rand_bf16(N, H, W, C, seed=31)
    dy = _rand_bf16(N, H, W, K, scale=0.1, seed=32)
    dw = torch.full((K, 3, 3, C), 0.25, device=DEV)
    _C().conv_wgrad(x, dy, dw, 1, 1)
    ref = torch.nn.grad.conv2d_weight(x.float().cpu().permute(0, 3, 1, 2), (K, C, 3, 3), dy.float().cpu().permute(0, 3, 1, 2),
                                      stride=1, padding=1).permute(0, 2, 3, 1) + 0.25
    _close(dw, ref, 5e-3, 2e-2, "halo wgrad")


@pytest.mark.parametrize("shape", [(4096, 768, 2304), (1000, 64, 64), (2048, 3072, 776), (8, 9, 9, 64, 128)])
def test_conv_wgrad_bias_gradient(shape):
    """dbias += column sums of dy, from the wgrad kernel itself (Linear shapes: an MFMA against ones) or from the
    column-sum fallback (conv shapes); dw must be unaffected."""
    if len(shape) == 3:
        M, C, K = shape
        x = _rand_bf16(M, 1, 1, C, seed=21)
        dy = _rand_bf16(M, 1, 1, K, scale=0.1, seed=22)
        R, pad = 1, 0
    else:
        N, H, W, C, K = shape
        x = _rand_bf16(N, H, W, C, seed=21)
        dy = _rand_bf16(N, H, W, K, scale=0.1, seed=22)
        R, pad = 3, 1
    dw = torch.zeros(K, R, R, C, device=DEV)
    dw_ref = torch.zeros(K, R, R, C, device=DEV)
    db = torch.full((K,), 0.25, device=DEV)
    _C().conv_wgrad(x, dy, dw, 1, pad, db)
    _C().conv_wgrad(x, dy, dw_ref, 1, pad)
    _close(db, dy.float().reshape(-1, K).sum(0) + 0.25, 2e-3, 2e-2, "dbias")
    _close(dw, dw_ref, 1e-4, 1e-3, "dw beside dbias")


def test_colsum():
    x = _rand_bf16(1000, 1000, seed=11)
    out = torch.ones(1000, device=DEV)
    _C().colsum(x, out)
    _close(out, x.float().sum(0) + 1, 1e-3, 1e-2, "colsum")
    x = _rand_bf16(300, 4096, seed=12)
    out = torch.zeros(4096, device=DEV)
    _C().colsum(x, out)
    _close(out, x.float().sum(0), 1e-3, 1e-2, "colsum wide")


@pytest.mark.parametrize("shape", [(4, 7, 7, 64), (2, 13, 11, 256), (3, 5, 5, 2048), (64, 28, 28, 128),
                                   (5, 9, 9, 48), (70, 15, 15, 136)])   # C / 8 = 6, 17: the apply kernels' per-chunk coefficient path
@pytest.mark.parametrize("relu,res", [(False, False), (True, False), (True, True)])
def test_batchnorm_fwd_bwd(shape, relu, res):
    from distributedpytorch_amd import ops
    C = shape[-1]
    x = (_rand_bf16(*shape, seed=13) * 2 + 0.5).to(torch.bfloat16)
    r = _rand_bf16(*shape, seed=14) if res else None
    gamma = (torch.rand(C, device=DEV) + 0.5).requires_grad_()
    beta = torch.randn(C, device=DEV).requires_grad_()
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    xg = x.clone().requires_grad_()
    rg = r.clone().requires_grad_() if res else None
    y = ops.batch_norm_nhwc(xg, gamma, beta, rm, rv, True, 0.1, 1e-5, relu=relu, residual=rg)
    dy = _rand_bf16(*shape, seed=15)
    y.backward(dy)
    # fp32 reference
    xr = x.float().cpu().requires_grad_()
    rr = r.float().cpu().requires_grad_() if res else None
    gr, br = gamma.detach().cpu().requires_grad_(), beta.detach().cpu().requires_grad_()
    rm2, rv2 = torch.zeros(C), torch.ones(C)
    yr = F.batch_norm(xr.permute(0, 3, 1, 2), rm2, rv2, gr, br, True, 0.1, 1e-5).permute(0, 2, 3, 1)
    if res:
        yr = yr + rr
    if relu:
        yr = F.relu(yr)
    yr.backward(dy.float().cpu())
    _close(y, yr, 1e-2, 2e-2, "bn y")
    _close(rm, rm2, 1e-3, 1e-3, "running mean")
    _close(rv, rv2, 1e-3, 1e-3, "running var")
    _close(xg.grad, xr.grad, 2e-2, 3e-2, "bn dx", max_bad_frac=1e-4 if relu else 0.0)
    n = x.numel() / C
    # with a ReLU, elements within one rounding of zero may sit on the other side of the mask than in the fp32 reference
    # (which ones depends on the last bits of the batch mean, i.e. on the order of the statistics atomics): each flip
    # moves a channel's dgamma / dbeta by ~|dy * xhat| ~ 1, so at 50k elements per channel one channel in a hundred can
    # land just outside the band
    _close(gamma.grad, gr.grad, 1e-2, 2e-2 * math.sqrt(n), "dgamma", max_bad_frac=0.02 if relu else 0.0)
    _close(beta.grad, br.grad, 1e-2, 2e-2 * math.sqrt(n), "dbeta", max_bad_frac=0.02 if relu else 0.0)
    if res:
        _close(rg.grad, rr.grad, 1e-2, 1e-2, "dres", max_bad_frac=1e-4 if relu else 0.0)


def test_conv_bn_fused_stats_match_unfused():
    from distributedpytorch_amd import ops
    x = _rand_bf16(8, 14, 14, 64, seed=16)
    w = _rand_bf16(128, 3, 3, 64, scale=0.05, seed=17).float()
    gamma, beta = torch.rand(128, device=DEV) + 0.5, torch.randn(128, device=DEV)
    y1, st = ops.conv2d_nhwc(x, w, None, None, 1, 1, want_stats=True)
    a = ops.batch_norm_nhwc(y1, gamma, beta, None, None, True, relu=True, stats=st)
    b = ops.batch_norm_nhwc(y1, gamma, beta, None, None, True, relu=True, stats=None)
    _close(a, b, 1e-2, 1e-2, "fused vs unfused stats")


def test_maxpool_avgpool():
    from distributedpytorch_amd import ops
    x = _rand_bf16(3, 17, 19, 64, seed=18).requires_grad_()
    y = ops.max_pool_3x3_s2_nhwc(x)
    dy = _rand_bf16(*y.shape, seed=19)
    y.backward(dy)
    xr = x.detach().float().cpu().requires_grad_()
    yr = F.max_pool2d(xr.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1)
    yr.backward(dy.float().cpu())
    _close(y, yr, 0, 0, "maxpool")
    _close(x.grad, xr.grad, 1e-2, 1e-2, "maxpool bwd")
    x = _rand_bf16(5, 7, 7, 2048, seed=20).requires_grad_()
    y = ops.global_avg_pool_nhwc(x)
    y.backward(torch.ones_like(y))
    _close(y, x.detach().float().mean((1, 2)), 1e-2, 1e-2, "avgpool")
    _close(x.grad, torch.full_like(x, 1 / 49.0), 1e-2, 1e-3, "avgpool bwd")


def test_nchw_to_nhwc_pad():
    from distributedpytorch_amd import ops
    x = torch.randn(3, 3, 20, 24, device=DEV)
    y = ops.to_nhwc(x, 8)
    assert y.shape == (3, 20, 24, 8) and y.dtype == torch.bfloat16
    _close(y[..., :3], x.permute(0, 2, 3, 1), 1e-2, 1e-2, "layout")
    assert float(y[..., 3:].abs().max()) == 0.0


@pytest.mark.parametrize("B,V,ld,dtype", [(12, 10, 10, torch.float32), (96, 1000, 1000, torch.bfloat16),
                                          (33, 50257, 50304, torch.bfloat16), (7, 13, 16, torch.bfloat16)])
def test_softmax_ce(B, V, ld, dtype):
    from distributedpytorch_amd import ops
    logits = (torch.randn(B, ld, device=DEV) * 3).to(dtype)
    t = torch.randint(0, V, (B,), device=DEV)
    lg = logits.clone().requires_grad_()
    loss = ops.softmax_cross_entropy(lg, t, V)
    (loss * 2.0).backward()
    lr = logits.float().cpu()[:, :V].requires_grad_()
    ref = F.cross_entropy(lr, t.cpu())
    (ref * 2.0).backward()
    _close(loss, ref, 1e-3, 1e-3, "ce loss")
    _close(lg.grad[:, :V], lr.grad, 2e-2, 1e-5, "ce grad")
    if ld > V:
        assert float(lg.grad[:, V:].abs().max()) == 0.0


@pytest.mark.parametrize("B,V,ld", [(33, 50257, 50304), (64, 1000, 1000), (9, 8, 8), (17, 13, 32), (5, 65536, 65536), (20, 8193, 8208)])
@pytest.mark.parametrize("upstream", [1.0, 2.5])
def test_one_pass_cross_entropy_overwrites_logits_with_their_gradient(B, V, ld, upstream):
    """``softmax_cross_entropy(..., inplace_backward=True)`` on bf16 logits that need a gradient: ONE kernel computes the
    loss and leaves (softmax - onehot) / B where the logits were (csrc/loss.hip ``ce_fused_kernel``: the row held in
    registers, 1..8 chunks per thread); backward multiplies by the upstream gradient only when it is not 1.  Against torch
    fp32, with ignored rows and padded columns."""
    from distributedpytorch_amd import ops
    torch.manual_seed(B + V)
    logits = (torch.randn(B, ld, device=DEV) * 3).to(torch.bfloat16)
    t = torch.randint(0, V, (B,), device=DEV)
    t[B // 2] = -100                                         # an ignored row
    assert _C().softmax_ce_fused_ok(logits, V)
    h = logits.clone().requires_grad_()
    lg = h * 1.0                                             # a non-leaf, as the LM head's output is
    loss = ops.softmax_cross_entropy(lg, t, V, inplace_backward=True)
    (loss * upstream).backward()
    lr = logits.float().cpu()[:, :V].requires_grad_()
    ref = F.cross_entropy(lr, t.cpu(), reduction="sum", ignore_index=-100) / B
    (ref * upstream).backward()
    _close(loss, ref, 1e-3, 1e-3, "one-pass ce loss")
    _close(h.grad[:, :V], lr.grad, 2e-2, 1e-5 * upstream, "one-pass ce grad")
    assert float(h.grad[B // 2].abs().max()) == 0.0
    if ld > V:
        assert float(h.grad[:, V:].abs().max()) == 0.0
    # and no gradient wanted (evaluation): the logits are left alone
    keep = logits.clone()
    with torch.no_grad():
        l2 = ops.softmax_cross_entropy(keep, t, V, inplace_backward=True)
    assert torch.equal(keep, logits)
    _close(l2, ref, 1e-3, 1e-3, "ce loss without grad")


def test_fused_sgd_matches_torch():
    n = 100003
    p = torch.randn(n, device=DEV); g = torch.randn(n, device=DEV); m = torch.randn(n, device=DEV)
    lp = torch.empty(n, dtype=torch.bfloat16, device=DEV)
    p0, g0, m0 = p.clone(), g.clone(), m.clone()
    _C().sgd_momentum(p, g, m, lp, 0.01, 0.5, 1e-4, 0.125, True, False, None, None)
    d = g0 * 0.125 + 1e-4 * p0
    mr = 0.5 * m0 + d
    pr = p0 - 0.01 * mr
    _close(p, pr, 1e-6, 1e-6, "sgd p")
    _close(m, mr, 1e-6, 1e-6, "sgd m")
    _close(lp, pr, 1e-2, 1e-3, "sgd bf16 shadow")
    assert float(g.abs().max()) == 0.0
    # per-segment weight decay table: [0, 64000) decays, the rest does not
    p = p0.clone(); g = g0.clone(); m = m0.clone()
    ends = torch.tensor([64000, n], dtype=torch.int64, device=DEV); wds = torch.tensor([1e-2, 0.0], device=DEV)
    _C().sgd_momentum(p, g, m, None, 0.01, 0.5, 0.0, 1.0, False, False, ends, wds)
    wd = torch.zeros(n, device=DEV); wd[:64000] = 1e-2
    mr = 0.5 * m0 + (g0 + wd * p0)
    _close(p, p0 - 0.01 * mr, 1e-6, 1e-6, "sgd segment table")


def test_mnist_ops_match_torch():
    """Fused conv+pool+relu / linear / log-softmax / nll vs the reference's op chain."""
    from distributedpytorch_amd import ops
    torch.manual_seed(0)
    for B in (12, 96):                      # ragged last batches of the reference configs
        x = torch.randn(B, 1, 28, 28, device=DEV, requires_grad=True)
        w1 = (torch.randn(10, 1, 5, 5, device=DEV) * 0.2).requires_grad_(); b1 = torch.randn(10, device=DEV).requires_grad_()
        w2 = (torch.randn(20, 10, 5, 5, device=DEV) * 0.1).requires_grad_(); b2 = torch.randn(20, device=DEV).requires_grad_()
        f1 = (torch.randn(50, 320, device=DEV) * 0.05).requires_grad_(); fb1 = torch.randn(50, device=DEV).requires_grad_()
        f2 = (torch.randn(10, 50, device=DEV) * 0.1).requires_grad_(); fb2 = torch.randn(10, device=DEV).requires_grad_()
        t = torch.randint(0, 10, (B,), device=DEV)
        h = ops.conv5_pool_relu(x, w1, b1)
        h = ops.conv5_pool_relu(h, w2, b2).reshape(-1, 320)
        h = ops.linear_f32(h, f1, fb1, relu=True)
        lp = ops.log_softmax(ops.linear_f32(h, f2, fb2))
        loss = ops.nll_loss(lp, t)
        loss.backward()
        params = [x, w1, b1, w2, b2, f1, fb1, f2, fb2]
        cp = [p.detach().cpu().requires_grad_() for p in params]
        xr, w1r, b1r, w2r, b2r, f1r, fb1r, f2r, fb2r = cp
        hr = F.relu(F.max_pool2d(F.conv2d(xr, w1r, b1r), 2))
        hr = F.relu(F.max_pool2d(F.conv2d(hr, w2r, b2r), 2)).view(-1, 320)
        hr = F.relu(F.linear(hr, f1r, fb1r))
        lpr = F.log_softmax(F.linear(hr, f2r, fb2r), dim=1)
        lossr = F.nll_loss(lpr, t.cpu())
        lossr.backward()
        _close(lp, lpr, 1e-4, 1e-4, "log-probs")
        _close(loss, lossr, 1e-4, 1e-5, "loss")
        for a, b, name in zip(params, cp, "x w1 b1 w2 b2 f1 fb1 f2 fb2".split()):
            _close(a.grad, b.grad, 1e-3, 1e-5, "grad " + name)


def test_mnist_fused_head_dropout_and_gradient_sink_match_torch():
    """The launch-lean forms of the same chain: Linear + log-softmax as one op (its backward folds the log-softmax
    gradient into the Linear's dx | dw launch), dropout drawn INSIDE the conv / Linear kernels (the stand-alone
    ``dropout_mask`` kernel reproduces the same counter-hash stream, so the stock-op reference can use its masks), and
    every weight gradient added straight into a FlatParams buffer (no tensors returned to autograd)."""
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.models import mnist
    from distributedpytorch_amd.parallel.flat import FlatParams
    C_ = _C()
    torch.manual_seed(1)
    B = 40
    net = mnist.Net(seed=77).to(DEV).train()
    flat = FlatParams(net, use_lp=False)
    flat.install_sink(None)
    flat.grad.fill_(0.25)                                  # gradients are ADDED to what is there
    x = torch.randn(B, 1, 28, 28, device=DEV, requires_grad=True)
    t = torch.randint(0, 10, (B,), device=DEV)
    net._drop.step = 4                                     # host-side counter: the next draws are steps 5 and 6
    off2d, off1d = 5 * 1_000_003 + 0, 6 * 1_000_003 + 7
    loss = mnist.nll_loss(net(x), t)
    loss.backward()
    torch.cuda.synchronize()
    m2d = C_.dropout_mask(B * 20, 0.5, 77, off2d, x, None).cpu().view(B, 20, 1, 1)
    m1d = C_.dropout_mask(B * 50, 0.5, 77, off1d, x, None).cpu().view(B, 50)
    assert 0.3 < float((m2d > 0).float().mean()) < 0.7 and 0.3 < float((m1d > 0).float().mean()) < 0.7
    ref = {n: p.detach().cpu().clone().requires_grad_() for n, p in net.named_parameters()}
    xr = x.detach().cpu().requires_grad_()
    h = F.relu(F.max_pool2d(F.conv2d(xr, ref["conv1.weight"], ref["conv1.bias"]), 2))
    h = F.relu(F.max_pool2d(F.conv2d(h, ref["conv2.weight"], ref["conv2.bias"]) * m2d, 2)).view(-1, 320)
    h = F.relu(F.linear(h, ref["fc1.weight"], ref["fc1.bias"])) * m1d
    lossr = F.nll_loss(F.log_softmax(F.linear(h, ref["fc2.weight"], ref["fc2.bias"]), dim=1), t.cpu())
    lossr.backward()
    _close(loss, lossr, 1e-4, 1e-5, "loss with in-kernel dropout")
    _close(x.grad, xr.grad, 1e-3, 1e-6, "grad x")
    for n, p in net.named_parameters():
        a, b = flat.slice_of(p)
        _close(flat.grad[a:b].view(p.shape) - 0.25, ref[n].grad, 1e-3, 2e-5, "sunk grad " + n)
    # eval mode: no dropout, same op
    net.eval()
    with torch.no_grad():
        lp = net(x)
    hr = F.relu(F.max_pool2d(F.conv2d(xr, ref["conv1.weight"], ref["conv1.bias"]), 2))
    hr = F.relu(F.max_pool2d(F.conv2d(hr, ref["conv2.weight"], ref["conv2.bias"]), 2)).view(-1, 320)
    hr = F.relu(F.linear(hr, ref["fc1.weight"], ref["fc1.bias"]))
    _close(lp, F.log_softmax(F.linear(hr, ref["fc2.weight"], ref["fc2.bias"]), dim=1).detach(), 1e-4, 1e-4, "eval log-probs")


def test_mnist_fused_head_loss_matches_unfused_path_and_torch():
    """``Net.loss`` (head + log-softmax + mean NLL: one kernel forward, one backward - csrc/smallnet.hip
    ``linear_lsm_nll_fwd_kernel`` and ``linear_bwd_kernel``'s lsm = 2 mode) against ``nll_loss(net(x))`` through the
    separate ops and against stock torch fp32; ragged row counts; the device dropout counter is bumped by the forward
    kernel exactly once per call."""
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.models import mnist
    torch.manual_seed(3)
    for rows, inn, out in ((128, 50, 10), (37, 50, 10), (200, 320, 16), (1, 7, 3)):
        h = torch.randn(rows, inn, device=DEV, requires_grad=True)
        w = (torch.randn(out, inn, device=DEV) * 0.2).requires_grad_()
        b = torch.randn(out, device=DEV, requires_grad=True)
        t = torch.randint(0, out, (rows,), device=DEV)
        ctr = torch.zeros(1, dtype=torch.int64, device=DEV)
        loss, logp = ops.linear_log_softmax_nll(h, w, b, t, ctr)
        assert not logp.requires_grad
        (loss * 3.0).backward()
        hr, wr, br = (v.detach().cpu().double().requires_grad_() for v in (h, w, b))
        lpr = F.log_softmax(F.linear(hr, wr, br), dim=1)
        lossr = F.nll_loss(lpr, t.cpu())
        (lossr * 3.0).backward()
        _close(loss, lossr.float(), 1e-5, 1e-5, f"fused loss {rows}x{inn}x{out}")
        _close(logp, lpr.detach().float(), 1e-5, 1e-5, "log-probs")
        _close(h.grad, hr.grad.float(), 1e-4, 1e-6, "grad h")
        _close(w.grad, wr.grad.float(), 1e-4, 1e-6, "grad w")
        _close(b.grad, br.grad.float(), 1e-4, 1e-6, "grad b")
        assert int(ctr.item()) == 1
    # whole net: fused loss == separate ops, same dropout draws (host counter rewound between the two)
    net = mnist.Net(seed=5).to(DEV).train()
    x = torch.randn(64, 1, 28, 28, device=DEV)
    t = torch.randint(0, 10, (64,), device=DEV)
    net._drop.step = 10
    l1 = net.loss(x, t); l1.backward()
    g1 = [p.grad.clone() for p in net.parameters()]
    net.zero_grad(set_to_none=True)
    net._drop.step = 10
    l2 = mnist.nll_loss(net(x), t); l2.backward()
    _close(l1, l2, 1e-6, 1e-6, "Net.loss vs nll_loss(net(x))")
    for a, p in zip(g1, net.parameters()):
        _close(a, p.grad, 1e-4, 1e-6, "Net.loss grads")


def test_batched_filter_transpose_matches_permute_on_every_twin():
    """``FlatParams.refresh_transposed`` (ONE launch of ``transpose_filters_batched_kernel`` for the whole model, 64 x 64
    tiles, 16-byte path and scalar path) against ``permute`` of the bf16 shadow, for every layer that has a transposed twin:
    conv filters [K,R,S,C] -> [C,R,S,K] of a ResNet (7x7 stem, 3x3, 1x1, the 1000-way head) and GPT-2's Linear weights and
    tied embedding matrix; plus a table of odd shapes straight through the kernel."""
    from distributedpytorch_amd.models import resnet, gpt2
    from distributedpytorch_amd.parallel.flat import FlatParams
    C_ = _C()
    for model in (resnet.tiny_resnet(), gpt2.gpt2_tiny(n_layer=1, n_embd=64, n_head=2, vocab=1000, n_positions=64)):
        model = model.to(DEV)
        flat = FlatParams(model)
        assert flat.lp_t is not None
        seen = 0
        for mod in model.modules():
            for pname in FlatParams._twin_names(mod):
                w = getattr(mod, pname, None)
                twin = getattr(mod, pname + "_lp_t", None)
                if w is None or twin is None:
                    continue
                a, b = flat.slice_of(w)
                shadow = flat.lp[a:b].view(w.shape)
                ref = shadow.permute(3, 1, 2, 0) if w.dim() == 4 else shadow.t()
                assert torch.equal(twin, ref.contiguous()), (type(mod).__name__, pname, tuple(w.shape))
                seen += 1
        assert seen >= 3
    # odd shapes: K, C not multiples of 8 (scalar path), ragged tiles, misaligned offsets
    shapes = [(70, 1, 130), (64, 9, 64), (13, 4, 5), (200, 1, 72), (8, 1, 8)]
    offs, rows, tiles, tt = 3, [], 0, C_.transpose_tile()          # start misaligned on purpose
    for K, RS, Cc in shapes:
        rows.append([offs, offs, K, RS, Cc, tiles])
        tiles += -(-K // tt) * -(-Cc // tt) * RS
        offs += K * RS * Cc
        offs += (-offs) % 8 if K % 8 == 0 and Cc % 8 == 0 else 1
    src = torch.randn(offs + 64, device=DEV).to(torch.bfloat16)
    dst = torch.zeros_like(src)
    C_.transpose_filters_batched(src, dst, torch.tensor(rows, dtype=torch.int64, device=DEV), tiles)
    for (o, _, K, RS, Cc, _t) in rows:
        ref = src[o:o + K * RS * Cc].view(K, RS, Cc).permute(2, 1, 0).contiguous()
        assert torch.equal(dst[o:o + K * RS * Cc].view(Cc, RS, K), ref), (K, RS, Cc)


def test_lm_head_shaped_linear_matches_fp32_reference_on_both_gemm_kernels():
    """ops.linear on an LM-head-shaped problem (wide vocabulary, nothing fused): forward, input gradient and weight
    gradient against the fp32 expression, once through csrc/gemm_wide.hip (forced) and once through the 128 x 128
    kernel.  No vendor BLAS anywhere (the library path of round 1 is gone)."""
    from distributedpytorch_amd import ops
    C_ = _C()
    M, K, N = 384, 128, 1000 + 24
    x0 = _rand_bf16(3, M // 3, K, seed=51)
    w0 = (torch.randn(N, K, generator=torch.Generator().manual_seed(52)) * 0.05).to(DEV)
    dy = _rand_bf16(3, M // 3, N, scale=0.1, seed=53)
    got = []
    keep = C_.wide_gemm_min_tiles()
    try:
        for wide in (1, 0):
            C_.set_wide_gemm_min_tiles(wide)
            x = x0.clone().requires_grad_(); w = w0.clone().requires_grad_()
            y = ops.linear(x, w, None, None)
            y.backward(dy)
            got.append((y.detach().float(), x.grad.float(), w.grad.float()))
    finally:
        C_.set_wide_gemm_min_tiles(keep)
    wr = w0.to(torch.bfloat16).float()
    yr = x0.float() @ wr.t()
    dxr = dy.float() @ wr
    dwr = dy.float().reshape(-1, N).t() @ x0.float().reshape(-1, K)
    for y, dx, dw in got:
        _close(y, yr, 2e-2, 2e-2, "lm head y")
        _close(dx, dxr, 2e-2, 2e-2, "lm head dx")
        _close(dw, dwr, 2e-2, 2e-2, "lm head dw")


def test_gelu_twin_output_equals_the_elementwise_pass():
    """conv_fprop(gelu_twin=True): the epilogue's second output is gelu_tanh of the bf16 values it stores - bit-equal
    to the elementwise kernel it replaces - and the MLP built on it matches the fp32 expression."""
    from distributedpytorch_amd import ops
    C_ = _C()
    M, K, N = 200, 64, 136
    x = _rand_bf16(M, 1, 1, K, seed=71); w = _rand_bf16(N, 1, 1, K, scale=0.2, seed=72)
    bias = torch.randn(N, device=DEV) * 0.1
    y, act = C_.conv_fprop(x, w, bias, 1, 0, 0, False, None, None, True)
    y0 = C_.conv_fprop(x, w, bias, 1, 0, 0, False, None)[0]
    assert torch.equal(y, y0) and torch.equal(act, C_.ew2(y0, y0, 3))
    ref = torch.nn.functional.gelu(x.float().view(M, K) @ w.float().view(N, K).t() + bias, approximate="tanh")
    _close(act.float().view(M, N), ref, 2e-2, 2e-2, "gelu twin")
    # through ops.linear: (h, gelu(h)) then the consumer with pre_gelu + gelu_act; gradients reach x and both weights
    xs = _rand_bf16(4, 50, K, seed=73).requires_grad_()
    w1 = (torch.randn(N, K, device=DEV) * 0.1).requires_grad_(); b1 = torch.zeros(N, device=DEV, requires_grad=True)
    w2 = (torch.randn(K, N, device=DEV) * 0.1).requires_grad_()
    h, a = ops.linear(xs, w1, None, b1, gelu_twin=True)
    out = ops.linear(h, w2, None, None, pre_gelu=True, gelu_act=a)
    out.float().square().sum().backward()
    xr = xs.detach().float().requires_grad_(); w1r = w1.detach().clone().requires_grad_(); w2r = w2.detach().clone().requires_grad_()
    hr = xr @ w1r.to(torch.bfloat16).float().t()
    outr = torch.nn.functional.gelu(hr, approximate="tanh") @ w2r.to(torch.bfloat16).float().t()
    outr.square().sum().backward()
    _close(out.float(), outr, 3e-2, 3e-2, "mlp out")
    sc = float(w1r.grad.abs().max())
    _close(w1.grad, w1r.grad, 5e-2, 5e-2 * sc, "dW1"); _close(xs.grad.float(), xr.grad, 5e-2, 5e-2 * float(xr.grad.abs().max()), "dx")
    _close(w2.grad, w2r.grad, 5e-2, 5e-2 * float(w2r.grad.abs().max()), "dW2")


def test_layernorm_backward_deferred_fold_equals_direct():
    """layernorm_bwd(defer_reduce=True) + ln_fold_partials (what ops/norm.py runs on the side stream) adds the same
    parameter gradients as the one-call form."""
    C_ = _C()
    rows, C = 300, 768
    x = _rand_bf16(rows, C, seed=61); dy = _rand_bf16(rows, C, scale=0.3, seed=62)
    g = torch.rand(C, device=DEV) + 0.5; b = torch.zeros(C, device=DEV)
    _, mean, rstd = C_.layernorm_fwd(x, g, b, 1e-5)
    dg1 = torch.full((C,), 0.25, device=DEV); db1 = torch.full((C,), -0.5, device=DEV)
    dg2, db2 = dg1.clone(), db1.clone()
    dx1 = C_.layernorm_bwd(dy, x, g, mean, rstd, dg1, db1)[0]
    dx2, part = C_.layernorm_bwd(dy, x, g, mean, rstd, None, None, None, True)
    C_.ln_fold_partials(part, dg2, db2)
    assert torch.equal(dx1, dx2)
    _close(dg2, dg1, 1e-5, 1e-4, "dgamma"); _close(db2, db1, 1e-5, 1e-4, "dbeta")
    xh = (x.float() - mean[:, None]) * rstd[:, None]
    _close(dg2 - 0.25, (dy.float() * xh).sum(0), 1e-2, 2e-2, "dgamma vs fp32")


def test_layernorm_gelu_embedding():
    from distributedpytorch_amd import ops
    x = _rand_bf16(37, 768, seed=21).requires_grad_()
    g = (torch.rand(768, device=DEV) + 0.5).requires_grad_(); b = torch.randn(768, device=DEV).requires_grad_()
    y = ops.layer_norm(x, g, b)
    dy = _rand_bf16(37, 768, seed=22)
    y.backward(dy)
    xr = x.detach().float().cpu().requires_grad_(); gr = g.detach().cpu().requires_grad_(); br = b.detach().cpu().requires_grad_()
    yr = F.layer_norm(xr, (768,), gr, br)
    yr.backward(dy.float().cpu())
    _close(y, yr, 1e-2, 2e-2, "ln")
    _close(x.grad, xr.grad, 2e-2, 2e-2, "ln dx")
    _close(g.grad, gr.grad, 1e-2, 5e-2, "ln dg")
    _close(b.grad, br.grad, 1e-2, 5e-2, "ln db")
    x = _rand_bf16(1000, 3072 // 8 * 8 + 8, seed=23).requires_grad_()
    y = ops.gelu(x); y.backward(torch.ones_like(y))
    xr = x.detach().float().cpu().requires_grad_()
    yr = F.gelu(xr, approximate="tanh"); yr.backward(torch.ones_like(yr))
    _close(y, yr, 1e-2, 1e-2, "gelu")
    _close(x.grad, xr.grad, 1e-2, 1e-2, "gelu grad")
    V, T, Cc = 500, 32, 128
    wte = torch.randn(V, Cc, device=DEV).requires_grad_(); wpe = torch.randn(64, Cc, device=DEV).requires_grad_()
    tok = torch.randint(0, V, (3, T), device=DEV)
    e = ops.embedding(tok, wte, wpe)
    de = _rand_bf16(3, T, Cc, seed=24)
    e.backward(de)
    wter = wte.detach().cpu().to(torch.bfloat16).float().requires_grad_(); wper = wpe.detach().cpu().to(torch.bfloat16).float().requires_grad_()
    er = wter[tok.cpu()] + wper[:T].unsqueeze(0)
    er.backward(de.float().cpu())
    _close(e, er, 1e-2, 1e-2, "embedding")
    _close(wte.grad, wter.grad, 1e-3, 1e-3, "dwte")
    _close(wpe.grad, wper.grad, 1e-3, 1e-3, "dwpe")


# ragged lengths (T % 128 != 0: 1, 200 and 1000 - the analogue of the reference's ragged last batches) and both head sizes
@pytest.mark.parametrize("B,T,H,D", [(1, 128, 1, 64), (2, 256, 3, 64), (1, 512, 2, 64), (2, 200, 2, 64), (1, 1000, 2, 64),
                                     (1, 1, 1, 64), (1, 65, 2, 64), (2, 256, 2, 128), (1, 200, 3, 128), (1, 1000, 1, 128)])
def test_causal_attention_fwd_bwd(B, T, H, D):
    """Flash attention kernels vs the fp32 softmax(QK^T/sqrt(d))V oracle (ops/tfm.py CPU path)."""
    from distributedpytorch_amd import ops
    qkv = _rand_bf16(B, T, 3 * H * D, seed=31).requires_grad_()
    out = ops.causal_attention(qkv, H)
    dout = _rand_bf16(B, T, H * D, seed=32)
    out.backward(dout)
    qr = qkv.detach().float().cpu().requires_grad_()
    q, k, v = qr.view(B, T, 3, H, D).permute(2, 0, 3, 1, 4)
    att = (q @ k.transpose(-2, -1)) / math.sqrt(D)
    mask = torch.ones(T, T, dtype=torch.bool).tril()
    att = att.masked_fill(~mask, float("-inf")).softmax(dim=-1)
    outr = (att @ v).permute(0, 2, 1, 3).reshape(B, T, H * D)
    outr.backward(dout.float().cpu())
    _close(out, outr, 2e-2, 2e-2, "attention out")
    g, gr = qkv.grad.float().cpu().view(B, T, 3, H, D), qr.grad.view(B, T, 3, H, D)
    for i, name in enumerate(("dq", "dk", "dv")):
        # P and dS are rounded to bf16 before the second MFMA product
        _close(g[:, :, i], gr[:, :, i], 3e-2, 3e-2, "attention " + name)


@pytest.mark.parametrize("B,T,H", [(1, 1, 1), (1, 65, 2), (2, 200, 2), (2, 256, 3), (1, 1000, 2), (2, 1024, 4)])
def test_causal_attention_pipelined_forward_kernel(B, T, H):
    """The opt-in software-pipelined forward kernel (csrc/attention.hip ``attn_fwd_pipe_kernel``: half tiles, scores of
    the next half tile issued in front of the softmax of the current one, K ring of three LDS tiles) against the fp32
    oracle and against the default kernel's log-sum-exp, ragged lengths included."""
    C, D = _C(), 64
    qkv = _rand_bf16(B, T, 3 * H * D, seed=51).view(B, T, 3, H, D)
    out0, lse0 = C.attention_fwd(qkv)
    C.set_attention_fwd_pipe(True)
    try:
        out1, lse1 = C.attention_fwd(qkv)
        torch.cuda.synchronize()
    finally:
        C.set_attention_fwd_pipe(False)
    q, k, v = qkv.float().cpu().permute(2, 0, 3, 1, 4)
    att = (q @ k.transpose(-2, -1)) / math.sqrt(D)
    att = att.masked_fill(~torch.ones(T, T, dtype=torch.bool).tril(), float("-inf"))
    ref = (att.softmax(dim=-1) @ v).permute(0, 2, 1, 3).reshape(B, T, H * D)
    _close(out1.reshape(B, T, H * D), ref, 2e-2, 2e-2, "pipelined attention out")
    _close(lse1, torch.logsumexp(att, dim=-1), 1e-3, 1e-3, "pipelined attention lse")
    _close(lse1, lse0, 1e-4, 1e-4, "pipelined vs default lse")
    _close(out1.reshape(B, T, H * D), out0.reshape(B, T, H * D), 1e-2, 1e-2, "pipelined vs default out")


@pytest.mark.parametrize("B,T,H,D", [(2, 200, 3, 64), (1, 256, 2, 64), (1, 130, 2, 128)])
def test_causal_attention_with_dropout_matches_masked_reference(B, T, H, D):
    """Attention dropout drawn INSIDE the three kernels (csrc/attention.hip ``AttnDrop``): the kept / dropped pattern is
    the counter hash at index ((b*H + h)*T + q)*T + k, which ``dropout_mask`` reproduces as a [B, H, T, T] tensor for
    the fp32 oracle ``(softmax(QK^T/sqrt(d)) * mask) V``; forward and all three gradients."""
    from distributedpytorch_amd import ops
    p, seed, offset = 0.25, 99, 5 << 36
    qkv = _rand_bf16(B, T, 3 * H * D, seed=41).requires_grad_()
    out = ops.causal_attention(qkv, H, drop=(p, seed, offset))
    dout = _rand_bf16(B, T, H * D, seed=42)
    out.backward(dout)
    keep = _C().dropout_mask(B * H * T * T, p, seed, offset, qkv, None).cpu().view(B, H, T, T)
    assert 0.65 < float((keep > 0).float().mean()) < 0.85
    qr = qkv.detach().float().cpu().requires_grad_()
    q, k, v = qr.view(B, T, 3, H, D).permute(2, 0, 3, 1, 4)
    att = (q @ k.transpose(-2, -1)) / math.sqrt(D)
    att = att.masked_fill(~torch.ones(T, T, dtype=torch.bool).tril(), float("-inf")).softmax(dim=-1) * keep
    outr = (att @ v).permute(0, 2, 1, 3).reshape(B, T, H * D)
    outr.backward(dout.float().cpu())
    _close(out, outr, 2e-2, 2e-2, "attention+dropout out")
    g, gr = qkv.grad.float().cpu().view(B, T, 3, H, D), qr.grad.view(B, T, 3, H, D)
    for i, name in enumerate(("dq", "dk", "dv")):
        _close(g[:, :, i], gr[:, :, i], 3e-2, 4e-2, "attention+dropout " + name)
    # the no-dropout path is untouched by the feature
    o0 = ops.causal_attention(qkv.detach(), H)
    o1 = ops.causal_attention(qkv.detach(), H, drop=(0.0, seed, offset))
    assert torch.equal(o0, o1)


def test_dropout_residual_bf16_and_gpt2_with_dropout():
    """``ops.dropout_residual`` (mask drawn in line, fused with the residual add) against the explicit mask, forward and
    backward; then a tiny GPT-2 with ``dropout = 0.1``: trains (finite loss and gradients), draws fresh masks per call,
    and is deterministic in eval mode."""
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.models import gpt2
    p, seed, offset = 0.1, 7, 3 << 36
    x = _rand_bf16(6, 40, 128, seed=51).requires_grad_()
    r = _rand_bf16(6, 40, 128, seed=52).requires_grad_()
    y = ops.dropout_residual(x, p, seed, offset, residual=r)
    dy = _rand_bf16(6, 40, 128, seed=53)
    y.backward(dy)
    keep = _C().dropout_mask(x.numel(), p, seed, offset, x, None).view(x.shape)
    _close(y, x.detach().float() * keep + r.detach().float(), 1e-2, 1e-2, "dropout_residual")
    _close(x.grad, dy.float() * keep, 1e-2, 1e-2, "dropout_residual dx")
    assert torch.equal(r.grad, dy)
    assert 0.85 < float((keep > 0).float().mean()) < 0.95
    torch.manual_seed(0)
    m = gpt2.gpt2_tiny(n_layer=2, n_embd=128, n_head=2, vocab=500, n_positions=128, dropout=0.1).to(DEV).train()
    tok = torch.randint(0, 500, (2, 128), device=DEV)
    l1 = m.loss(tok, tok); l1.backward()
    assert torch.isfinite(l1) and all(torch.isfinite(q.grad).all() for q in m.parameters() if q.grad is not None)
    l2 = m.loss(tok, tok)
    m.eval()
    with torch.no_grad():
        e1, e2 = m.loss(tok, tok), m.loss(tok, tok)
    assert abs(float(e1) - float(e2)) < 1e-4 and abs(float(l1) - float(l2)) > 1e-4     # (loss rows are summed with float atomics)


@pytest.mark.parametrize("N,H,W", [(4, 64, 64), (2, 50, 38), (3, 224, 224)])
def test_pair_packed_stem_matches_the_channel_padded_stem(N, H, W):
    """The stem on the pixel-pair layout (``nchw_to_pairs`` + 7 x 4-tap convolution with stride (2, 1) / padding (3, 2) +
    packed weight gradient, ops/block.py ``_StemPackedFn``) against the same node on the 8-channel-padded image: output,
    filter gradient (in the parameter's [K,7,7,8] layout; padded channels stay zero) and BatchNorm gradients.  Also the
    layout kernels against their index definitions."""
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.models import resnet
    from distributedpytorch_amd.ops import block as _block
    from distributedpytorch_amd.parallel.flat import FlatParams
    C_ = _C()
    torch.manual_seed(N + H)
    x = torch.randn(N, 3, H, W, device=DEV)
    xp = C_.nchw_to_pairs(x)
    ref = torch.zeros(N, H, W // 2, 8, device=DEV)
    ref[..., 0:3] = x[:, :, :, 0::2].permute(0, 2, 3, 1)
    ref[..., 3:6] = x[:, :, :, 1::2].permute(0, 2, 3, 1)
    assert torch.equal(xp, ref.to(torch.bfloat16))
    stem = resnet.Stem().to(DEV).train()
    flat = FlatParams(stem)
    assert _block.can_fuse_stem_packed(stem, x)
    w = stem.conv1.weight_lp                                             # [64, 7, 7, 8] bf16, channels 3.. zero
    wp = C_.stem_pack_filter(w, 3)
    for sp in range(4):
        for half in range(2):
            s_ = 2 * sp + half - 1
            want = w[:, :, s_, :3] if s_ >= 0 else torch.zeros_like(w[:, :, 0, :3])
            assert torch.equal(wp[:, :, sp, half * 3:half * 3 + 3], want)
    assert float(wp[..., 6:].abs().max()) == 0.0

    def run(packed):
        flat.grad.zero_()
        y = _block.stem_packed(stem, x) if packed else _block.stem(stem, ops.to_nhwc(x, 8))
        g = torch.Generator(device=DEV).manual_seed(5)
        dy = torch.randn(y.shape, generator=g, device=DEV).to(torch.bfloat16)
        y.backward(dy)
        torch.cuda.synchronize()
        return y.detach().float(), [p.grad.detach().clone() for p in (stem.conv1.weight, stem.bn1.weight, stem.bn1.bias)]

    y0, g0 = run(False)
    y1, g1 = run(True)
    assert y1.shape == y0.shape
    _close(y1, y0, 2e-2, 2e-2, "packed stem output", max_bad_frac=1e-3)
    scale = float(g0[0].abs().max())
    assert float((g1[0] - g0[0]).abs().max()) <= 2e-2 * scale, (float((g1[0] - g0[0]).abs().max()), scale)
    assert float(g1[0][..., 3:].abs().max()) == 0.0
    for a, b, name in ((g1[1], g0[1], "dgamma"), (g1[2], g0[2], "dbeta")):
        assert float((a - b).abs().max()) <= 3e-2 * float(b.abs().max()) + 1e-3, name


def test_bn_apply_with_affine_residual_matches_fp32_reference():
    """bn_apply(x, r, ..., res_scale, res_shift): relu(x*s + b + (r*rs + rb)) - the projection shortcut's BatchNorm folded
    into bn3's apply (ops/block.py fold_ds) - against the plain fp32 expression."""
    C_ = _C()
    N, H, W, C = 3, 7, 9, 72
    x = _rand_bf16(N, H, W, C, seed=41); r = _rand_bf16(N, H, W, C, seed=42)
    s, b, rs, rb = (torch.randn(C, device=DEV) for _ in range(4))
    y, mask = C_.bn_apply(x, r, s, b, True, True, rs, rb)
    ref = torch.relu(x.float() * s + b + (r.float() * rs + rb))
    _close(y.float(), ref, 1e-2, 1e-2, "bn_apply affine residual")
    # the two-pass form it replaces rounds the shortcut to bf16 in between: agree within that rounding
    idn, _ = C_.bn_apply(r, None, rs, rb, False, False)
    y2, mask2 = C_.bn_apply(x, idn, s, b, True, True)
    _close(y.float(), y2.float(), 2e-2, 3e-2, "bn_apply vs two-pass")
    with pytest.raises(RuntimeError):
        C_.bn_apply(x, None, s, b, True, True, rs, rb)


@pytest.mark.parametrize("shape", [(4, 32, 32, 64), (3, 17, 23, 16)])
def test_fused_bn_relu_maxpool_matches_unfused_ops(shape):
    """bn_relu_maxpool_fwd / pool_bn_backward (the stem node) against bn_apply + maxpool and their backwards."""
    N, H, W, C = shape
    C_ = _C()
    x = _rand_bf16(N, H, W, C, seed=31)
    gamma = torch.rand(C, device=DEV) + 0.5
    beta = torch.randn(C, device=DEV) * 0.3
    xf = x.float()
    mean = xf.mean((0, 1, 2)); var = xf.var((0, 1, 2), unbiased=False)
    invstd = torch.rsqrt(var + 1e-5)
    scale = (gamma * invstd).contiguous(); shift = (beta - mean * gamma * invstd).contiguous()
    # unfused reference chain on the same kernels
    a, mask = C_.bn_apply(x, None, scale, shift, True, True)
    y_ref, idx_ref = C_.maxpool_fwd(a)
    y, idx = C_.bn_relu_maxpool_fwd(x, scale, shift)
    assert torch.equal(y, y_ref)
    dp = _rand_bf16(*y.shape, scale=0.2, seed=32)
    da = C_.maxpool_bwd(dp, idx_ref, H, W)
    dx_ref, dg_ref, db_ref = C_.bn_backward(da, mask, x, gamma, mean, invstd, True, None, None, None)
    sbuf = torch.zeros(32, 2, C, device=DEV)
    dx, dg, db = C_.pool_bn_backward(dp, idx, x, gamma, mean, invstd, None, None, sbuf)
    assert float(sbuf.abs().max()) == 0.0                       # finalize re-zeroed the accumulator
    # the unfused chain rounds the un-pooled gradient (sums of up to 4 pooled gradients) to bf16 before reducing
    # thousands of them; the fused kernels reduce the fp32 sums - compare at the scale of the sums
    sc = float(dg_ref.abs().max())
    _close(dg, dg_ref, 1e-2, 5e-3 * sc, "dgamma")
    _close(db, db_ref, 1e-2, 5e-3 * max(sc, float(db_ref.abs().max())), "dbeta")
    _close(dx, dx_ref.float(), 2e-2, 3e-2, "dx", max_bad_frac=1e-3)


@pytest.mark.parametrize("case", [(2, 14, 14, 128, 64, 1, 0, 1), (2, 14, 14, 64, 64, 3, 1, 1), (3, 9, 11, 256, 136, 1, 0, 1),
                                  (2, 14, 14, 64, 128, 3, 1, 2), (2, 15, 13, 128, 64, 1, 0, 2)])
def test_dgrad_with_fused_bn_backward_statistics(case):
    """conv_dgrad(..., bn_x, bn_mask, bn_stats) + bn_bwd_coefs + bn_bwd_apply == conv_dgrad + bn_backward."""
    N, H, W, K, Cc, R, pad, st = case                 # dy has K channels, the gradient da / BatchNorm has Cc
    C_ = _C()
    Ho, Wo = (H + 2 * pad - R) // st + 1, (W + 2 * pad - R) // st + 1
    dy = _rand_bf16(N, Ho, Wo, K, scale=0.1, seed=41)
    w = _rand_bf16(K, R, R, Cc, scale=0.1, seed=42)
    wt = C_.transpose_filter(w)
    x = _rand_bf16(N, H, W, Cc, seed=43)              # the BatchNorm's raw input
    gamma = torch.rand(Cc, device=DEV) + 0.5
    xf = x.float()
    mean = xf.mean((0, 1, 2)); invstd = torch.rsqrt(xf.var((0, 1, 2), unbiased=False) + 1e-5)
    bits = torch.rand(N, H, W, Cc, generator=torch.Generator().manual_seed(44)) > 0.4
    mask = (bits.view(-1, 8).to(torch.uint8) << torch.arange(8, dtype=torch.uint8)).sum(1).to(torch.uint8).to(DEV)
    da_ref = C_.conv_dgrad(dy, wt, H, W, st, pad, None, None)
    dx_ref, dg_ref, db_ref = C_.bn_backward(da_ref, mask, x, gamma, mean, invstd, True, None, None, None)
    sb = torch.zeros(32, 2, Cc, device=DEV)
    da = C_.conv_dgrad(dy, wt, H, W, st, pad, None, None, x, mask, sb)
    assert torch.equal(da, da_ref)
    co = C_.bn_bwd_coefs(sb, N * H * W, gamma, mean, invstd, None, None)
    assert float(sb.abs().max()) == 0.0
    dx = C_.bn_bwd_apply(da, mask, x, co, True)
    sc = max(float(dg_ref.abs().max()), float(db_ref.abs().max()))
    _close(co[0], dg_ref, 1e-2, 5e-3 * sc, "dgamma")
    _close(co[1], db_ref, 1e-2, 5e-3 * sc, "dbeta")
    _close(dx, dx_ref.float(), 2e-2, 2e-2, "dx", max_bad_frac=1e-3)


def test_dgrad_with_gelu_backward_epilogue():
    """conv_dgrad(..., gelu_x=h) == conv_dgrad(...) * gelu'(h) (the ew2 mode-4 kernel)."""
    C_ = _C()
    M, K, Cc = 512, 256, 768
    dy = _rand_bf16(M, 1, 1, K, scale=0.2, seed=61)
    wt = C_.transpose_filter(_rand_bf16(K, 1, 1, Cc, scale=0.1, seed=62))
    h = _rand_bf16(M, 1, 1, Cc, scale=1.5, seed=63)
    ref = C_.ew2(C_.conv_dgrad(dy, wt, 1, 1, 1, 0, None, None), h, 4)
    got = C_.conv_dgrad(dy, wt, 1, 1, 1, 0, None, None, None, None, None, h)
    _close(got, ref.float(), 2e-2, 2e-3, "gelu-backward epilogue", max_bad_frac=1e-3)
