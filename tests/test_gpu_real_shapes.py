"""Numerics at the shapes the benchmarks run (VERDICT r1 "pin numerics at real shapes").

* One downsampling + one identity ResNet-50 bottleneck of EACH stage at batch 256 against a stock-PyTorch fp32
  evaluation of the same two blocks on the GPU (``F.conv2d`` / ``F.batch_norm``): block output, input gradient
  and every parameter gradient.  This is where the 12,544-workgroup grids, the split factors of the real weight
  gradients, the cross-block BatchNorm statistics and the fused bottleneck tail (csrc/tail_bwd.hip) are exercised -
  none of which a 5-image test reaches.
* One GPT-2 small block at 8 x 1024 tokens against an fp32 evaluation with stock ops.
* 50 SGD steps of ResNet-50 at batch 256, fused vs one-autograd-node-per-op: same loss within 1 %.
* tiny_resnet: our bf16 pipeline against the fp32 gradient, next to a stock bf16-autocast model against the same
  fp32 gradient - the "11-12 % from fp32" of round 1 is bf16-inherent if stock bf16 lands equally far.

Tolerances: bf16 storage rounds every activation to 8 bits of mantissa.  Forward that costs < 1 % (relative L2); in
backward every BatchNorm removes the mean and the xhat-projection of the gradient, so what is left is a difference of
large terms and ANY bf16 pipeline lands 3-12 % from the fp32 gradient at these depths.  The ResNet assertions therefore
use two oracles: fp32 for the forward output, and "no further from fp32 than stock PyTorch under bf16 autocast" for the
gradients (observed: ours is 2-4 % CLOSER than stock on every tensor).  Observed values are recorded beside each assert.
"""
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm() / (b.norm() + 1e-20)).item()


# ------------------------------------------------------------------------------------------ ResNet-50 bottlenecks
def _smooth_weights(shape):
    """Positive, separable weights r[n,h,w,c] = a[n] b[h] c[w] d[c] for the scalar objective sum(out * r).  NOT white
    noise: with zero-mean random weights every gradient is a random-walk sum over 800k pixels, and the handful of ReLU
    bits that differ between a bf16 and an fp32 forward (activations within rounding of zero) then move it by
    sqrt(fraction flipped) - 5-15 % at 0.5 % flipped bits, measured - which tests the objective's conditioning, not
    the kernels.  Real losses give coherent gradients; so does this one."""
    g = torch.Generator(device="cpu").manual_seed(17)
    r = torch.ones(shape)
    for dim, n in enumerate(shape):
        v = torch.rand(n, generator=g) + 0.5
        view = [1] * len(shape); view[dim] = n
        r = r * v.view(view)
    return r.to(torch.bfloat16).to(DEV)


def _ref_conv_bn(x, conv, bn, relu, residual=None):
    w = conv.weight.detach().float()[..., : conv.in_channels].permute(0, 3, 1, 2).contiguous().requires_grad_()
    g = bn.weight.detach().float().clone().requires_grad_()
    b = bn.bias.detach().float().clone().requires_grad_()
    y = F.conv2d(x, w, None, conv.stride, conv.padding)
    y = F.batch_norm(y, None, None, g, b, True, 0.1, bn.eps)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return y, [(conv.weight, w, "conv"), (bn.weight, g, "bn"), (bn.bias, b, "bn")]


def _ref_block(x, blk):
    leaves = []
    identity = x
    if blk.downsample is not None:
        identity, l = _ref_conv_bn(x, blk.downsample[0], blk.downsample[1], False); leaves += l
    y, l = _ref_conv_bn(x, blk.conv1, blk.bn1, True); leaves += l
    y, l = _ref_conv_bn(y, blk.conv2, blk.bn2, True); leaves += l
    y, l = _ref_conv_bn(y, blk.conv3, blk.bn3, True, identity); leaves += l
    return y, leaves


STAGES = [  # inplanes, planes, stride, H (input spatial size) -- ResNet-50 at 224 x 224
    (64, 64, 1, 56),
    (256, 128, 2, 56),
    (512, 256, 2, 28),
    (1024, 512, 2, 14),
]


@pytest.mark.parametrize("inplanes,planes,stride,H", STAGES)
def test_resnet50_stage_blocks_at_batch_256_match_fp32(inplanes, planes, stride, H):
    from distributedpytorch_amd.models.resnet import Bottleneck
    from distributedpytorch_amd.parallel.flat import FlatParams
    N = 256
    torch.manual_seed(inplanes)
    net = torch.nn.Sequential(Bottleneck(inplanes, planes, stride, downsample=True),
                              Bottleneck(planes * 4, planes)).to(DEV).train()
    for mod in net.modules():                      # non-trivial BatchNorm affine (bn3's gamma is zero-initialised)
        if hasattr(mod, "running_mean"):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5); torch.nn.init.normal_(mod.bias, 0.0, 0.2)
    flat = FlatParams(net)
    flat.install_sink(None)
    x = torch.randn(N, H, H, inplanes, device=DEV).to(torch.bfloat16).requires_grad_()
    out = net(x)
    r = _smooth_weights(out.shape)
    (out.float() * r.float()).sum().backward()
    torch.cuda.synchronize()

    xr = x.detach().float().permute(0, 3, 1, 2).contiguous().requires_grad_()
    y1, l1 = _ref_block(xr, net[0])
    y2, l2 = _ref_block(y1, net[1])
    (y2 * r.float().permute(0, 3, 1, 2)).sum().backward()
    torch.cuda.synchronize()

    e_out = _rel(out.detach().permute(0, 3, 1, 2), y2.detach())
    e_dx = _rel(x.grad.permute(0, 3, 1, 2), xr.grad)
    # The input gradient of a BatchNorm network is what is left after each BatchNorm backward has removed the mean and
    # the xhat-projection of a coherent gradient: a small difference of large terms, so ANY bf16 pipeline is far from
    # fp32 on it in relative terms.  Second oracle: stock PyTorch under bf16 autocast, same two blocks, same objective.
    xa = x.detach().float().permute(0, 3, 1, 2).contiguous().requires_grad_()
    with torch.autocast("cuda", dtype=torch.bfloat16):
        a1, s1 = _ref_block(xa, net[0])
        a2, s2 = _ref_block(a1, net[1])
    (a2.float() * r.float().permute(0, 3, 1, 2)).sum().backward()
    torch.cuda.synchronize()
    e_dx_stock = _rel(xa.grad, xr.grad)
    stock = {"conv": 0.0, "bn": 0.0}
    for (_, leaf16, kind), (_, leaf32, _) in zip(s1 + s2, l1 + l2):
        stock[kind] = max(stock[kind], _rel(leaf16.grad, leaf32.grad))
    worst = {"conv": 0.0, "bn": 0.0}
    for p, leaf, kind in l1 + l2:
        gref = leaf.grad
        if kind == "conv":
            gref = gref.permute(0, 2, 3, 1)          # OIHW -> KRSC
            got = p.grad[..., : gref.shape[-1]]
        else:
            got = p.grad
        worst[kind] = max(worst[kind], _rel(got, gref))
    print(f"stage {inplanes}->{planes}: out {e_out:.4f} | dx {e_dx:.4f} (stock bf16 autocast {e_dx_stock:.4f}) | worst conv "
          f"gradient {worst['conv']:.4f} ({stock['conv']:.4f}) | worst bn gradient {worst['bn']:.4f} ({stock['bn']:.4f})")
    # observed on MI355X, ours (stock bf16): stage 1 dx 0.113 (0.118) conv 0.027 bn 0.025; stages 2-4 dx 0.117-0.121
    # (0.122-0.126) conv 0.087-0.122 bn 0.070-0.092 - everywhere at or below what stock bf16 autocast gives
    assert e_out < 1.5e-2, e_out
    assert e_dx < 1.25 * e_dx_stock + 0.01, (e_dx, e_dx_stock)
    assert worst["conv"] < 1.25 * stock["conv"] + 0.01, (worst, stock)
    assert worst["bn"] < 1.25 * stock["bn"] + 0.01, (worst, stock)


# ------------------------------------------------------------------------------------------ GPT-2 block
def test_gpt2_block_at_8x1024_matches_fp32():
    from distributedpytorch_amd.models import gpt2
    from distributedpytorch_amd.ops import _ext
    from distributedpytorch_amd.parallel.flat import FlatParams
    if not hasattr(_ext.load(), "attention_fwd"):
        pytest.skip("attention kernels not built")
    cfg = gpt2.GPT2Config()
    torch.manual_seed(7)
    blk = gpt2.Block(cfg).to(DEV).train()
    for n, p in blk.named_parameters():            # biases / LayerNorm affine away from their trivial initial values
        if p.dim() == 1:
            with torch.no_grad():
                p.add_(torch.randn_like(p) * 0.1)
    flat = FlatParams(blk)
    flat.install_sink(None)
    B, T, C, Hh = 8, 1024, cfg.n_embd, cfg.n_head
    x = torch.randn(B, T, C, device=DEV).to(torch.bfloat16).requires_grad_()
    out = blk(x)
    r = _smooth_weights(out.shape) * torch.randn(out.shape, device=DEV).to(torch.bfloat16)   # no ReLU bits here: noise is fine
    (out.float() * r.float()).sum().backward()
    torch.cuda.synchronize()

    P = {n: p.detach().float().clone().requires_grad_() for n, p in blk.named_parameters()}
    xr = x.detach().float().requires_grad_()
    h = F.layer_norm(xr, (C,), P["ln_1.weight"], P["ln_1.bias"], blk.ln_1.eps)
    qkv = F.linear(h, P["attn.c_attn.weight"], P["attn.c_attn.bias"])
    q, k, v = qkv.view(B, T, 3, Hh, C // Hh).permute(2, 0, 3, 1, 4)
    a = F.scaled_dot_product_attention(q, k, v, is_causal=True).permute(0, 2, 1, 3).reshape(B, T, C)
    x1 = xr + F.linear(a, P["attn.c_proj.weight"], P["attn.c_proj.bias"])
    h2 = F.layer_norm(x1, (C,), P["ln_2.weight"], P["ln_2.bias"], blk.ln_2.eps)
    m = F.linear(F.gelu(F.linear(h2, P["mlp.c_fc.weight"], P["mlp.c_fc.bias"]), approximate="tanh"),
                 P["mlp.c_proj.weight"], P["mlp.c_proj.bias"])
    ref = x1 + m
    (ref * r.float()).sum().backward()
    torch.cuda.synchronize()

    e_out, e_dx = _rel(out.detach(), ref.detach()), _rel(x.grad, xr.grad)
    worst = max(_rel(p.grad, P[n].grad) for n, p in blk.named_parameters())
    print(f"gpt2 block: out {e_out:.4f} dx {e_dx:.4f} worst parameter gradient {worst:.4f}")
    # observed on MI355X: out 0.0024, dx 0.0021, worst parameter gradient 0.0050
    assert e_out < 1e-2 and e_dx < 2.5e-2 and worst < 2.5e-2, (e_out, e_dx, worst)


# ------------------------------------------------------------------------------------------ 50 steps, fused vs not
def test_resnet50_fused_and_unfused_paths_reach_the_same_loss_after_50_steps():
    """docs/DESIGN.md section 1 as a test: ResNet-50, batch 256, same seed, 50 SGD steps on one synthetic batch."""
    from distributedpytorch_amd.models import resnet
    from distributedpytorch_amd.ops import block
    from distributedpytorch_amd.optim import FusedSGD
    from distributedpytorch_amd.parallel.dp import DataParallel
    g = torch.Generator(device="cpu").manual_seed(0)
    x = torch.randn(256, 3, 224, 224, generator=g).to(DEV)
    y = torch.randint(0, 1000, (256,), generator=g).to(DEV)
    final = {}
    keep = block.ENABLED
    try:
        for fused in (True, False):
            block.ENABLED = fused
            torch.manual_seed(1)
            m = resnet.resnet50().to(DEV).train()
            dp = DataParallel(m)
            opt = FusedSGD(dp.flat, lr=0.1, momentum=0.9, weight_decay=1e-4)
            for _ in range(50):
                loss = m.loss(x, y)
                loss.backward(); dp.finish(); opt.step()
            final[fused] = float(loss.detach())
            del m, dp, opt
            torch.cuda.empty_cache()
    finally:
        block.ENABLED = keep
    print(f"loss after 50 steps: fused {final[True]:.4f}, one node per op {final[False]:.4f}")
    assert math.isfinite(final[True]) and final[True] < 6.5            # it trains (ln 1000 = 6.9 at the start)
    assert abs(final[True] - final[False]) <= 0.01 * abs(final[False]) + 0.02, final


# ------------------------------------------------------------------------------------------ second oracle: stock bf16
class _StockBottleneck(torch.nn.Module):
    def __init__(self, inp, planes, stride):
        super().__init__()
        nn = torch.nn
        self.conv1, self.bn1 = nn.Conv2d(inp, planes, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv2, self.bn2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False), nn.BatchNorm2d(planes)
        self.conv3, self.bn3 = nn.Conv2d(planes, planes * 4, 1, bias=False), nn.BatchNorm2d(planes * 4)
        self.downsample = nn.Sequential(nn.Conv2d(inp, planes * 4, 1, stride, bias=False), nn.BatchNorm2d(planes * 4))

    def forward(self, x):
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        return F.relu(self.bn3(self.conv3(y)) + self.downsample(x))


class _StockTiny(torch.nn.Module):
    def __init__(self, width=16, classes=16):
        super().__init__()
        nn = torch.nn
        self.conv1, self.bn1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False), nn.BatchNorm2d(64)
        self.block1, self.block2 = _StockBottleneck(64, width, 1), _StockBottleneck(width * 4, width, 2)
        self.fc = nn.Linear(width * 4, classes)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        x = self.block2(self.block1(x))
        return self.fc(x.mean((2, 3)))


def _load_stock_from_ours(stock, ours):
    pairs = [(stock.conv1, ours.stem.conv1), (stock.bn1, ours.stem.bn1), (stock.fc, ours.head.fc)]
    for sb, ob in ((stock.block1, ours.block1), (stock.block2, ours.block2)):
        pairs += [(sb.conv1, ob.conv1), (sb.bn1, ob.bn1), (sb.conv2, ob.conv2), (sb.bn2, ob.bn2), (sb.conv3, ob.conv3),
                  (sb.bn3, ob.bn3), (sb.downsample[0], ob.downsample[0]), (sb.downsample[1], ob.downsample[1])]
    with torch.no_grad():
        for s, o in pairs:
            if isinstance(s, torch.nn.Conv2d):
                s.weight.copy_(o.weight.detach()[..., : s.in_channels].permute(0, 3, 1, 2))
            elif isinstance(s, torch.nn.Linear):
                s.weight.copy_(o.weight.detach()[: s.out_features]); s.bias.copy_(o.bias.detach()[: s.out_features])
            else:
                s.weight.copy_(o.weight.detach()); s.bias.copy_(o.bias.detach())
    return pairs


def _stock_grads(pairs):
    out = []
    for s, o in pairs:
        if isinstance(s, torch.nn.Conv2d):
            out.append(s.weight.grad.permute(0, 2, 3, 1).reshape(-1))
        else:
            out.append(s.weight.grad.reshape(-1))
            out.append(s.bias.grad.reshape(-1))
    return torch.cat([t.float() for t in out])


def _our_grads(pairs):
    out = []
    for s, o in pairs:
        if isinstance(s, torch.nn.Conv2d):
            out.append(o.weight.grad[..., : s.in_channels].reshape(-1))
        elif isinstance(s, torch.nn.Linear):
            out.append(o.weight.grad[: s.out_features].reshape(-1)); out.append(o.bias.grad[: s.out_features].reshape(-1))
        else:
            out.append(o.weight.grad.reshape(-1)); out.append(o.bias.grad.reshape(-1))
    return torch.cat([t.float() for t in out])


def test_tiny_resnet_gradient_error_is_what_stock_bf16_gives():
    from distributedpytorch_amd.models import resnet
    from distributedpytorch_amd.parallel.dp import DataParallel
    torch.manual_seed(2)
    ours = resnet.tiny_resnet().to(DEV).train()
    for mod in ours.modules():
        if hasattr(mod, "running_mean"):
            torch.nn.init.uniform_(mod.weight, 0.5, 1.5); torch.nn.init.normal_(mod.bias, 0.0, 0.2)
    stock = _StockTiny().to(DEV).train()
    pairs = _load_stock_from_ours(stock, ours)
    dp = DataParallel(ours)
    g = torch.Generator(device="cpu").manual_seed(9)
    x = torch.randn(32, 3, 64, 64, generator=g).to(DEV); y = torch.randint(0, 16, (32,), generator=g).to(DEV)

    ours.loss(x, y).backward(); dp.finish()
    g_ours = _our_grads(pairs)
    F.cross_entropy(stock(x), y).backward()
    g_fp32 = _stock_grads(pairs)
    stock.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16):
        loss16 = F.cross_entropy(stock(x).float(), y)
    loss16.backward()
    g_bf16 = _stock_grads(pairs)
    e_ours, e_stock = _rel(g_ours, g_fp32), _rel(g_bf16, g_fp32)
    print(f"distance from the fp32 gradient (L2, all parameters): ours {e_ours:.4f}, stock bf16 autocast {e_stock:.4f}")
    # observed on MI355X: ours 0.170, stock bf16 autocast 0.175 - the distance from fp32 is bf16's, not these kernels'
    assert e_ours < max(2.0 * e_stock, 0.03) + 0.02, (e_ours, e_stock)
    assert e_ours < 0.2


def _lm_head_reference(x, w, t, n_classes, rows_chunk=2048):
    """fp32 oracle of mean CE((x @ w^T)[:, :n_classes], t) with its gradients w.r.t. x and w, row-chunked so that the
    [rows, vocab] logits never exist at once either."""
    rows = x.shape[0]
    xf, wf = x.float(), w.float()[:n_classes]
    loss = torch.zeros((), device=x.device, dtype=torch.float64)
    dx = torch.empty_like(xf)
    dw = torch.zeros_like(wf)
    for r0 in range(0, rows, rows_chunk):
        lg = xf[r0:r0 + rows_chunk] @ wf.t()
        lse = torch.logsumexp(lg, dim=1)
        tt = t[r0:r0 + rows_chunk]
        loss += (lse - lg.gather(1, tt[:, None])[:, 0]).double().sum()
        p = torch.softmax(lg, dim=1)
        p[torch.arange(p.shape[0], device=x.device), tt] -= 1.0
        p /= rows
        dx[r0:r0 + rows_chunk] = p @ wf
        dw += p.t() @ xf[r0:r0 + rows_chunk]
    return float(loss / rows), dx, dw


@pytest.mark.parametrize("rows,chunk", [(4096, 1024), (3000, 1024)])
def test_chunked_lm_head_loss_matches_the_unchunked_path(rows, chunk):
    """ops/lmhead.py against ops.linear + ops.softmax_cross_entropy at GPT-2's head shape (768 -> 50,304 padded columns,
    50,257 classes), ragged last chunk included: same loss, same input gradient, same weight gradient."""
    from distributedpytorch_amd import ops
    from distributedpytorch_amd.ops import lmhead
    torch.manual_seed(3)
    V, Vp, Cin = 50257, 50304, 768
    w = torch.zeros(Vp, Cin, device=DEV)
    w[:V] = torch.randn(V, Cin, device=DEV) * 0.02
    w_lp = w.to(torch.bfloat16)
    x0 = (torch.randn(rows, Cin, device=DEV)).to(torch.bfloat16)
    t = torch.randint(0, V, (rows,), device=DEV)
    wa = w.clone().requires_grad_(True)
    xa = x0.clone().requires_grad_(True)
    la = lmhead.lm_head_loss(xa, wa, w_lp, None, t, V, chunk_rows=chunk)
    la.backward()
    wb = w.clone().requires_grad_(True)
    xb = x0.clone().requires_grad_(True)
    lb = ops.softmax_cross_entropy(ops.linear(xb, wb, w_lp), t, V, inplace_backward=True)
    lb.backward()
    assert abs(float(la) - float(lb)) < 2e-3 * abs(float(lb))
    assert _rel(xa.grad.float(), xb.grad.float()) < 0.02
    assert _rel(wa.grad, wb.grad) < 0.02
    ref_loss, ref_dx, ref_dw = _lm_head_reference(x0, w_lp, t, V)
    assert abs(float(la) - ref_loss) < 5e-3 * abs(ref_loss)
    assert _rel(xa.grad.float(), ref_dx) < 0.03 and _rel(wa.grad[:V], ref_dw) < 0.03


def test_lm_head_loss_at_32x1024_tokens_beyond_the_one_operand_limit():
    """32 x 1024 x 50,304 logits are 1.65 G elements - past the 2^30-element range of one GEMM operand, which capped
    GPT-2 at batch 20.  Row-chunked (8 x 4,096 rows here) the head + loss run, and agree with a chunked fp32 oracle:
    the loss over all rows, the input gradient on sampled rows, the weight gradient in aggregate."""
    from distributedpytorch_amd.ops import lmhead
    torch.manual_seed(5)
    rows, V, Vp, Cin = 32 * 1024, 50257, 50304, 768
    assert lmhead.chunk_rows_for(rows, Vp) > 0 and lmhead.chunk_rows_for(16 * 1024, Vp) == lmhead._FORCE_ROWS
    w = torch.zeros(Vp, Cin, device=DEV)
    w[:V] = torch.randn(V, Cin, device=DEV) * 0.02
    w_lp = w.to(torch.bfloat16)
    x0 = torch.randn(rows, Cin, device=DEV).to(torch.bfloat16)
    t = torch.randint(0, V, (rows,), device=DEV)
    wa = w.clone().requires_grad_(True)
    xa = x0.clone().requires_grad_(True)
    loss = lmhead.lm_head_loss(xa, wa, w_lp, None, t, V, chunk_rows=4096)
    loss.backward()
    torch.cuda.synchronize()
    ref_loss, ref_dx, ref_dw = _lm_head_reference(x0, w_lp, t, V)
    assert abs(float(loss) - ref_loss) < 5e-3 * abs(ref_loss), (float(loss), ref_loss)
    sample = torch.randint(0, rows, (256,), device=DEV)
    assert _rel(xa.grad.float()[sample], ref_dx[sample]) < 0.03
    assert _rel(wa.grad[:V], ref_dw) < 0.03
    assert float(wa.grad[V:].abs().max()) == 0.0            # padding rows of the vocabulary take no gradient
