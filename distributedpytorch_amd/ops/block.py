"""Whole-bottleneck forward/backward as ONE autograd node with a hand-scheduled backward.

A ResNet bottleneck is 3-4 convs + BNs with a residual join.  Expressed op by op,
autograd has to SUM the two gradients that reach the block input (main branch dgrad +
residual branch) with a separate elementwise kernel over the largest tensors of the
network.  Scheduling the block's backward by hand lets the residual-branch gradient ride
the ``addend`` epilogue of the conv1 (or downsample) dgrad kernel instead - one read
fused into a GEMM epilogue replaces a 3-pass elementwise kernel - and removes ~20
Python autograd nodes per block from the launch path.  BN statistics still come out of
the conv epilogues; BN parameter gradients and filter gradients are accumulated straight
into the flat fp32 gradient buffer when a sink is installed.

GPU only (the CPU oracle path composes the plain ops).
"""
from __future__ import annotations

import torch

import os

from . import _ext
from . import side as _side
from . import sink as _sink
from . import wgroup as _wgroup

ENABLED = os.environ.get("DPA_NO_FUSED_BLOCK", "0") != "1"
# Optional: conv3 + bn3 RECOMPUTED instead of stored when conv3's output is at least DPA_RECOMPUTE_MB big.
# The 1x1 conv is cheap on the matrix cores and its output is the largest tensor of the block, so forward can
# run it twice (statistics pass, then BN + residual + ReLU in the conv epilogue) and backward twice more
# (reduction pass, apply pass) without ever writing the raw conv output: -3.25 tensor passes and -1 stored
# tensor per block.  Measured on MI355X, batch 256 (scripts/bench_recompute.py): layer-1 forward 394 -> 323 us,
# but backward 422 -> 508 us - the conv kernel's epilogue streams at ~3 TB/s where the dedicated BatchNorm
# kernels reach ~5 TB/s - so it is a MEMORY-saving option (off by default), not a speed-up, until the epilogue
# is as good a streamer as norm.hip.
RECOMPUTE_MIN_BYTES = int(float(os.environ.get("DPA_RECOMPUTE_MB", "1e12")) * 2 ** 20)
_DS_SIDE = os.environ.get("DPA_DS_SIDE", "1") != "0"
# weight gradients of the late stages' 1 x 1 layers through the wide multi-layer kernel (A/B in profiles/r3/wgrad_wide_resnet_ab.txt)
_WIDE_1X1 = os.environ.get("DPA_WGRAD_WIDE_1X1", "1") == "1"
_FOLD_DS_APPLY = os.environ.get("DPA_NO_FOLD_DS_APPLY", "0") != "1"


def _conv_bn_fwd(C, conv, bn, x, relu, residual=None, res_affine=None):
    y, stats = C.conv_fprop(x, conv.weight_lp, None, conv.stride, conv.padding, 0, True, bn.stats_buffer(x.device))
    P = y.numel() // y.shape[-1]
    mean, invstd, scale, shift = C.bn_finalize(stats, P, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                               bn.running_var, bn.momentum, bn.eps, True)
    if res_affine is not None:
        a, mask = C.bn_apply(y, residual, scale, shift, relu, relu, res_affine[0], res_affine[1])
    else:
        a, mask = C.bn_apply(y, residual, scale, shift, relu, relu)
    return y, a, mean, invstd, mask


def _conv_bn_stats_only(C, conv, bn, x):
    """Conv + the BatchNorm's statistics, WITHOUT its apply pass: for the projection shortcut, whose normalised
    output is only ever added to bn3's - bn3's apply kernel applies (scale, shift) to the raw conv output as it
    reads it, so the shortcut's own read-write pass over the block's widest tensor is gone."""
    y, stats = C.conv_fprop(x, conv.weight_lp, None, conv.stride, conv.padding, 0, True, bn.stats_buffer(x.device))
    P = y.numel() // y.shape[-1]
    mean, invstd, scale, shift = C.bn_finalize(stats, P, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                               bn.running_var, bn.momentum, bn.eps, True)
    return y, mean, invstd, scale, shift


def _bn_bwd(C, bn, dy, mask, x_raw, mean, invstd, relu, grads, slot):
    tg, tb = _sink.target(bn.weight), _sink.target(bn.bias)
    if tg is not None and tb is not None:
        dx, _, _ = C.bn_backward(dy, mask, x_raw, bn.weight.detach(), mean, invstd, relu, tg, tb, bn.stats_buffer_bwd(dy.device))
        _sink.ready(bn.weight); _sink.ready(bn.bias)
    else:
        dx, dg, db = C.bn_backward(dy, mask, x_raw, bn.weight.detach(), mean, invstd, relu, None, None,
                                   bn.stats_buffer_bwd(dy.device))
        grads[slot], grads[slot + 1] = dg, db
    return dx


_DGRAD_BN_STATS = os.environ.get("DPA_NO_DGRAD_BN_STATS", "0") != "1"
_CROSS_BN3 = _DGRAD_BN_STATS and os.environ.get("DPA_NO_CROSS_BN3", "0") != "1"
_TAIL_BWD = os.environ.get("DPA_NO_TAIL_BWD", "0") != "1"


def _bn_coefs(C, bn, dy, mask, x_raw, mean, invstd, grads, slot):
    """Reduction pass + finalize of a BatchNorm(+ReLU) backward WITHOUT the apply pass: co [5, C] (rows 2..4 = cA, cB, cC);
    the parameter gradients go to the sink or into ``grads``."""
    tg, tb = _sink.target(bn.weight), _sink.target(bn.bias)
    sunk = tg is not None and tb is not None
    co = C.bn_backward_coefs(dy, mask, x_raw, bn.weight.detach(), mean, invstd, True, tg if sunk else None,
                             tb if sunk else None, bn.stats_buffer_bwd(dy.device))
    if sunk:
        _sink.ready(bn.weight); _sink.ready(bn.bias)
    else:
        grads[slot], grads[slot + 1] = co[0], co[1]
    return co


def _coefs_from_stats(C, bn, sb, P, mean, invstd, grads, slot):
    """Finalize of a BatchNorm backward whose reduction pass was accumulated into ``sb`` by a conv epilogue."""
    tg, tb = _sink.target(bn.weight), _sink.target(bn.bias)
    sunk = tg is not None and tb is not None
    co = C.bn_bwd_coefs(sb, P, bn.weight.detach(), mean, invstd, tg if sunk else None, tb if sunk else None)
    if sunk:
        _sink.ready(bn.weight); _sink.ready(bn.bias)
    else:
        grads[slot], grads[slot + 1] = co[0], co[1]
    return co


def _tail_ok(C, conv, x_in, dout):
    """The one-kernel backward of ``1x1 conv -> BatchNorm -> mask`` (csrc/tail_bwd.hip) covers this layer."""
    return (_TAIL_BWD and conv.stride == 1 and conv.padding == 0 and conv.weight_lp.shape[1] == 1
            and x_in.shape[:3] == dout.shape[:3]
            and C.tail_bwd_ok(dout.numel() // dout.shape[-1], x_in.shape[-1], dout.shape[-1]))


def _tail_bwd(C, conv, co, dout, kout, craw, x_in, grads, slot, bn_x=None, bn_mask=None, bn_stats=None):
    """g = BatchNorm-backward apply(dout under kout, craw; co), never stored; conv.weight's gradient += g^T x_in (sink or
    ``grads[slot]``); returns g @ w, with the reduction pass of the BatchNorm behind it added into ``bn_stats``."""
    tgt = _sink.target(conv.weight)
    dw = tgt if tgt is not None else torch.zeros(conv.weight.shape, dtype=torch.float32, device=dout.device)
    dx = C.tail_bwd(dout, kout, craw, co, x_in, _wt(C, conv), dw, bn_x, bn_mask, bn_stats)
    if tgt is not None:
        _sink.ready(conv.weight)
    else:
        grads[slot] = dw
    return dx


def _dgrad_then_bn_bwd(C, conv, bn, dy, hw, mask, x_raw, mean, invstd, grads, slot):
    """``da = dgrad(conv, dy)`` followed by the backward of the BatchNorm (+ReLU) whose output ``da`` is the gradient
    of.  The dgrad kernel's epilogue also produces that BatchNorm's reduction pass (sum g', sum g' * x
    from the tile it is about to write), so the stand-alone statistics kernel - one full read of ``da`` and of the
    BatchNorm input - is skipped: finalize + apply only."""
    tg, tb = _sink.target(bn.weight), _sink.target(bn.bias)
    sunk = tg is not None and tb is not None
    if _DGRAD_BN_STATS:
        sb = bn.stats_buffer_bwd(dy.device)
        da = C.conv_dgrad(dy, _wt(C, conv), hw[0], hw[1], conv.stride, conv.padding, None, None, x_raw, mask, sb)
        co = C.bn_bwd_coefs(sb, da.numel() // da.shape[-1], bn.weight.detach(), mean, invstd,
                            tg if sunk else None, tb if sunk else None)
        if sunk:
            _sink.ready(bn.weight); _sink.ready(bn.bias)
        else:
            grads[slot], grads[slot + 1] = co[0], co[1]
        return C.bn_bwd_apply(da, mask, x_raw, co, True)
    da = C.conv_dgrad(dy, _wt(C, conv), hw[0], hw[1], conv.stride, conv.padding, None, None)
    return _bn_bwd(C, bn, da, mask, x_raw, mean, invstd, True, grads, slot)


def _wt(C, conv):
    """Filter with K and C swapped for dgrad: the twin FlatParams keeps current, else transposed on the spot."""
    wt = getattr(conv, "weight_lp_t", None)
    return wt if wt is not None else C.transpose_filter(conv.weight_lp)


def need_dx_early(ctx) -> bool:
    return bool(ctx.needs_input_grad[0])


def _wgrad(C, conv, x, dy, grads, slot, in_line=False):
    tgt = _sink.target(conv.weight)
    if tgt is not None:
        if (_WIDE_1X1 and not in_line and _wgroup.ENABLED and _side.active(x) and conv.weight.shape[1] == 1 and conv.weight.shape[2] == 1
                and conv.stride == 1 and conv.padding == 0):
            # the 1 x 1 layers of the late stages (filters of at least 192 x 192): several blocks' weight gradients in
            # ONE launch of the wide kernel (ops/wgroup.py, csrc/wgrad_wide.hip)
            rows, cin, cout = x.numel() // x.shape[-1], x.shape[-1], dy.shape[-1]
            ok, tiles = _wgroup.query(rows, cout, cin)
            if ok and (x.data_ptr() | dy.data_ptr() | tgt.data_ptr()) % 16 == 0:
                _wgroup.for_device(x.device.index).add(x.view(rows, cin), dy.view(rows, cout), tgt.view(cout, cin), None,
                                                       [conv.weight], tiles)
                return
        if _side.active(x) and not in_line:   # off the critical path: overlaps the BN-backward / dgrad chain (ops/side.py)
            _side.run(lambda: C.conv_wgrad(x, dy, tgt, conv.stride, conv.padding), (x, dy))
        else:
            C.conv_wgrad(x, dy, tgt, conv.stride, conv.padding)
        _sink.ready(conv.weight)
    else:
        dw = torch.zeros(conv.weight.shape, dtype=torch.float32, device=x.device)
        C.conv_wgrad(x, dy, dw, conv.stride, conv.padding)
        grads[slot] = dw


class _Link:
    """What ONE forward call of a bottleneck leaves for the backward of the block that consumes its output (and what
    that backward leaves for this block's): per call, not per module - under a pipeline schedule several micro-batches
    are in flight through the same module."""
    __slots__ = ("blk", "c3", "k3", "stats_for")

    def __init__(self, blk):
        self.blk, self.c3, self.k3, self.stats_for = blk, None, None, None


class _BottleneckFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, blk, prev, link, *params):
        C = _ext.load()
        ds = blk.downsample is not None
        ctx.prev, ctx.link = prev, link  # links of the block whose output x is (or None) and of this call (_CROSS_BN3)
        ds_done = None
        conv3, bn3 = blk.conv3, blk.bn3
        s2 = blk.conv2.stride
        out_bytes = x.shape[0] * -(-x.shape[1] // s2) * -(-x.shape[2] // s2) * conv3.weight_lp.shape[0] * 2
        recompute = out_bytes >= RECOMPUTE_MIN_BYTES and conv3.stride == 1 and conv3.padding == 0
        fold_ds = ds and _FOLD_DS_APPLY and not recompute
        res_affine = None

        def shortcut():
            if fold_ds:
                return _conv_bn_stats_only(C, blk.downsample[0], blk.downsample[1], x)
            cd, idn, md, isd, _ = _conv_bn_fwd(C, blk.downsample[0], blk.downsample[1], x, False)
            return cd, md, isd, idn, None

        if ds and _DS_SIDE and _side.active(x):
            # the projection shortcut is independent of conv1 -> conv2: it runs on the side stream beside them
            box = []
            ds_done = _side.run(lambda: box.append(shortcut()), (x,))
            cd, md, isd, r0, r1 = box[0]
        elif ds:
            cd, md, isd, r0, r1 = shortcut()
        else:
            cd = md = isd = None
            r0, r1 = x, None
        c1, a1, m1, i1, k1 = _conv_bn_fwd(C, blk.conv1, blk.bn1, x, True)
        c2, a2, m2, i2, k2 = _conv_bn_fwd(C, blk.conv2, blk.bn2, a1, True)
        if ds_done is not None:
            _side.wait(ds_done, (cd, md, isd, r0) + ((r1,) if r1 is not None else ()))
        if fold_ds:
            idn, res_affine = cd, (r0, r1)
        else:
            idn = r0
        if recompute:
            sb = bn3.stats_buffer(x.device)
            C.conv_fprop_stats(a2, conv3.weight_lp, 1, 0, sb)
            P = a2.numel() // a2.shape[-1]
            m3, i3, scale, shift = C.bn_finalize(sb, P, bn3.weight.detach(), bn3.bias.detach(), bn3.running_mean,
                                                 bn3.running_var, bn3.momentum, bn3.eps, True)
            out, k3 = C.conv_bn_res_relu(a2, conv3.weight_lp, 1, 0, scale, shift, idn)
            c3 = a2.new_empty(0)                       # placeholder: the raw conv output does not exist
        else:
            c3, out, m3, i3, k3 = _conv_bn_fwd(C, conv3, bn3, a2, True, idn, res_affine)
        ctx.recompute = recompute
        # the ReLU masks travel as 1 bit per element: the backward pass never re-reads an activation for its sign
        saved = [x, c1, a1, m1, i1, k1, c2, a2, m2, i2, k2, c3, m3, i3, k3]
        if ds:
            saved += [cd, md, isd]
        ctx.save_for_backward(*saved)
        ctx.blk = blk
        # what the NEXT block's backward needs to fold this block's bn3 reduction pass into its last dgrad
        if not recompute:
            link.c3, link.k3 = c3, k3
        return out

    @staticmethod
    def backward(ctx, dout):
        C = _ext.load()
        blk = ctx.blk
        ds = blk.downsample is not None
        sv = ctx.saved_tensors
        x, c1, a1, m1, i1, k1, c2, a2, m2, i2, k2, c3, m3, i3, k3 = sv[:15]
        dout = dout.contiguous()
        n_params = 12 if ds else 9
        grads = [None] * n_params          # order: conv1.w bn1.w bn1.b conv2.w bn2.w bn2.b conv3.w bn3.w bn3.b [ds.w dsbn.w dsbn.b]
        # bn3 behind the residual join: the masked dout is both bn3's incoming gradient and the gradient of the
        # shortcut branch; it is never materialised (consumers apply the bit mask k3 themselves)
        ds_done, ds_box = None, []
        if (ds and need_dx_early(ctx) and _DS_SIDE and _side.active(dout)
                and _sink.target(blk.downsample[0].weight) is not None and _sink.target(blk.downsample[1].weight) is not None):
            # the projection shortcut's backward (BN backward, wgrad, dgrad) depends on dout only: the whole branch
            # runs on the side stream beside the main branch and meets it again in the last dgrad's epilogue
            cd_, md_, isd_ = sv[15:18]
            dconv_ = blk.downsample[0]

            def ds_branch():
                if _tail_ok(C, dconv_, x, dout):
                    co_ = _bn_coefs(C, blk.downsample[1], dout, k3, cd_, md_, isd_, grads, 10)
                    ds_box.append(_tail_bwd(C, dconv_, co_, dout, k3, cd_, x, grads, 9))
                    return
                dd_ = _bn_bwd(C, blk.downsample[1], dout, k3, cd_, md_, isd_, True, grads, 10)
                _wgrad(C, dconv_, x, dd_, grads, 9, in_line=True)
                ds_box.append(C.conv_dgrad(dd_, _wt(C, dconv_), x.shape[1], x.shape[2], dconv_.stride, 0, None, None))
                ds_box.append(dd_)
            ds_done = _side.run(ds_branch, (dout, k3, cd_, md_, isd_, x))
        if ctx.recompute:
            bn3, w3 = blk.bn3, blk.conv3.weight_lp
            sb = bn3.stats_buffer_bwd(dout.device)
            C.conv_bnb_stats(a2, w3, 1, 0, dout, k3, sb)
            tg, tb = _sink.target(bn3.weight), _sink.target(bn3.bias)
            sunk = tg is not None and tb is not None
            co = C.bn_bwd_coefs(sb, dout.numel() // dout.shape[-1], bn3.weight.detach(), m3, i3,
                                tg if sunk else None, tb if sunk else None)
            if sunk:
                _sink.ready(bn3.weight); _sink.ready(bn3.bias)
            else:
                grads[7], grads[8] = co[0], co[1]
            d3 = C.conv_bnb_apply(a2, w3, 1, 0, dout, k3, co[2], co[3], co[4])
        else:
            co3 = None
            if ctx.link.stats_for is not None:
                # the block after this one already accumulated bn3's reduction pass while writing dout (_CROSS_BN3)
                sb = blk.bn3.stats_buffer_bwd(dout.device)
                if ctx.link.stats_for == (dout.data_ptr(), tuple(dout.shape)):
                    co3 = _coefs_from_stats(C, blk.bn3, sb, dout.numel() // dout.shape[-1], m3, i3, grads, 7)
                else:                   # dout is not the tensor those statistics were taken from (another consumer
                    sb.zero_()          # of the block output added its gradient): discard them
                ctx.link.stats_for = None
            if co3 is None:
                co3 = _bn_coefs(C, blk.bn3, dout, k3, c3, m3, i3, grads, 7)
            if _tail_ok(C, blk.conv3, a2, dout):
                # stage 1: bn3's apply pass, conv3's wgrad and dgrad and bn2's reduction pass in ONE kernel - the
                # gradient of the block's widest tensor is read once and never written (csrc/tail_bwd.hip)
                d3 = None
                if _DGRAD_BN_STATS:
                    sb2 = blk.bn2.stats_buffer_bwd(dout.device)
                    da2 = _tail_bwd(C, blk.conv3, co3, dout, k3, c3, a2, grads, 6, c2, k2, sb2)
                    co2 = _coefs_from_stats(C, blk.bn2, sb2, da2.numel() // da2.shape[-1], m2, i2, grads, 4)
                    d2 = C.bn_bwd_apply(da2, k2, c2, co2, True)
                else:
                    da2 = _tail_bwd(C, blk.conv3, co3, dout, k3, c3, a2, grads, 6)
                    d2 = _bn_bwd(C, blk.bn2, da2, k2, c2, m2, i2, True, grads, 4)
            else:
                d3 = C.bn_bwd_apply(dout, k3, c3, co3, True)
        ctx.link.c3 = ctx.link.k3 = None
        # each wgrad goes to the side stream BEFORE the dgrad that shares its operand is enqueued: the two GEMMs
        # run side by side and the wgrad's tail overlaps the next BatchNorm backward (measured better than
        # enqueueing it after the dgrad, and better than keeping the HBM-bound stage-1/2 1x1 wgrads in line)
        if d3 is not None:
            _wgrad(C, blk.conv3, a2, d3, grads, 6)
            d2 = _dgrad_then_bn_bwd(C, blk.conv3, blk.bn2, d3, a2.shape[1:3], k2, c2, m2, i2, grads, 4)
        _wgrad(C, blk.conv2, a1, d2, grads, 3)
        d1 = _dgrad_then_bn_bwd(C, blk.conv2, blk.bn1, d2, a1.shape[1:3], k1, c1, m1, i1, grads, 1)
        _wgrad(C, blk.conv1, x, d1, grads, 0)
        dx = None
        need_dx = ctx.needs_input_grad[0]
        # bn3 of the PREVIOUS block: its incoming gradient is the dx this block is about to write, so the reduction
        # pass of its backward (sum g', sum g' * c3 under its ReLU bits) rides this dgrad's epilogue
        prev, bn_args = ctx.prev, (None, None, None)
        if _CROSS_BN3 and need_dx and prev is not None and prev.c3 is not None and prev.c3.shape == x.shape:
            bn_args = (prev.c3, prev.k3, prev.blk.bn3.stats_buffer_bwd(dout.device))
        if ds_done is not None:
            _side.wait(ds_done, ds_box)
            dx = C.conv_dgrad(d1, _wt(C, blk.conv1), x.shape[1], x.shape[2], 1, 0, ds_box[0], None, *bn_args)
        elif ds:
            cd, md, isd = sv[15:18]
            dconv = blk.downsample[0]
            if need_dx and _tail_ok(C, dconv, x, dout):
                cod = _bn_coefs(C, blk.downsample[1], dout, k3, cd, md, isd, grads, 10)
                dxd = _tail_bwd(C, dconv, cod, dout, k3, cd, x, grads, 9)
            else:
                dd = _bn_bwd(C, blk.downsample[1], dout, k3, cd, md, isd, True, grads, 10)   # "relu" = the join's mask
                _wgrad(C, dconv, x, dd, grads, 9)
                dxd = C.conv_dgrad(dd, _wt(C, dconv), x.shape[1], x.shape[2], dconv.stride, 0, None, None) if need_dx else None
            if need_dx:
                dx = C.conv_dgrad(d1, _wt(C, blk.conv1), x.shape[1], x.shape[2], 1, 0, dxd, None, *bn_args)
        elif need_dx:
            # identity shortcut: the masked residual gradient is added in the dgrad epilogue
            dx = C.conv_dgrad(d1, _wt(C, blk.conv1), x.shape[1], x.shape[2], 1, 0, dout, k3, *bn_args)
        if bn_args[0] is not None and dx is not None:
            prev.stats_for = (dx.data_ptr(), tuple(dx.shape))
        return (dx, None, None, None, *grads)


def bottleneck_params(blk):
    ps = [blk.conv1.weight, blk.bn1.weight, blk.bn1.bias, blk.conv2.weight, blk.bn2.weight, blk.bn2.bias,
          blk.conv3.weight, blk.bn3.weight, blk.bn3.bias]
    if blk.downsample is not None:
        ps += [blk.downsample[0].weight, blk.downsample[1].weight, blk.downsample[1].bias]
    return ps


def can_fuse(blk, x: torch.Tensor) -> bool:
    return (ENABLED and x.is_cuda and blk.training and torch.is_grad_enabled() and blk.conv1.weight_lp is not None
            and blk.conv2.weight_lp is not None and blk.conv3.weight_lp is not None
            and (blk.downsample is None or blk.downsample[0].weight_lp is not None))


def bottleneck(blk, x: torch.Tensor) -> torch.Tensor:
    """Fused bottleneck block (GPU, training, bf16 shadows present).  Consecutive blocks are linked through a tensor
    attribute so that a block's backward can serve the bn3 of the block before it (see _BottleneckFn.backward)."""
    link = _Link(blk)
    out = _BottleneckFn.apply(x, blk, getattr(x, "_dpa_producer", None), link, *bottleneck_params(blk))
    out._dpa_producer = link
    return out


# ---------------------------------------------------------------------------------------------- stem
STEM_FUSED = os.environ.get("DPA_NO_FUSED_STEM", "0") != "1"


class _StemFn(torch.autograd.Function):
    """conv7x7/2 -> BN -> ReLU -> maxpool3x3/2 as one node.  The post-BN activation (the largest tensor of the
    network) and its gradient are never materialised: forward pools straight from the raw conv output
    (``bn_relu_maxpool_fwd``), backward rebuilds each position's gradient from the pooled gradient and the
    winning-tap bytes inside the BatchNorm-backward kernels (``pool_bn_backward``).  Replaces, per step at batch
    256, 0.85 GB of forward and 1.1 GB of backward HBM traffic of the op-by-op path."""

    @staticmethod
    def forward(ctx, x, stem, *params):
        C = _ext.load()
        conv, bn = stem.conv1, stem.bn1
        c1, stats = C.conv_fprop(x, conv.weight_lp, None, conv.stride, conv.padding, 0, True, bn.stats_buffer(x.device))
        P = c1.numel() // c1.shape[-1]
        mean, invstd, scale, shift = C.bn_finalize(stats, P, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                                   bn.running_var, bn.momentum, bn.eps, True)
        y, idx = C.bn_relu_maxpool_fwd(c1, scale, shift)
        ctx.save_for_backward(x, c1, mean, invstd, idx)
        ctx.stem = stem
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.load()
        x, c1, mean, invstd, idx = ctx.saved_tensors
        conv, bn = ctx.stem.conv1, ctx.stem.bn1
        grads = [None, None, None]          # conv1.weight bn1.weight bn1.bias
        tg, tb = _sink.target(bn.weight), _sink.target(bn.bias)
        sunk = tg is not None and tb is not None
        dc1, dg, db = C.pool_bn_backward(dy.contiguous(), idx, c1, bn.weight.detach(), mean, invstd,
                                         tg if sunk else None, tb if sunk else None, bn.stats_buffer_bwd(dy.device))
        if sunk:
            _sink.ready(bn.weight); _sink.ready(bn.bias)
        else:
            grads[1], grads[2] = dg, db
        _wgrad(C, conv, x, dc1, grads, 0)
        return (None, None, *grads)         # the image batch needs no gradient


class _StemPackedFn(torch.autograd.Function):
    """``_StemFn`` on the PAIR-PACKED image (csrc/elementwise.hip): the 3-channel image travels as [N, H, W/2, 8] pixel
    pairs and the 7x7 / stride 2 convolution runs as 7 x 4 taps with stride (2, 1), padding (3, 2) - 224 instead of 392
    GEMM-K for the forward conv and for its weight gradient (the last kernel of the step, with nothing beside it)."""

    @staticmethod
    def forward(ctx, xp, stem, *params):
        C = _ext.load()
        conv, bn = stem.conv1, stem.bn1
        wp = C.stem_pack_filter(conv.weight_lp, 3)
        c1, stats = C.conv_fprop(xp, wp, None, 2, 3, 0, True, bn.stats_buffer(xp.device), None, False, 1, 2, xp.shape[2])
        P = c1.numel() // c1.shape[-1]
        mean, invstd, scale, shift = C.bn_finalize(stats, P, bn.weight.detach(), bn.bias.detach(), bn.running_mean,
                                                   bn.running_var, bn.momentum, bn.eps, True)
        y, idx = C.bn_relu_maxpool_fwd(c1, scale, shift)
        ctx.save_for_backward(xp, c1, mean, invstd, idx)
        ctx.stem = stem
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.load()
        xp, c1, mean, invstd, idx = ctx.saved_tensors
        conv, bn = ctx.stem.conv1, ctx.stem.bn1
        grads = [None, None, None]          # conv1.weight bn1.weight bn1.bias
        tg, tb = _sink.target(bn.weight), _sink.target(bn.bias)
        sunk = tg is not None and tb is not None
        dc1, dg, db = C.pool_bn_backward(dy.contiguous(), idx, c1, bn.weight.detach(), mean, invstd,
                                         tg if sunk else None, tb if sunk else None, bn.stats_buffer_bwd(dy.device))
        if sunk:
            _sink.ready(bn.weight); _sink.ready(bn.bias)
        else:
            grads[1], grads[2] = dg, db
        K, R = conv.weight_lp.shape[0], conv.weight_lp.shape[1]
        tgt = _sink.target(conv.weight)
        dst = tgt if tgt is not None else torch.zeros(conv.weight_lp.shape, dtype=torch.float32, device=dy.device)

        def wgrad():
            dwp = torch.zeros(K, R, 4, 8, dtype=torch.float32, device=dy.device)
            C.conv_wgrad(xp, dc1, dwp, 2, 3, None, 1, 2)
            C.stem_unpack_dw(dwp, dst.view(K, R, 7, 8), 3)

        if tgt is not None and _side.active(dy):
            _side.run(wgrad, (xp, dc1))
        else:
            wgrad()
        if tgt is not None:
            _sink.ready(conv.weight)
        else:
            grads[0] = dst
        return (None, None, *grads)         # the image batch needs no gradient


_STEM_PACKED = os.environ.get("DPA_NO_PACKED_STEM", "0") != "1"


def can_fuse_stem_packed(stem, x: torch.Tensor) -> bool:
    """``x``: the NCHW image batch.  3 channels, even width, the standard 7x7 / 2 / 3 stem convolution."""
    conv = stem.conv1
    return (_STEM_PACKED and STEM_FUSED and ENABLED and x.is_cuda and stem.training and torch.is_grad_enabled()
            and conv.weight_lp is not None and not x.requires_grad and x.dim() == 4 and x.shape[1] == 3
            and x.shape[3] % 2 == 0 and tuple(conv.weight_lp.shape[1:]) == (7, 7, 8) and conv.stride == 2 and conv.padding == 3)


def stem_packed(stem_mod, x: torch.Tensor) -> torch.Tensor:
    xp = _ext.load().nchw_to_pairs(x.contiguous())
    return _StemPackedFn.apply(xp, stem_mod, stem_mod.conv1.weight, stem_mod.bn1.weight, stem_mod.bn1.bias)


def can_fuse_stem(stem, x: torch.Tensor) -> bool:
    return (STEM_FUSED and ENABLED and x.is_cuda and stem.training and torch.is_grad_enabled()
            and stem.conv1.weight_lp is not None and not x.requires_grad)


def stem(stem_mod, x: torch.Tensor) -> torch.Tensor:
    return _StemFn.apply(x, stem_mod, stem_mod.conv1.weight, stem_mod.bn1.weight, stem_mod.bn1.bias)
