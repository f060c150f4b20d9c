"""BatchNorm (NHWC, fused with ReLU and the residual add) and LayerNorm.

GPU path: the per-channel statistics normally arrive from the producing
convolution's epilogue (``conv2d_nhwc(..., want_stats=True)``), a tiny finalize
kernel turns them into per-channel scale/shift, and ONE streaming kernel applies
``relu(x*scale + shift + residual)``.  Backward is two streaming passes (reduce,
apply) with the ReLU mask and residual gradient folded in.
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn.functional as F

from . import _ext
from . import side as _side
from . import sink as _sink


def _grad_out(param, value):
    """Route a small fp32 parameter gradient: into the sink when present, else return it."""
    tgt = _sink.target(param)
    if tgt is not None:
        tgt.add_(value.view(tgt.shape))
        _sink.ready(param)
        return None
    return value


class _BNTrainFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, stats, gamma, beta, running_mean, running_var, residual, relu, momentum, eps, reset_stats):
        C = _ext.load()
        P = x.numel() // x.shape[-1]
        if stats is None:
            stats = C.bn_stats(x)
            reset_stats = False
        mean, invstd, scale, shift = C.bn_finalize(stats, P, gamma.detach(), beta.detach(), running_mean, running_var,
                                                   momentum, eps, reset_stats)
        y, mask = C.bn_apply(x, residual, scale, shift, relu, relu)   # relu: 1 bit / element for the backward pass
        if relu:
            ctx.save_for_backward(x, mean, invstd, mask)
        else:
            ctx.save_for_backward(x, mean, invstd)
        ctx.gamma, ctx.beta = gamma, beta
        ctx.relu, ctx.has_res = relu, residual is not None
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.load()
        if ctx.relu:
            x, mean, invstd, mask = ctx.saved_tensors
        else:
            (x, mean, invstd), mask = ctx.saved_tensors, None
        dy = dy.contiguous()
        want_res = ctx.has_res and ctx.needs_input_grad[6]
        tg, tb = _sink.target(ctx.gamma), _sink.target(ctx.beta)
        if tg is not None and tb is not None:       # finalize kernel adds straight into the flat gradient buffer
            dx, _, _ = C.bn_backward(dy, mask, x, ctx.gamma.detach(), mean, invstd, ctx.relu, tg, tb, None)
            _sink.ready(ctx.gamma); _sink.ready(ctx.beta)
            dg = db = None
        else:
            dx, dgamma, dbeta = C.bn_backward(dy, mask, x, ctx.gamma.detach(), mean, invstd, ctx.relu, None, None, None)
            dg = dgamma if ctx.gamma.requires_grad else None
            db = dbeta if ctx.beta.requires_grad else None
        dres = None
        if want_res:                                # gradient of the residual branch: dy under the ReLU mask
            dres = C.relu_mask_mul(dy, mask) if ctx.relu else dy
        return dx, None, dg, db, None, None, dres, None, None, None, None


def batch_norm_nhwc(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor,
                    running_mean: Optional[torch.Tensor], running_var: Optional[torch.Tensor],
                    training: bool = True, momentum: float = 0.1, eps: float = 1e-5, relu: bool = False,
                    residual: Optional[torch.Tensor] = None, stats: Optional[torch.Tensor] = None,
                    reset_stats: bool = False) -> torch.Tensor:
    """``relu?(BN(x) + residual?)`` over the last (channel) axis of an NHWC tensor."""
    if _ext.use_hip(x):
        C = _ext.load()
        if training:
            return _BNTrainFn.apply(x, stats, gamma, beta, running_mean, running_var, residual, relu, momentum, eps,
                                    reset_stats)
        # inference: fold running stats into scale/shift (cold path), apply with the same streaming kernel
        scale = (gamma.detach() * torch.rsqrt(running_var + eps)).float().contiguous()
        shift = (beta.detach() - running_mean * scale).float().contiguous()
        return C.bn_apply(x, residual, scale, shift, relu, False)[0]
    xc = x.permute(0, 3, 1, 2) if x.dim() == 4 else x
    y = F.batch_norm(xc.float(), running_mean, running_var, gamma, beta, training, momentum, eps).to(x.dtype)
    if x.dim() == 4:
        y = y.permute(0, 2, 3, 1)
    if residual is not None:
        y = y + residual
    if relu:
        y = F.relu(y)
    return y.contiguous()


_LN_FOLD_SIDE = os.environ.get("DPA_LN_FOLD_SIDE", "1") != "0"


def _ln_bwd_into(C, dy, x, gamma, mean, rstd, tg, tb, addend):
    """LayerNorm backward whose parameter gradients go straight into the flat gradient buffer.  The kernel leaves
    per-block partial column sums; folding them (a 25 us latency-bound kernel, 25 of them per GPT-2 step) is work
    nobody waits for before the bucket all-reduce, so it runs on the side stream (ops/side.py)."""
    if _LN_FOLD_SIDE and _side.active(x):
        dx, part = C.layernorm_bwd(dy, x, gamma, mean, rstd, None, None, addend, True)
        _side.run(lambda: C.ln_fold_partials(part, tg, tb), (part,))
        return dx
    return C.layernorm_bwd(dy, x, gamma, mean, rstd, tg, tb, addend)[0]


class _LayerNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        C = _ext.load()
        y, mean, rstd = C.layernorm_fwd(x, gamma.detach(), beta.detach(), eps)
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma, ctx.beta = gamma, beta
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.load()
        x, mean, rstd = ctx.saved_tensors
        tg, tb = _sink.target(ctx.gamma), _sink.target(ctx.beta)
        if tg is not None and tb is not None:       # the fold kernel adds straight into the flat gradient buffer
            dx = _ln_bwd_into(C, dy.contiguous(), x, ctx.gamma.detach(), mean, rstd, tg, tb, None)
            _sink.ready(ctx.gamma); _sink.ready(ctx.beta)
            return dx, None, None, None
        dx, dg, db = C.layernorm_bwd(dy.contiguous(), x, ctx.gamma.detach(), mean, rstd, None, None)
        dg = _grad_out(ctx.gamma, dg) if ctx.gamma.requires_grad else None
        db = _grad_out(ctx.beta, db) if ctx.beta.requires_grad else None
        return dx, dg, db, None


class _LayerNormForkFn(torch.autograd.Function):
    """``(x, LayerNorm(x))``: the pre-norm residual block's fork.  The first output IS the input; routing it through
    this node lets backward add the gradient that arrives over the residual connection inside the LayerNorm-backward
    kernel instead of in a separate elementwise pass."""

    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        C = _ext.load()
        y, mean, rstd = C.layernorm_fwd(x, gamma.detach(), beta.detach(), eps)
        ctx.save_for_backward(x, mean, rstd)
        ctx.gamma, ctx.beta = gamma, beta
        return x.view_as(x), y

    @staticmethod
    def backward(ctx, dres, dy):
        C = _ext.load()
        x, mean, rstd = ctx.saved_tensors
        tg, tb = _sink.target(ctx.gamma), _sink.target(ctx.beta)
        ad = None if dres is None else dres.contiguous()
        if dy is None:                      # the normalised branch was not used
            return dres, None, None, None
        if tg is not None and tb is not None:
            dx = _ln_bwd_into(C, dy.contiguous(), x, ctx.gamma.detach(), mean, rstd, tg, tb, ad)
            _sink.ready(ctx.gamma); _sink.ready(ctx.beta)
            return dx, None, None, None
        dx, dg, db = C.layernorm_bwd(dy.contiguous(), x, ctx.gamma.detach(), mean, rstd, None, None, ad)
        dg = _grad_out(ctx.gamma, dg) if ctx.gamma.requires_grad else None
        db = _grad_out(ctx.beta, db) if ctx.beta.requires_grad else None
        return dx, dg, db, None


def layer_norm_fork(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5):
    """``(x, layer_norm(x))`` for ``x + f(layer_norm(x))`` blocks: use the returned x for the residual connection."""
    if _ext.use_hip(x):
        return _LayerNormForkFn.apply(x.contiguous(), gamma, beta, eps)
    return x, F.layer_norm(x.float(), (x.shape[-1],), gamma, beta, eps).to(x.dtype)


def layer_norm(x: torch.Tensor, gamma: torch.Tensor, beta: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    if _ext.use_hip(x):
        return _LayerNormFn.apply(x.contiguous(), gamma, beta, eps)
    return F.layer_norm(x.float(), (x.shape[-1],), gamma, beta, eps).to(x.dtype)
