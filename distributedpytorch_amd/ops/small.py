"""fp32 NCHW ops of the reference's MNIST networks, fused for launch count.

Every weight gradient is ADDED into the parallel engine's flat buffer when a gradient sink covers the parameter
(ops/sink.py) - no per-parameter ATen add / fill kernels - and dropout masks are drawn inside the kernels.

``conv5_pool_relu``  = ``relu(max_pool2d(dropout2d?(conv2d(x, w, b)), 2))``
(reference multinode-DP/multinodeDP.py:89-90; pool and ReLU commute because both are
monotone, and channel dropout is a non-negative per-channel scale so it commutes
with both), ``linear_relu`` = ``relu?(x @ w^T + b)`` (DP.py:92,94), ``log_softmax``
and ``nll_loss`` (DP.py:95,152).
"""
from __future__ import annotations

from typing import Optional, Tuple

import torch
import torch.nn.functional as F

from . import _ext
from . import sink as _sink


# Dropout argument of the fused kernels: (p, seed, offset, step) - ``step`` an optional device int64 counter.  The mask
# is drawn INSIDE the kernel from a counter hash of (seed, offset + step, element); backward needs no mask at all (a
# dropped element has output 0, a kept one was scaled by the constant 1/(1-p)).
Drop = Optional[Tuple[float, int, int, Optional[torch.Tensor]]]
_NO_DROP = (0.0, 0, 0, None)


def _sink_pair(w, b):
    tw, tb = _sink.target(w), _sink.target(b)
    return (tw, tb) if (tw is not None and tb is not None) else (None, None)


class _Conv5PoolReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, drop):
        C = _ext.load()
        p, seed, offset, step = drop
        y, idx = C.conv5_pool_relu_fwd(x, w.detach(), b.detach(), p, seed, offset, step)
        ctx.save_for_backward(x, y, idx)
        ctx.w, ctx.b = w, b                        # the Parameter objects themselves: the gradient sink hangs off them
        ctx.gscale = 1.0 / (1.0 - p) if p > 0 else 1.0
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.load()
        x, y, idx = ctx.saved_tensors
        w, b = ctx.w, ctx.b
        tw, tb = _sink_pair(w, b)
        dx, dw, db = C.conv5_pool_relu_bwd(x, w.detach(), dy.contiguous(), y, idx, ctx.gscale, ctx.needs_input_grad[0], tw, tb)
        if tw is not None:                         # added straight into the flat gradient buffer
            _sink.ready(w); _sink.ready(b)
            dw = db = None
        return (dx if ctx.needs_input_grad[0] else None), dw, db, None


def channel_dropout_mask(n: int, c: int, p: float, like: torch.Tensor, seed: int, offset: int,
                         step: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[n*c]`` mask of 0 or 1/(1-p): one Bernoulli per (sample, channel) = ``Dropout2d`` - what the fused conv kernel
    draws in line on the GPU (same counter hash: ``dropout_mask`` reproduces it), a host generator on the CPU."""
    if like.is_cuda:
        return _ext.load().dropout_mask(n * c, p, seed, offset, like, step)
    g = torch.Generator().manual_seed(seed * 1000003 + offset)
    return (torch.rand(n * c, generator=g) >= p).float() / (1.0 - p)


def conv5_pool_relu(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, drop: Drop = None):
    """x ``[N,Cin,H,W]`` fp32, w ``[Cout,Cin,5,5]`` -> ``[N,Cout,(H-4)/2,(W-4)/2]``; ``drop``: channel dropout behind it."""
    if _ext.use_hip(x):
        return _Conv5PoolReluFn.apply(x.contiguous(), w, b, drop if drop is not None else _NO_DROP)
    y = F.conv2d(x, w, b)
    if drop is not None and drop[0] > 0:
        mask = channel_dropout_mask(y.shape[0], y.shape[1], drop[0], y, drop[1], drop[2], drop[3])
        y = y * mask.view(y.shape[0], y.shape[1], 1, 1)
    return F.relu(F.max_pool2d(y, 2))


class _LinearF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, relu, drop, lsm):
        C = _ext.load()
        p, seed, offset, step = drop
        if lsm:
            y = C.linear_log_softmax_fwd(x, w.detach(), b.detach())
        else:
            y = C.linear_f32_fwd(x, w.detach(), b.detach(), relu, p, seed, offset, step)
        ctx.save_for_backward(x, y)
        ctx.w, ctx.b = w, b
        ctx.relu, ctx.lsm = relu, lsm
        ctx.gscale = 1.0 / (1.0 - p) if p > 0 else 1.0
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.load()
        x, y = ctx.saved_tensors
        w, b = ctx.w, ctx.b
        tw, tb = _sink_pair(w, b)
        # a dropout behind a ReLU-less layer would need its mask (0 outputs are not all drops): not a case these nets have
        dx, dw, db = C.linear_f32_bwd(x, w.detach(), y, dy.contiguous(), ctx.relu, ctx.gscale, ctx.lsm, ctx.needs_input_grad[0], tw, tb)
        if tw is not None:
            _sink.ready(w); _sink.ready(b)
            dw = db = None
        return (dx if ctx.needs_input_grad[0] else None), dw, db, None, None, None


def linear_f32(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, relu: bool = False, drop: Drop = None) -> torch.Tensor:
    """``dropout(relu?(x @ w^T + b))`` (reference DP.py:92-94) as one kernel; ``drop`` needs ``relu`` (see backward)."""
    if drop is not None and drop[0] > 0 and not relu:
        raise ValueError("linear_f32: fused dropout is implemented behind a ReLU only")
    if _ext.use_hip(x):
        return _LinearF32Fn.apply(x.contiguous(), w, b, relu, drop if drop is not None else _NO_DROP, False)
    y = F.linear(x, w, b)
    y = F.relu(y) if relu else y
    if drop is not None and drop[0] > 0:
        g = torch.Generator().manual_seed(drop[1] * 1000003 + drop[2])
        y = y * ((torch.rand(y.numel(), generator=g) >= drop[0]).float().view(y.shape) / (1.0 - drop[0]))
    return y


def linear_log_softmax(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """``log_softmax(x @ w^T + b, dim=1)`` (reference DP.py:94-95) as one kernel forward; backward folds the
    log-softmax gradient into the Linear's dx | dw launch."""
    if _ext.use_hip(x) and w.shape[0] <= 64:
        return _LinearF32Fn.apply(x.contiguous(), w, b, False, _NO_DROP, True)
    return log_softmax(linear_f32(x, w, b))


class _LinearNllFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, target, bump):
        loss, logp = _ext.load().linear_lsm_nll_fwd(x, w.detach(), b.detach(), target, bump)
        ctx.save_for_backward(x, logp, target)
        ctx.w, ctx.b = w, b
        ctx.mark_non_differentiable(logp)
        return loss, logp

    @staticmethod
    def backward(ctx, gloss, _glogp):
        x, logp, target = ctx.saved_tensors
        w, b = ctx.w, ctx.b
        tw, tb = _sink_pair(w, b)
        dx, dw, db = _ext.load().linear_lsm_nll_bwd(x, w.detach(), logp, gloss.reshape(1).contiguous(), target,
                                                    ctx.needs_input_grad[0], tw, tb)
        if tw is not None:
            _sink.ready(w); _sink.ready(b)
            dw = db = None
        return (dx if ctx.needs_input_grad[0] else None), dw, db, None, None


def linear_log_softmax_nll(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, target: torch.Tensor,
                           bump: Optional[torch.Tensor] = None):
    """``(F.nll_loss(log_softmax(x @ w^T + b), target), log_probs)`` - the reference's classifier head and its loss
    (DP.py:94-95,152) as ONE kernel forward and ONE backward (log-softmax, NLL, dx | dw | db).  ``log_probs`` comes back
    for accuracy counting and carries no gradient.  ``bump``: a device int64 counter incremented by the forward kernel
    (the dropout stream of ``models.mnist``: saves a one-thread launch per replayed step)."""
    if _ext.use_hip(x) and w.shape[0] <= 16 and w.shape[1] <= 448:
        return _LinearNllFn.apply(x.contiguous(), w, b, target.contiguous(), bump)
    logp = linear_log_softmax(x, w, b)
    if bump is not None:
        bump += 1
    return nll_loss(logp, target), logp.detach()


class _DropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, mask):
        ctx.save_for_backward(mask)
        return _ext.load().mul_f32(x, mask).view(x.shape)

    @staticmethod
    def backward(ctx, dy):
        (mask,) = ctx.saved_tensors
        return _ext.load().mul_f32(dy.contiguous(), mask).view(dy.shape), None


def dropout(x: torch.Tensor, p: float, training: bool, seed: int, offset: int,
            step: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Stand-alone element dropout (mask tensor + multiply): for callers outside the fused layers."""
    if not training or p == 0.0:
        return x
    if _ext.use_hip(x):
        mask = _ext.load().dropout_mask(x.numel(), p, seed, offset, x, step)
        return _DropoutFn.apply(x.contiguous(), mask)
    g = torch.Generator().manual_seed(seed * 1000003 + offset)
    mask = (torch.rand(x.numel(), generator=g) >= p).float().view(x.shape) / (1.0 - p)
    return x * mask


class _LogSoftmaxFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        y = _ext.load().log_softmax_fwd(x)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        (y,) = ctx.saved_tensors
        return _ext.load().log_softmax_bwd(dy.contiguous(), y)


def log_softmax(x: torch.Tensor) -> torch.Tensor:
    """Row-wise log-softmax of a 2-D fp32 tensor (dim=1, reference DP.py:95)."""
    if _ext.use_hip(x):
        return _LogSoftmaxFn.apply(x.contiguous())
    return F.log_softmax(x, dim=1)


class _NllFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logp, target):
        ctx.save_for_backward(target)
        ctx.shape = tuple(logp.shape)
        return _ext.load().nll_fwd(logp, target)

    @staticmethod
    def backward(ctx, g):
        (target,) = ctx.saved_tensors
        return _ext.load().nll_bwd(g.reshape(1).contiguous(), target, ctx.shape[0], ctx.shape[1]), None


def nll_loss(logp: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """Mean negative log-likelihood (reference DP.py:152)."""
    if _ext.use_hip(logp):
        return _NllFn.apply(logp.contiguous(), target.contiguous())
    return F.nll_loss(logp, target)
