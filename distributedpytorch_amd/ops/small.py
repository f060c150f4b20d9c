"""fp32 NCHW ops of the reference's MNIST networks, fused for launch count.

``conv5_pool_relu``  = ``relu(max_pool2d(dropout2d?(conv2d(x, w, b)), 2))``
(reference multinode-DP/multinodeDP.py:89-90; pool and ReLU commute because both are
monotone, and channel dropout is a non-negative per-channel scale so it commutes
with both), ``linear_relu`` = ``relu?(x @ w^T + b)`` (DP.py:92,94), ``log_softmax``
and ``nll_loss`` (DP.py:95,152).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F

from . import _ext


class _Conv5PoolReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, drop):
        C = _ext.load()
        y, idx = C.conv5_pool_relu_fwd(x, w.detach(), b.detach(), drop)
        ctx.save_for_backward(x, w, y, idx)
        ctx.drop = drop
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.load()
        x, w, y, idx = ctx.saved_tensors
        dx, dw, db = C.conv5_pool_relu_bwd(x, w.detach(), dy.contiguous(), y, idx, ctx.drop, ctx.needs_input_grad[0])
        return (dx if ctx.needs_input_grad[0] else None), dw, db, None


def channel_dropout_mask(n: int, c: int, p: float, like: torch.Tensor, seed: int, offset: int,
                         step: Optional[torch.Tensor] = None) -> torch.Tensor:
    """``[n*c]`` mask of 0 or 1/(1-p): one Bernoulli per (sample, channel) = ``Dropout2d``.
    ``step``: device int64 step counter added to the stream position (hipGraph-replay safe)."""
    if like.is_cuda:
        return _ext.load().dropout_mask(n * c, p, seed, offset, like, step)
    g = torch.Generator().manual_seed(seed * 1000003 + offset)
    return (torch.rand(n * c, generator=g) >= p).float() / (1.0 - p)


def conv5_pool_relu(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, drop_mask: Optional[torch.Tensor] = None):
    """x ``[N,Cin,H,W]`` fp32, w ``[Cout,Cin,5,5]`` -> ``[N,Cout,(H-4)/2,(W-4)/2]``."""
    if _ext.use_hip(x):
        return _Conv5PoolReluFn.apply(x.contiguous(), w, b, drop_mask)
    y = F.conv2d(x, w, b)
    if drop_mask is not None:
        y = y * drop_mask.view(y.shape[0], y.shape[1], 1, 1)
    return F.relu(F.max_pool2d(y, 2))


class _LinearF32Fn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, b, relu):
        C = _ext.load()
        y = C.linear_f32_fwd(x, w.detach(), b.detach(), relu)
        ctx.save_for_backward(x, w, y)
        ctx.relu = relu
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.load()
        x, w, y = ctx.saved_tensors
        dx, dw, db = C.linear_f32_bwd(x, w.detach(), y, dy.contiguous(), ctx.relu, ctx.needs_input_grad[0])
        return (dx if ctx.needs_input_grad[0] else None), dw, db, None


def linear_f32(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, relu: bool = False) -> torch.Tensor:
    if _ext.use_hip(x):
        return _LinearF32Fn.apply(x.contiguous(), w, b, relu)
    y = F.linear(x, w, b)
    return F.relu(y) if relu else y


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
    """Element dropout (reference DP.py:93) with a counter-based mask generated on device."""
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
