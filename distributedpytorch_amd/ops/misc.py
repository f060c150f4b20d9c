"""Pooling, layout conversion and elementwise ops (NHWC bf16 on the GPU)."""
from __future__ import annotations

import torch
import torch.nn.functional as F

from . import _ext


class _MaxPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        C = _ext.load()
        y, idx = C.maxpool_fwd(x)
        ctx.save_for_backward(idx)
        ctx.hw = (x.shape[1], x.shape[2])
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.load()
        (idx,) = ctx.saved_tensors
        return C.maxpool_bwd(dy.contiguous(), idx, ctx.hw[0], ctx.hw[1])


def max_pool_3x3_s2_nhwc(x: torch.Tensor) -> torch.Tensor:
    """3x3 / stride 2 / pad 1 max-pool (the ResNet stem pool)."""
    if _ext.use_hip(x):
        return _MaxPoolFn.apply(x)
    return F.max_pool2d(x.permute(0, 3, 1, 2), 3, 2, 1).permute(0, 2, 3, 1).contiguous()


class _AvgPoolFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        C = _ext.load()
        ctx.hw = (x.shape[1], x.shape[2])
        return C.avgpool_fwd(x)

    @staticmethod
    def backward(ctx, dy):
        C = _ext.load()
        return C.avgpool_bwd(dy.contiguous(), ctx.hw[0], ctx.hw[1])


def global_avg_pool_nhwc(x: torch.Tensor) -> torch.Tensor:
    """``[N,H,W,C] -> [N,C]``."""
    if _ext.use_hip(x):
        return _AvgPoolFn.apply(x)
    return x.mean(dim=(1, 2))


def to_nhwc(x: torch.Tensor, c_pad: int) -> torch.Tensor:
    """NCHW (fp32/bf16) -> NHWC with channels zero-padded to ``c_pad`` (bf16 on GPU)."""
    if _ext.use_hip(x):
        C = _ext.load()
        return C.nchw_to_nhwc_pad(x.contiguous(), c_pad)
    y = x.permute(0, 2, 3, 1)
    if c_pad > y.shape[-1]:
        y = F.pad(y, (0, c_pad - y.shape[-1]))
    return y.contiguous()


class _AddReluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        C = _ext.load()
        y = C.ew2(a, b, 1)
        ctx.save_for_backward(y)
        return y

    @staticmethod
    def backward(ctx, dy):
        C = _ext.load()
        (y,) = ctx.saved_tensors
        d = C.ew2(dy.contiguous(), y, 2)
        return d, d


def add_relu(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    if _ext.use_hip(a):
        return _AddReluFn.apply(a.contiguous(), b.contiguous())
    return F.relu(a + b)


class _AddFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, b):
        return _ext.load().ew2(a, b, 0)

    @staticmethod
    def backward(ctx, dy):
        return dy, dy


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    if _ext.use_hip(a):
        return _AddFn.apply(a.contiguous(), b.contiguous())
    return a + b


class _GeluFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x):
        ctx.save_for_backward(x)
        return _ext.load().ew2(x, x, 3)

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        return _ext.load().ew2(dy.contiguous(), x, 4)


def gelu(x: torch.Tensor) -> torch.Tensor:
    """tanh-approximated GELU (GPT-2's ``gelu_new``)."""
    if _ext.use_hip(x):
        return _GeluFn.apply(x.contiguous())
    return F.gelu(x, approximate="tanh")
