"""Modules over the gfx950 ops.  Parameters are fp32 masters in the kernels' layout
(conv filters KRSC); ``*_lp`` attributes are bf16 shadows maintained by the fused
optimizer (``parallel.flat.FlatParams``), absent on CPU.  ``checkpoint.py`` converts
to the stock PyTorch ``state_dict`` layout (OIHW, fp32) at the save/load edge.
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn as nn

from .. import ops


def _pad8(c: int) -> int:
    return (c + 7) // 8 * 8


class Conv2d(nn.Module):
    """NHWC convolution, filter ``[out, kh, kw, in_padded]``."""

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int, stride: int = 1, padding: int = 0,
                 bias: bool = False):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.kernel_size, self.stride, self.padding = kernel_size, stride, padding
        self.in_padded = _pad8(in_channels)
        w = torch.zeros(out_channels, kernel_size, kernel_size, self.in_padded)
        fan_out = out_channels * kernel_size * kernel_size
        w[..., :in_channels].normal_(0.0, math.sqrt(2.0 / fan_out))   # kaiming_normal_(mode="fan_out", relu)
        self.weight = nn.Parameter(w)
        self.bias = nn.Parameter(torch.zeros(out_channels)) if bias else None
        self.weight_lp: Optional[torch.Tensor] = None
        self.weight_lp_t: Optional[torch.Tensor] = None    # [C,R,S,K] twin for dgrad, kept by FlatParams

    def forward(self, x: torch.Tensor, want_stats: bool = False, stats_buf=None):
        y, stats = ops.conv2d_nhwc(x, self.weight, self.weight_lp, self.bias, self.stride, self.padding, want_stats,
                                   stats_buf, self.weight_lp_t)
        return (y, stats) if want_stats else y

    def extra_repr(self):
        return (f"{self.in_channels}, {self.out_channels}, kernel_size={self.kernel_size}, stride={self.stride}, "
                f"padding={self.padding}, layout=NHWC/KRSC")


class BatchNorm2d(nn.Module):
    """Training-mode batch norm over NHWC, optionally fused with a residual add and ReLU."""

    def __init__(self, num_features: int, eps: float = 1e-5, momentum: float = 0.1, zero_init: bool = False):
        super().__init__()
        self.num_features, self.eps, self.momentum = num_features, eps, momentum
        self.weight = nn.Parameter(torch.zeros(num_features) if zero_init else torch.ones(num_features))
        self.bias = nn.Parameter(torch.zeros(num_features))
        self.register_buffer("running_mean", torch.zeros(num_features))
        self.register_buffer("running_var", torch.ones(num_features))
        self.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))

        self._stats_buf = None
        self._stats_buf_bwd = None

    STATS_ROWS = 32     # replicas of the (sum, sumsq) accumulator the conv epilogue adds into

    def stats_buffer(self, device) -> torch.Tensor:
        """Persistent zeroed ``[rows,2,C]`` fp32 accumulator for the producing conv's epilogue;
        the finalize kernel re-zeroes it after reading, so it is allocated exactly once."""
        if self._stats_buf is None or self._stats_buf.device != device:
            self._stats_buf = torch.zeros(self.STATS_ROWS, 2, self.num_features, dtype=torch.float32, device=device)
        return self._stats_buf

    def stats_buffer_bwd(self, device) -> torch.Tensor:
        """Same, for the two backward reductions (sum g, sum g*xhat)."""
        if self._stats_buf_bwd is None or self._stats_buf_bwd.device != device:
            self._stats_buf_bwd = torch.zeros(self.STATS_ROWS, 2, self.num_features, dtype=torch.float32, device=device)
        return self._stats_buf_bwd

    def forward(self, x, stats=None, relu: bool = False, residual=None, reset_stats: bool = False):
        return ops.batch_norm_nhwc(x, self.weight, self.bias, self.running_mean, self.running_var, self.training,
                                   self.momentum, self.eps, relu=relu, residual=residual, stats=stats,
                                   reset_stats=reset_stats)


def conv_bn(conv: Conv2d, bn: BatchNorm2d, x, relu: bool, residual=None):
    """conv -> BN (-> +residual) (-> ReLU); on the GPU the BN statistics come from the conv epilogue."""
    fuse = x.is_cuda and bn.training
    if fuse:
        y, stats = conv(x, want_stats=True, stats_buf=bn.stats_buffer(x.device))
        return bn(y, stats=stats, relu=relu, residual=residual, reset_stats=True)
    return bn(conv(x), relu=relu, residual=residual)


class Linear(nn.Module):
    """``y = x W^T + b`` on the implicit-GEMM kernel; ``out_features`` may be padded to a
    multiple of 8 internally (``out_padded``) - the extra rows are zero and sliced off."""

    def __init__(self, in_features: int, out_features: int, bias: bool = True, init_std: Optional[float] = None):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.out_padded = _pad8(out_features)
        assert in_features % 8 == 0, "in_features must be a multiple of 8"
        w = torch.zeros(self.out_padded, in_features)
        if init_std is None:
            bound = 1.0 / math.sqrt(in_features)
            w[:out_features].uniform_(-bound, bound)
        else:
            w[:out_features].normal_(0.0, init_std)
        self.weight = nn.Parameter(w)
        if bias:
            b = torch.zeros(self.out_padded)
            if init_std is None:
                b[:out_features].uniform_(-1.0 / math.sqrt(in_features), 1.0 / math.sqrt(in_features))
            self.bias = nn.Parameter(b)
        else:
            self.bias = None
        self.weight_lp: Optional[torch.Tensor] = None
        self.weight_lp_t: Optional[torch.Tensor] = None    # [in, out] twin for the input gradient, kept by FlatParams

    def forward(self, x, keep_padding: bool = False, residual=None, pre_gelu: bool = False, gelu_twin: bool = False,
                gelu_act=None):
        """``gelu_twin``: returns ``(y, gelu(y))`` (second output of the same GEMM epilogue; None off the GPU);
        ``pre_gelu`` + ``gelu_act``: ``y = gelu(x) W^T + b`` where ``gelu_act`` is that already-computed gelu(x)."""
        if residual is not None and self.out_padded != self.out_features:
            return self.forward(x, keep_padding, None, pre_gelu, False, gelu_act) + residual    # padded outputs: no fused residual
        if gelu_twin and self.out_padded != self.out_features:
            return self.forward(x, keep_padding, residual, pre_gelu, False, gelu_act), None
        y = ops.linear(x, self.weight, self.weight_lp, self.bias, self.weight_lp_t, residual, pre_gelu,
                       gelu_twin=gelu_twin, gelu_act=gelu_act)
        if gelu_twin:
            return y
        if not keep_padding and self.out_padded != self.out_features:
            y = y[..., : self.out_features]
        return y


class LayerNorm(nn.Module):
    def __init__(self, dim: int, eps: float = 1e-5):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))
        self.eps = eps

    def forward(self, x):
        return ops.layer_norm(x, self.weight, self.bias, self.eps)

    def fork(self, x):
        """``(x, self(x))``; use the returned x for the residual connection around the normalised branch."""
        return ops.layer_norm_fork(x, self.weight, self.bias, self.eps)
