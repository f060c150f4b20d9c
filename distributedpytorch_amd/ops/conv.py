"""Convolution and Linear on the implicit-GEMM MFMA kernels.

Layouts are the kernels' own: activations NHWC bf16, filters ``[K, R, S, C]`` (KRSC).
Master weights stay fp32; a bf16 shadow (written by the fused optimizer) is what the
GEMMs read.  Gradients of the weights are produced in fp32 by the wgrad kernel
(fp32 atomics over the pixel split), so no bf16 rounding ever enters the gradient
all-reduce.  A Linear layer is the 1x1 / H=W=1 case of the same kernels.

Replaces the cuDNN / cuBLAS calls behind ``nn.Conv2d`` / ``nn.Linear`` in the
reference (multinode-DP/multinodeDP.py:82-86).
"""
from __future__ import annotations

import os
from typing import Optional

import torch
import torch.nn.functional as F

from . import _ext
from . import side as _side
from . import sink as _sink
from . import wgroup as _wgroup


def _lp(weight: torch.Tensor, weight_lp: Optional[torch.Tensor]) -> torch.Tensor:
    """bf16 copy of a master weight: the optimizer-maintained shadow when present."""
    if weight_lp is not None:
        return weight_lp
    C = _ext.load()
    out = torch.empty(weight.shape, dtype=torch.bfloat16, device=weight.device)
    C.cast_f32_to_bf16(weight.detach().contiguous(), out)
    return out


class _ConvFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, weight_lp, bias, stride, pad, want_stats, act, wshape, stats_buf=None, weight_lp_t=None,
                residual=None, pre_gelu=False, gelu_twin=False, gelu_act=None):
        C = _ext.load()
        w = _lp(weight, weight_lp).view(wshape)   # KRSC view (a Linear weight [out,in] is K,1,1,C)
        # the second output (statistics / GELU twin) is not differentiable: without this autograd hands backward a
        # ZERO-FILLED tensor of its shape - a 100 MB fill kernel per GPT-2 block (seen in the kernel trace)
        ctx.set_materialize_grads(False)
        ctx.pre_gelu = pre_gelu
        if pre_gelu:                              # y = layer(gelu(x)): the GELU's backward rides the dgrad epilogue
            ctx.gelu_in = x
            # gelu(x) already exists when the layer that produced x wrote it beside x (gelu_twin): no elementwise pass
            x = gelu_act if gelu_act is not None else C.ew2(x, x, 3)
        # + residual in the epilogue; gelu_twin: second output gelu(y) for the next layer (second slot of the result)
        outs = C.conv_fprop(x, w, bias, stride, pad, act, want_stats, stats_buf, residual, gelu_twin)
        ctx.has_residual = residual is not None
        ctx.save_for_backward(x, w)
        ctx.weight = weight
        ctx.bias = bias
        ctx.geom = (stride, pad, act)
        ctx.wt = weight_lp_t        # optimizer-maintained [C,R,S,K] twin (FlatParams), or None
        if want_stats or gelu_twin:
            ctx.mark_non_differentiable(outs[1])
            return outs[0], outs[1]
        return outs[0], None

    @staticmethod
    def backward(ctx, dy, _dstats):
        C = _ext.load()
        x, w = ctx.saved_tensors
        stride, pad, act = ctx.geom
        assert act == 0, "fused activation epilogues are forward-only"
        if dy is None:                            # (materialize_grads is off) nothing flows back through this layer
            return (None,) * 15
        dy = dy.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            wt = ctx.wt if ctx.wt is not None else C.transpose_filter(w)
            dx = C.conv_dgrad(dy, wt, x.shape[1], x.shape[2], stride, pad, None, None, None, None, None,
                              ctx.gelu_in if ctx.pre_gelu else None)
        weight, bias = ctx.weight, ctx.bias
        # bias gradient = column sums of dy: produced inside the wgrad kernel (one more MFMA against a tile of
        # ones) when a weight gradient is computed anyway, by the column-sum kernels otherwise
        db_tgt = None
        if bias is not None and bias.requires_grad:
            db_tgt = _sink.target(bias)
            if db_tgt is None:
                db = db_tgt = torch.zeros(bias.shape, dtype=torch.float32, device=bias.device)
        if weight.requires_grad:
            tgt = _sink.target(weight)
            if tgt is not None:     # accumulate straight into the flat grad buffer, off the critical path (ops/side.py)
                linear_shaped = w.shape[1] == 1 and w.shape[2] == 1 and x.shape[1] == 1 and x.shape[2] == 1 and stride == 1 and pad == 0
                if (linear_shaped and _wgroup.ENABLED and _side.active(x) and (db_tgt is None or db is None)
                        and _wgroup.query(x.shape[0], w.shape[0], w.shape[3])[0]
                        and (x.data_ptr() | dy.data_ptr() | tgt.data_ptr()) % 16 == 0):
                    # a large Linear layer: held back and launched TOGETHER with its neighbours' weight gradients
                    # (ops/wgroup.py, csrc/wgrad_wide.hip); the batcher reports weight and bias ready when it launches
                    rows, cout, cin = x.shape[0], w.shape[0], w.shape[3]
                    _wgroup.for_device(x.device.index).add(
                        x.view(rows, cin), dy.view(rows, cout), tgt.view(cout, cin), db_tgt,
                        [weight] + ([bias] if db_tgt is not None else []), _wgroup.query(rows, cout, cin)[1])
                    return (dx, None, None, None, None, None, None, None, None, None, None,
                            dy if (ctx.has_residual and ctx.needs_input_grad[11]) else None, None, None, None)
                if _side.active(x) and (db_tgt is None or db is None):
                    tv = tgt.view(w.shape)
                    _side.run(lambda: C.conv_wgrad(x, dy, tv, stride, pad, db_tgt), (x, dy))
                else:
                    C.conv_wgrad(x, dy, tgt.view(w.shape), stride, pad, db_tgt)
                _sink.ready(weight)
            else:
                dw = torch.zeros(weight.shape, dtype=torch.float32, device=weight.device)
                C.conv_wgrad(x, dy, dw.view(w.shape), stride, pad, db_tgt)
        elif db_tgt is not None:
            C.colsum(dy, db_tgt)
        if db_tgt is not None and db is None:
            _sink.ready(bias)
        # the residual connection passes the gradient through unchanged
        dres = dy if (ctx.has_residual and ctx.needs_input_grad[11]) else None
        return dx, dw, None, db, None, None, None, None, None, None, None, dres, None, None, None


def conv2d_nhwc(x: torch.Tensor, weight: torch.Tensor, weight_lp: Optional[torch.Tensor] = None,
                bias: Optional[torch.Tensor] = None, stride: int = 1, padding: int = 0, want_stats: bool = False,
                stats_buf: Optional[torch.Tensor] = None, weight_lp_t: Optional[torch.Tensor] = None):
    """``x [N,H,W,C]`` * ``weight [K,R,S,C]`` -> ``y [N,Ho,Wo,K]``.

    Returns ``(y, stats)``; ``stats`` holds per-channel (sum, sumsq) partials of ``y``
    from the GEMM epilogue when ``want_stats`` (GPU only), else None.  ``stats_buf`` is an
    optional persistent zero-initialised ``[rows,2,K]`` buffer to accumulate them into (the
    BatchNorm finalize kernel re-zeroes it), which avoids one allocation + memset per layer.
    """
    if _ext.use_hip(x):
        return _ConvFn.apply(x, weight, weight_lp, bias, stride, padding, want_stats, 0, tuple(weight.shape), stats_buf,
                             weight_lp_t)
    y = F.conv2d(x.permute(0, 3, 1, 2), weight.to(x.dtype).permute(0, 3, 1, 2),
                 None if bias is None else bias.to(x.dtype), stride=stride, padding=padding)
    return y.permute(0, 2, 3, 1).contiguous(), None


def linear(x: torch.Tensor, weight: torch.Tensor, weight_lp: Optional[torch.Tensor] = None,
           bias: Optional[torch.Tensor] = None, weight_lp_t: Optional[torch.Tensor] = None,
           residual: Optional[torch.Tensor] = None, pre_gelu: bool = False,
           gelu_twin: bool = False, gelu_act: Optional[torch.Tensor] = None):
    """``y = x @ weight^T + bias (+ residual)`` with ``weight [out, in]``; any leading dims on ``x``.
    ``weight_lp_t`` is the optional bf16 ``[in, out]`` twin the input gradient multiplies by; ``residual`` (the
    output's shape) is added in the GEMM epilogue; ``pre_gelu``: ``y = gelu_tanh(x) @ weight^T ...`` with the GELU's
    backward folded into the input-gradient GEMM's epilogue; ``gelu_twin``: returns ``(y, gelu_tanh(y))``, the second
    written by the same epilogue (None off the GPU) - hand it to the next layer as ``gelu_act`` with ``pre_gelu``.
    Every GEMM runs on this repository's own kernels: csrc/conv_igemm.hip (128 x 128 tiles) or, for large launches
    such as the tied LM head (8192 x 50304 x 768 forward, 8192 x 768 x 50304 split-K input gradient),
    csrc/gemm_wide.hip (persistent 256 x 256 tiles); no vendor BLAS is called."""
    if _ext.use_hip(x):
        lead = x.shape[:-1]
        cin, cout = weight.shape[1], weight.shape[0]
        x4 = x.reshape(-1, 1, 1, cin)
        wt = None if weight_lp_t is None else weight_lp_t.view(cin, 1, 1, cout)
        r4 = None if residual is None else residual.contiguous().reshape(-1, 1, 1, cout)
        ga = None if gelu_act is None else gelu_act.reshape(-1, 1, 1, cin)
        y, twin = _ConvFn.apply(x4.contiguous(), weight, weight_lp, bias, 1, 0, False, 0, (cout, 1, 1, cin), None, wt, r4, pre_gelu,
                                gelu_twin, ga)
        if gelu_twin:
            return y.reshape(*lead, cout), twin.reshape(*lead, cout)
        return y.reshape(*lead, cout)
    if pre_gelu:
        x = F.gelu(x, approximate="tanh")
    y = F.linear(x, weight.to(x.dtype), None if bias is None else bias.to(x.dtype))
    y = y if residual is None else y + residual
    return (y, None) if gelu_twin else y
