"""Fused softmax cross-entropy (mean reduction) on logits in bf16 or fp32.

Replaces ``log_softmax`` + ``nll_loss`` (reference multinode-DP/multinodeDP.py:95,152)
with one forward kernel (online max/sum, loss accumulated on device - no host sync)
and one backward kernel writing ``(softmax - onehot) / B`` in place over the logits.
"""
from __future__ import annotations

import torch
import torch.nn.functional as F

import os

from . import _ext

_FUSED = os.environ.get("DPA_NO_FUSED_CE", "0") != "1"


class _SoftmaxCEFusedFn(torch.autograd.Function):
    """One pass: the forward kernel also overwrites the logits with their gradient (for an upstream gradient of 1);
    backward multiplies by the real upstream gradient only when it is not 1 (tested on the device)."""

    @staticmethod
    def forward(ctx, logits, targets, n_classes, ignore_index):
        C = _ext.load()
        scale = 1.0 / max(logits.shape[0], 1)
        loss = C.softmax_ce_fused(logits, targets, n_classes, scale, ignore_index)
        ctx.save_for_backward(logits)            # now d loss / d logits
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        (d,) = ctx.saved_tensors
        _ext.load().scale_unless_one(d, g.reshape(1).float().contiguous())
        return d, None, None, None


class _SoftmaxCEFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits, targets, n_classes, ignore_index, inplace):
        C = _ext.load()
        B = logits.shape[0]
        scale = 1.0 / max(B, 1)
        loss, lse = C.softmax_ce_fwd(logits, targets, n_classes, scale, ignore_index)
        ctx.save_for_backward(logits, targets, lse)
        ctx.meta = (n_classes, scale, ignore_index, inplace)
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        C = _ext.load()
        logits, targets, lse = ctx.saved_tensors
        n_classes, scale, ignore_index, inplace = ctx.meta
        gs = g.reshape(1).float().contiguous()
        # inplace: the logits are dead after the loss (saves a [B, V] allocation for the 50k-class head)
        d = C.softmax_ce_bwd(logits, targets, lse, gs, n_classes, scale, ignore_index, inplace)
        return d, None, None, None, None


def softmax_cross_entropy(logits: torch.Tensor, targets: torch.Tensor, n_classes: int = None,
                          ignore_index: int = -100, inplace_backward: bool = False) -> torch.Tensor:
    """Mean cross-entropy of ``logits [B, ld]`` (only the first ``n_classes`` columns
    are classes; the rest is padding) against ``targets [B]``.  Returns an fp32 scalar.

    The mean is over ALL rows (like the reference's ``F.nll_loss`` on a full batch);
    rows whose target equals ``ignore_index`` contribute zero.  ``inplace_backward`` hands the
    ``logits`` buffer over: it is overwritten with the gradient - by the backward kernel, or, when a
    gradient is going to be asked for (bf16, row length a multiple of 8 and <= 65,536), already by the ONE forward
    kernel that then does both passes' work - so the logits must not be read after this call.
    """
    if logits.dim() != 2:
        logits = logits.reshape(-1, logits.shape[-1])
        targets = targets.reshape(-1)
    n_classes = logits.shape[1] if n_classes is None else n_classes
    if _ext.use_hip(logits):
        logits = logits.contiguous()
        if (inplace_backward and _FUSED and torch.is_grad_enabled() and logits.requires_grad
                and _ext.load().softmax_ce_fused_ok(logits, n_classes)):
            # the logits are dead after the loss AND a gradient will be asked for: one kernel reads them once and leaves
            # their gradient in their place (csrc/loss.hip ce_fused_kernel)
            return _SoftmaxCEFusedFn.apply(logits, targets.contiguous(), n_classes, ignore_index)
        return _SoftmaxCEFn.apply(logits.contiguous(), targets.contiguous(), n_classes, ignore_index, inplace_backward)
    lg = logits[:, :n_classes].float()
    per_row = F.cross_entropy(lg, targets, reduction="none", ignore_index=ignore_index)
    return per_row.sum() / max(logits.shape[0], 1)
