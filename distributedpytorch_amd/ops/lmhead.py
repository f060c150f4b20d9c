"""Row-chunked LM head + softmax cross-entropy: the logits never exist as one tensor.

GPT-2's tied head is ``logits = h @ wte^T`` with 50,304 (padded) columns: at 32 x 1024 tokens the logits are
1.65 G elements - 3.3 GB in bf16, and past the 2^30-element range of the kernels' 32-bit gather offsets, which used to cap
the per-GPU batch at 20 sequences.  Here the rows are processed in chunks (default 16,384: whole rounds of ``gemm_wide.hip``
tiles): per chunk the head GEMM writes an 824 MB logits block, the one-pass cross-entropy kernel
(``csrc/loss.hip ce_fused_kernel``) turns it into its own gradient in place, and the chunk's two backward GEMMs - input
gradient and weight gradient - run at once, while the block is still hot; then the block is reused.  So forward ALSO
produces ``d loss / d h`` and ``d loss / d wte`` (for an upstream gradient of 1, as the one-pass kernel does), and
backward only multiplies them by the real upstream gradient - on the device, no host read.

The weight gradient is accumulated in a private fp32 buffer and added to the parameter's gradient (the flat gradient
buffer when a sink is installed) in backward, so gradient-ready tracking and ``no_sync`` accumulation see it where they
expect it: during backward.

Reference: the classifier of multinode-DP/multinodeDP.py:94-95,152 (``fc2`` -> ``log_softmax`` -> ``nll_loss``) at
language-model scale.  CPU / fp32 path: the plain composition (the oracle of the tests).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import _ext
from . import sink as _sink

_FORCE_ROWS = int(os.environ.get("DPA_LMHEAD_CHUNK_ROWS", "0"))      # > 0: always chunk, by this many rows
# chunk length when the logits do not fit one operand: 16,384 rows x 768 -> 50,304 is the shape whose tiles make whole rounds
# on 256 CUs (the input gradient: 64 x 4 tiles of 256 x 192); 8,192-row chunks ran it at half a round
# (GPT-2 at 32 x 1024: profiles/r3/lmhead_chunk_ab.txt)
_CHUNK_ROWS = int(os.environ.get("DPA_LMHEAD_CHUNK_DEFAULT", "16384"))
LIMIT = 1 << 30                                                     # elements of one operand of the GEMM kernels


def chunk_rows_for(rows: int, cols: int) -> int:
    """0 = the un-chunked path is fine; else the chunk length."""
    if _FORCE_ROWS > 0:
        return _FORCE_ROWS
    if rows * cols < LIMIT:
        return 0
    fit = max(1024, (LIMIT - 1) // cols // 1024 * 1024)
    return min(_CHUNK_ROWS, fit)


class _LMHeadLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x2d, weight, weight_lp, weight_lp_t, targets, n_classes, ignore_index, chunk):
        C = _ext.load()
        rows, cin = x2d.shape
        cout = weight_lp.shape[0]
        scale = 1.0 / max(rows, 1)
        w4 = weight_lp.view(cout, 1, 1, cin)
        wt4 = weight_lp_t.view(cin, 1, 1, cout) if weight_lp_t is not None else C.transpose_filter(w4)
        dx = torch.empty_like(x2d)
        dw = torch.zeros(cout, cin, dtype=torch.float32, device=x2d.device)
        loss = None
        for r0 in range(0, rows, chunk):
            r1 = min(rows, r0 + chunk)
            xc = x2d[r0:r1].view(r1 - r0, 1, 1, cin)
            logits = C.conv_fprop(xc, w4, None, 1, 0, 0, False, None)[0]
            g2 = logits.view(r1 - r0, cout)
            part = C.softmax_ce_fused(g2, targets[r0:r1], n_classes, scale, ignore_index)      # g2 now holds d loss / d logits
            loss = part if loss is None else loss + part
            dx[r0:r1].copy_(C.conv_dgrad(logits, wt4, 1, 1, 1, 0, None, None).view(r1 - r0, cin))
            C.conv_wgrad(xc, logits, dw.view(cout, 1, 1, cin), 1, 0)
            del logits, g2
        ctx.save_for_backward(dx, dw)
        ctx.weight = weight
        return loss.view(())

    @staticmethod
    def backward(ctx, g):
        C = _ext.load()
        dx, dw = ctx.saved_tensors
        gs = g.reshape(1).float().contiguous()
        C.scale_unless_one(dx, gs)
        weight = ctx.weight
        dweight = None
        if weight.requires_grad:
            tgt = _sink.target(weight)
            if tgt is not None:
                C.axpy_dev(tgt.view(-1), dw.view(-1), gs)      # += g * dw, straight into the flat gradient buffer
                _sink.ready(weight)
            else:
                dweight = dw * gs
                if dweight.shape != weight.shape:
                    dweight = dweight.view(weight.shape)
        return dx, dweight, None, None, None, None, None, None


def lm_head_loss(x: torch.Tensor, weight: torch.Tensor, weight_lp: Optional[torch.Tensor], weight_lp_t: Optional[torch.Tensor],
                 targets: torch.Tensor, n_classes: int, ignore_index: int = -100, chunk_rows: Optional[int] = None) -> torch.Tensor:
    """Mean cross-entropy of ``(x @ weight^T)[:, :n_classes]`` against ``targets`` (``x [..., C]``, ``weight [V_pad, C]``),
    with the logits formed and consumed ``chunk_rows`` rows at a time.  GPU + bf16 shadow only; callers fall back to
    ``ops.linear`` + ``ops.softmax_cross_entropy`` elsewhere."""
    x2d = x.reshape(-1, x.shape[-1]).contiguous()
    t = targets.reshape(-1).contiguous()
    chunk = chunk_rows or chunk_rows_for(x2d.shape[0], weight.shape[0]) or x2d.shape[0]
    return _LMHeadLossFn.apply(x2d, weight, weight_lp, weight_lp_t, t, n_classes, ignore_index, int(chunk))


def usable(x: torch.Tensor, weight_lp: Optional[torch.Tensor], n_classes: int) -> bool:
    if not (_ext.use_hip(x) and weight_lp is not None and x.dtype == torch.bfloat16 and torch.is_grad_enabled()):
        return False
    probe = torch.empty(0, weight_lp.shape[0], dtype=torch.bfloat16, device=x.device)
    return bool(_ext.load().softmax_ce_fused_ok(probe, n_classes))
