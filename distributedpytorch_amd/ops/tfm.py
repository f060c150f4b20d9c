"""Transformer ops: token+position embedding and causal self-attention (GPT-2)."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

from . import _ext
from . import sink as _sink


class _EmbeddingFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, tok, wte, wte_lp, wpe, wpe_lp):
        C = _ext.load()
        ctx.save_for_backward(tok)
        ctx.wte, ctx.wpe = wte, wpe
        return C.embedding_fwd(tok, wte_lp, wpe_lp)

    @staticmethod
    def backward(ctx, dy):
        C = _ext.load()
        (tok,) = ctx.saved_tensors
        wte, wpe = ctx.wte, ctx.wpe
        t_te, t_pe = _sink.target(wte), _sink.target(wpe)
        dte = t_te if t_te is not None else torch.zeros(wte.shape, dtype=torch.float32, device=wte.device)
        dpe = t_pe if t_pe is not None else torch.zeros(wpe.shape, dtype=torch.float32, device=wpe.device)
        C.embedding_bwd(tok, dy.contiguous(), dte.view(wte.shape), dpe.view(wpe.shape))
        if t_te is not None:
            _sink.ready(wte)
        if t_pe is not None:
            _sink.ready(wpe)
        return None, (None if t_te is not None else dte), None, (None if t_pe is not None else dpe), None


def embedding(tok: torch.Tensor, wte: torch.Tensor, wpe: torch.Tensor, wte_lp=None, wpe_lp=None) -> torch.Tensor:
    """``wte[tok] + wpe[arange(T)]`` for ``tok [B,T]`` -> ``[B,T,C]``."""
    if _ext.use_hip(wte):
        from .conv import _lp
        return _EmbeddingFn.apply(tok.contiguous(), wte, _lp(wte, wte_lp), wpe, _lp(wpe, wpe_lp))
    T = tok.shape[1]
    return wte[tok] + wpe[:T].unsqueeze(0)


class _AttentionFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, drop):
        C = _ext.load()
        p, seed, offset = drop
        out, lse = C.attention_fwd(qkv, p, seed, offset)
        ctx.save_for_backward(qkv, out, lse)
        ctx.drop = drop
        return out

    @staticmethod
    def backward(ctx, dout):
        C = _ext.load()
        qkv, out, lse = ctx.saved_tensors
        p, seed, offset = ctx.drop
        return C.attention_bwd(qkv, out, dout.contiguous(), lse, p, seed, offset), None


def _cpu_seed(seed: int, offset: int) -> int:
    """CPU fallback streams: the host generator keeps 32 bits of its seed, so the (large, 2^36-aligned) offsets of the
    kernel streams are folded down instead of added."""
    return (seed * 1000003 + (offset >> 36) * 2654435761 + (offset & ((1 << 36) - 1))) & 0x7FFFFFFF


class _DropoutResidualFn(torch.autograd.Function):
    """``residual + dropout(x)`` on bf16 with the mask drawn in line from (seed, offset, element index): nothing is
    stored, backward re-draws the same mask (``dx = dropout(dy)``, ``dresidual = dy``)."""

    @staticmethod
    def forward(ctx, x, residual, p, seed, offset):
        ctx.meta = (p, seed, offset, residual is not None)
        return _ext.load().dropout_bf16(x, p, seed, offset, residual)

    @staticmethod
    def backward(ctx, dy):
        p, seed, offset, has_res = ctx.meta
        dy = dy.contiguous()
        return _ext.load().dropout_bf16(dy, p, seed, offset, None), (dy if has_res else None), None, None, None


def dropout_residual(x: torch.Tensor, p: float, seed: int, offset: int, residual: torch.Tensor = None,
                     training: bool = True) -> torch.Tensor:
    """``residual + dropout(x, p)`` (GPT-2's ``resid_pdrop`` / ``embd_pdrop``): one kernel, no mask tensor."""
    if not training or p <= 0.0:
        return x if residual is None else x + residual
    if _ext.use_hip(x) and x.dtype == torch.bfloat16 and x.numel() % 8 == 0:
        return _DropoutResidualFn.apply(x.contiguous(), None if residual is None else residual.contiguous(), p, seed, offset)
    g = torch.Generator().manual_seed(_cpu_seed(seed, offset))
    keep = (torch.rand(x.shape, generator=g) >= p).to(x.device, x.dtype) / (1.0 - p)
    y = x * keep
    return y if residual is None else y + residual


def causal_attention(qkv: torch.Tensor, n_head: int, drop=None) -> torch.Tensor:
    """Causal multi-head self-attention.  ``qkv [B,T,3*C]`` (q | k | v along the last
    axis, heads contiguous inside each) -> ``[B,T,C]``.  ``drop = (p, seed, offset)``: dropout on the attention
    probabilities (GPT-2's ``attn_pdrop``), drawn inside the kernels: element (b, h, q, k) is element
    ``((b*H + h)*T + q)*T + k`` of the counter-hash stream ``ops.dropout_mask`` reproduces."""
    B, T, C3 = qkv.shape
    Cm = C3 // 3
    D = Cm // n_head
    if drop is not None and drop[0] <= 0.0:
        drop = None
    if _ext.use_hip(qkv):
        out = _AttentionFn.apply(qkv.contiguous().view(B, T, 3, n_head, D), drop if drop is not None else (0.0, 0, 0))
        return out.view(B, T, Cm)
    q, k, v = qkv.float().view(B, T, 3, n_head, D).permute(2, 0, 3, 1, 4)
    att = (q @ k.transpose(-2, -1)) / math.sqrt(D)
    mask = torch.ones(T, T, dtype=torch.bool, device=qkv.device).tril()
    att = att.masked_fill(~mask, float("-inf")).softmax(dim=-1)
    if drop is not None:
        g = torch.Generator().manual_seed(_cpu_seed(drop[1], drop[2]))
        att = att * ((torch.rand(att.shape, generator=g) >= drop[0]).to(att.device, att.dtype) / (1.0 - drop[0]))
    out = (att @ v).permute(0, 2, 1, 3).reshape(B, T, Cm)
    return out.to(qkv.dtype)
