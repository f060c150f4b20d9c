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
    def forward(ctx, qkv):
        C = _ext.load()
        out, lse = C.attention_fwd(qkv)
        ctx.save_for_backward(qkv, out, lse)
        return out

    @staticmethod
    def backward(ctx, dout):
        C = _ext.load()
        qkv, out, lse = ctx.saved_tensors
        return C.attention_bwd(qkv, out, dout.contiguous(), lse)


def causal_attention(qkv: torch.Tensor, n_head: int) -> torch.Tensor:
    """Causal multi-head self-attention.  ``qkv [B,T,3*C]`` (q | k | v along the last
    axis, heads contiguous inside each) -> ``[B,T,C]``."""
    B, T, C3 = qkv.shape
    Cm = C3 // 3
    D = Cm // n_head
    if _ext.use_hip(qkv):
        out = _AttentionFn.apply(qkv.contiguous().view(B, T, 3, n_head, D))
        return out.view(B, T, Cm)
    q, k, v = qkv.float().view(B, T, 3, n_head, D).permute(2, 0, 3, 1, 4)
    att = (q @ k.transpose(-2, -1)) / math.sqrt(D)
    mask = torch.ones(T, T, dtype=torch.bool, device=qkv.device).tril()
    att = att.masked_fill(~mask, float("-inf")).softmax(dim=-1)
    out = (att @ v).permute(0, 2, 1, 3).reshape(B, T, Cm)
    return out.to(qkv.dtype)
