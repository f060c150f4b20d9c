"""GPT-2 (small = 124M) on the gfx950 kernels: BASELINE.json config 5.

Not in the reference.  Parameter names follow the HuggingFace ``GPT2Model`` layout
(``wte wpe h.N.ln_1 h.N.attn.c_attn h.N.attn.c_proj h.N.ln_2 h.N.mlp.c_fc
h.N.mlp.c_proj ln_f``); ``checkpoint.py`` transposes the ``Conv1D`` weights at the
save/load edge.  All GEMMs (QKV, proj, MLP, tied LM head) run on the implicit-GEMM
MFMA kernel in bf16 with fp32 accumulation; attention is the causal flash kernel;
LayerNorm / GELU / residual adds are streaming kernels; the 50,257-class softmax
cross-entropy is fused and its backward runs in place over the logits.  Dropout (``GPT2Config.dropout``: embedding,
attention probabilities, residual branches) is drawn inside the kernels from a counter hash - no mask tensors.
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass
from typing import List

import torch
import torch.nn as nn

from .. import ops
from ..nn import Linear, LayerNorm


@dataclass
class GPT2Config:
    vocab_size: int = 50257
    n_positions: int = 1024
    n_embd: int = 768
    n_layer: int = 12
    n_head: int = 12
    dropout: float = 0.0       # embd_pdrop = attn_pdrop = resid_pdrop (HuggingFace's default is 0.1; 0.0 is the usual
                               # pre-training / benchmark setting and what bench.py runs)
    dropout_seed: int = 1234

    @property
    def vocab_padded(self) -> int:
        return (self.vocab_size + 63) // 64 * 64


class _DropSites:
    """Dropout streams of one module: site ``k`` of call ``n`` draws from the counter-hash stream at offset
    ``(n * sites + k) << 36`` - disjoint index ranges (a site has far fewer than 2^36 elements), masks a pure function
    of (seed, call, site, element), so backward re-draws them and nothing is stored.  Every module counts its own
    calls, which keeps pipeline stages (each holding only its blocks) independent of one another."""

    def __init__(self, p: float, seed: int, sites: int):
        self.p, self.seed, self.sites, self.calls = float(p), int(seed), sites, 0

    def begin(self, training: bool) -> bool:
        """One forward call of the owning module; returns whether dropout is active."""
        on = training and self.p > 0.0
        if on:
            self.calls += 1
        return on

    def site(self, k: int):
        return (self.p, self.seed, ((self.calls * self.sites + k) << 36) & ((1 << 62) - 1))


class Attention(nn.Module):
    def __init__(self, cfg: GPT2Config, layer: int = 0):
        super().__init__()
        self.n_head = cfg.n_head
        self.c_attn = Linear(cfg.n_embd, 3 * cfg.n_embd, init_std=0.02)
        self.c_proj = Linear(cfg.n_embd, cfg.n_embd, init_std=0.02 / math.sqrt(2 * cfg.n_layer))
        self._drop = _DropSites(cfg.dropout, cfg.dropout_seed + 7919 * (2 * layer + 1), 2)

    def forward(self, x, residual=None):
        if not self._drop.begin(self.training):
            return self.c_proj(ops.causal_attention(self.c_attn(x), self.n_head), residual=residual)
        # attn_pdrop inside the attention kernels, resid_pdrop fused with the residual add behind the projection
        a = ops.causal_attention(self.c_attn(x), self.n_head, drop=self._drop.site(0))
        return ops.dropout_residual(self.c_proj(a), *self._drop.site(1), residual=residual)


# Round 2 (8 x 1024 tokens: c_fc on the 128 x 128 kernel): 1 % SLOWER than the elementwise GELU pass it removes (599.0k vs
# 604.7k tok/s, docs/DESIGN.md §6).  Round 3 (16 x 1024: c_fc is 768 tiles of the 256 x 256 kernel, whose epilogue now
# writes the twin from registers): measured again, docs/ROUND3.md.  DPA_GELU_TWIN=0/1.
_GELU_TWIN = os.environ.get("DPA_GELU_TWIN", "1") == "1"


class MLP(nn.Module):
    def __init__(self, cfg: GPT2Config, layer: int = 0):
        super().__init__()
        self.c_fc = Linear(cfg.n_embd, 4 * cfg.n_embd, init_std=0.02)
        self.c_proj = Linear(4 * cfg.n_embd, cfg.n_embd, init_std=0.02 / math.sqrt(2 * cfg.n_layer))
        self._drop = _DropSites(cfg.dropout, cfg.dropout_seed + 7919 * (2 * layer + 2), 1)

    def forward(self, x, residual=None):
        if self._drop.begin(self.training):
            y = self.c_proj(self.c_fc(x), pre_gelu=True)
            return ops.dropout_residual(y, *self._drop.site(0), residual=residual)
        # gelu inside c_proj: its backward is an epilogue of c_proj's input-gradient GEMM.  With DPA_GELU_TWIN=1 c_fc's
        # epilogue also writes gelu(h) beside h, so the forward elementwise pass disappears too
        if _GELU_TWIN:
            h, act = self.c_fc(x, gelu_twin=True)
            return self.c_proj(h, residual=residual, pre_gelu=True, gelu_act=act)
        return self.c_proj(self.c_fc(x), residual=residual, pre_gelu=True)


class Block(nn.Module):
    def __init__(self, cfg: GPT2Config, layer: int = 0):
        super().__init__()
        self.ln_1 = LayerNorm(cfg.n_embd)
        self.attn = Attention(cfg, layer)
        self.ln_2 = LayerNorm(cfg.n_embd)
        self.mlp = MLP(cfg, layer)

    def forward(self, x):
        # pre-norm residual blocks: the forward add rides the projection GEMM's epilogue, the backward add of the
        # residual-path gradient rides the LayerNorm-backward kernel (LayerNorm.fork)
        x, h = self.ln_1.fork(x)
        x = self.attn(h, residual=x)
        x, h = self.ln_2.fork(x)
        return self.mlp(h, residual=x)


class Embedding(nn.Module):
    """Holds ``wte`` / ``wpe`` (fp32 masters + bf16 shadows)."""

    def __init__(self, cfg: GPT2Config):
        super().__init__()
        wte = torch.zeros(cfg.vocab_padded, cfg.n_embd)
        wte[: cfg.vocab_size].normal_(0.0, 0.02)
        self.wte = nn.Parameter(wte)
        self.wte._dpa_uses = 2            # embedding gather + tied LM head both add into its gradient
        self.wpe = nn.Parameter(torch.randn(cfg.n_positions, cfg.n_embd) * 0.01)
        self.wte_lp = None
        self.wpe_lp = None
        self.wte_lp_t = None               # bf16 [n_embd, vocab] twin for the tied head's input gradient (FlatParams)
        self._dpa_twins = ("wte",)
        self._dpa_tie = ("wte", "first")   # pipeline: this matrix also lives in the LM-head stage (parallel/pp.py)
        self._drop = _DropSites(cfg.dropout, cfg.dropout_seed, 1)

    def forward(self, tokens):
        x = ops.embedding(tokens, self.wte, self.wpe, self.wte_lp, self.wpe_lp)
        if x.is_cuda and x.dtype != torch.bfloat16:
            x = x.to(torch.bfloat16)
        if self._drop.begin(self.training):      # embd_pdrop
            x = ops.dropout_residual(x, *self._drop.site(0))
        return x


class GPT2(nn.Module):
    # FusedSGD.overlap_with_backward was measured to pay on this family (+0.9 % on one GPU; the ResNets lose 0.3 %:
    # their step is HBM-bound throughout) - what train.run_dp's --opt-overlap auto and bench.py consult
    optimizer_overlap_pays = True

    def __init__(self, cfg: GPT2Config = GPT2Config()):
        super().__init__()
        self.cfg = cfg
        self.emb = Embedding(cfg)
        self.h = nn.ModuleList([Block(cfg, i) for i in range(cfg.n_layer)])
        self.ln_f = LayerNorm(cfg.n_embd)

    def hidden(self, tokens: torch.Tensor) -> torch.Tensor:
        x = self.emb(tokens)
        for blk in self.h:
            x = blk(x)
        return self.ln_f(x)

    def forward(self, tokens: torch.Tensor) -> torch.Tensor:
        """Logits ``[B, T, vocab_padded]`` (tied LM head; columns >= vocab_size are padding)."""
        x = self.hidden(tokens)
        return ops.linear(x, self.emb.wte, self.emb.wte_lp, None, weight_lp_t=self.emb.wte_lp_t)

    def loss(self, tokens: torch.Tensor, targets: torch.Tensor) -> torch.Tensor:
        from ..ops import lmhead
        rows = tokens.numel()
        if lmhead.chunk_rows_for(rows, self.cfg.vocab_padded):
            # [rows, 50,304] logits would not fit one GEMM operand (32 x 1024 tokens: 1.65 G elements): head + loss row-chunked
            x = self.hidden(tokens)
            if lmhead.usable(x, self.emb.wte_lp, self.cfg.vocab_size):
                return lmhead.lm_head_loss(x, self.emb.wte, self.emb.wte_lp, self.emb.wte_lp_t, targets, self.cfg.vocab_size)
            logits = ops.linear(x, self.emb.wte, self.emb.wte_lp, None, weight_lp_t=self.emb.wte_lp_t)
        else:
            logits = self.forward(tokens)
        return ops.softmax_cross_entropy(logits.reshape(-1, logits.shape[-1]), targets.reshape(-1),
                                         self.cfg.vocab_size, inplace_backward=True)

    def pipeline_layers(self) -> List[nn.Module]:
        return [self.emb, *self.h, _LMHead(self)]


class _LMHead(nn.Module):
    """ln_f + tied head as a pipeline unit.  ``wte`` is the embedding's Parameter object: on one stage the two uses
    share it by reference; when the embedding and the head land on different pipeline stages each stage holds a copy
    (identical at start: every rank builds the model from the same seed) and the pipeline engine sums the two stages'
    gradients of it every step (``_dpa_tie``, parallel/pp.py), so both copies take the same update and stay tied."""

    def __init__(self, model: GPT2):
        super().__init__()
        self.ln_f = model.ln_f
        self.wte = model.emb.wte
        self.wte_lp = None                 # bound by FlatParams, like Embedding.wte_lp
        self.wte_lp_t = None               # ... and the [n_embd, vocab] twin the head's input gradient multiplies by
        self._dpa_twins = ("wte",)
        self._dpa_tie = ("wte", "last")

    def forward(self, x):
        return ops.linear(self.ln_f(x), self.wte, self.wte_lp, None, weight_lp_t=self.wte_lp_t)


def gpt2_small() -> GPT2:
    return GPT2(GPT2Config())


def gpt2_tiny(n_layer: int = 2, n_embd: int = 128, n_head: int = 2, vocab: int = 500, n_positions: int = 128,
              dropout: float = 0.0) -> GPT2:
    return GPT2(GPT2Config(vocab_size=vocab, n_positions=n_positions, n_embd=n_embd, n_layer=n_layer, n_head=n_head,
                           dropout=dropout))


def flops_per_token(cfg: GPT2Config, seq_len: int) -> float:
    """Training FLOPs per token (fwd+bwd = 3x forward), dense GEMMs + causal attention."""
    c, l = cfg.n_embd, cfg.n_layer
    gemm = 2.0 * l * (3 * c * c + c * c + 8 * c * c) + 2.0 * c * cfg.vocab_padded
    attn = 2.0 * l * 2 * seq_len * c / 2.0
    return 3.0 * (gemm + attn)
