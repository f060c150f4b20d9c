"""Flat parameter / gradient / bf16-shadow buffers.

All parameters of a model (or pipeline stage) are re-pointed into ONE contiguous
fp32 master buffer, their ``.grad`` into ONE contiguous fp32 gradient buffer, and -
on the GPU - a bf16 shadow buffer holds the low-precision copy the MFMA kernels
read.  Consequences:

* the optimizer is a single fused kernel launch over the flat buffer (replaces the
  reference's per-parameter Python loop, multinode-DP/multinodeDP.py:156);
* ``zero_grad`` is folded into that kernel (reference: 8 memsets, DP.py:150);
* gradient buckets for the data-parallel all-reduce are plain slices of the flat
  gradient buffer - no copy into bucket memory, few large RCCL calls
  (reference: one blocking all_reduce per tensor, DP.py:126-129);
* parameters are laid out in REVERSE registration order, so the slices that become
  ready first during backward (the last layers) are contiguous at the front.
"""
from __future__ import annotations

from typing import Dict, Iterable, List, Optional, Tuple

import torch
import torch.nn as nn

from ..ops import sink as _sink

ALIGN = 64  # elements; keeps every parameter 256-byte (fp32) / 128-byte (bf16) aligned


class FlatParams:
    def __init__(self, module: nn.Module, lp_dtype: Optional[torch.dtype] = None, reverse: bool = True,
                 use_lp: bool = True):
        params: List[Tuple[str, nn.Parameter]] = []
        seen = set()
        for name, p in module.named_parameters():
            if p.requires_grad and id(p) not in seen:
                seen.add(id(p))
                params.append((name, p))
        if not params:
            raise ValueError("module has no trainable parameters")
        self.module = module
        self.names = [n for n, _ in params]
        self.params = [p for _, p in params]
        order = list(range(len(params)))
        if reverse:
            order.reverse()
        self.order = order                      # flat position -> parameter index
        device = self.params[0].device
        self.device = device
        if lp_dtype is None and device.type == "cuda":
            lp_dtype = torch.bfloat16
        self.lp_dtype = lp_dtype if (device.type == "cuda" and use_lp) else None

        offsets: Dict[int, int] = {}
        cursor = 0
        for idx in order:
            offsets[idx] = cursor
            cursor += (self.params[idx].numel() + ALIGN - 1) // ALIGN * ALIGN
        self.total = cursor
        self.offsets = offsets
        self.master = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.grad = torch.zeros(self.total, dtype=torch.float32, device=device)
        self.lp = torch.zeros(self.total, dtype=self.lp_dtype, device=device) if self.lp_dtype is not None else None

        self._grad_views: Dict[int, torch.Tensor] = {}
        for idx, p in enumerate(self.params):
            off, n = offsets[idx], p.numel()
            self.master[off:off + n].copy_(p.detach().reshape(-1).float())
            p.data = self.master[off:off + n].view(p.shape)
            g = self.grad[off:off + n].view(p.shape)
            p.grad = g
            self._grad_views[id(p)] = g
        # transposed twin of the bf16 shadow for conv filters ([C,R,S,K]) and Linear weights ([in,out]) - what the
        # input-gradient GEMM consumes - refreshed by ONE
        # batched kernel after every optimizer step instead of one transpose launch per layer per backward
        self.lp_t = None
        self._tr_table = None
        self._tr_tiles = 0
        if self.lp is not None:
            from ..ops import _ext
            tt = _ext.load().transpose_tile()       # tile edge of csrc/conv_igemm.hip's batched transpose
            rows, tiles, done = [], 0, set()
            for mod in module.modules():
                for pname in self._twin_names(mod):
                    w = getattr(mod, pname, None)
                    if not isinstance(w, nn.Parameter) or w.dim() not in (2, 4) or id(w) in done:
                        continue
                    i = next((k for k, q in enumerate(self.params) if q is w), None)
                    if i is None:
                        continue
                    done.add(id(w))                 # a tied matrix (GPT-2 wte) is transposed once
                    K, R, S, C = w.shape if w.dim() == 4 else (w.shape[0], 1, 1, w.shape[1])   # Linear [out,in] = K,1,1,C
                    rows.append([offsets[i], offsets[i], K, R * S, C, tiles])
                    tiles += ((K + tt - 1) // tt) * ((C + tt - 1) // tt) * R * S
            if rows:
                self.lp_t = torch.zeros(self.total, dtype=self.lp_dtype, device=device)
                self._tr_table = torch.tensor(rows, dtype=torch.int64, device=device)
                self._tr_tiles = tiles
                self._tr_rows, self._tr_tile_edge, self._tr_sub = rows, tt, {}
        self.refresh_lp()
        self._bind_shadows()

    @staticmethod
    def _twin_names(mod: nn.Module):
        """Parameters of ``mod`` that want a transposed bf16 twin ``<name>_lp_t``: ``weight`` when the module has a
        ``weight_lp_t`` attribute, plus whatever it lists in ``_dpa_twins`` (GPT-2's tied embedding matrix)."""
        names = list(getattr(mod, "_dpa_twins", ()))
        if hasattr(mod, "weight_lp_t") and "weight" not in names:
            names.append("weight")
        return names

    # ------------------------------------------------------------------ views
    def slice_of(self, p: nn.Parameter) -> Tuple[int, int]:
        idx = next(i for i, q in enumerate(self.params) if q is p)
        return self.offsets[idx], self.offsets[idx] + p.numel()

    def lp_view(self, p: nn.Parameter) -> Optional[torch.Tensor]:
        if self.lp is None:
            return None
        a, b = self.slice_of(p)
        return self.lp[a:b].view(p.shape)

    def _bind_shadows(self) -> None:
        """Point every module's ``<name>_lp`` attribute at its parameter's bf16 shadow."""
        if self.lp is None:
            return
        index = {id(p): i for i, p in enumerate(self.params)}
        for mod in self.module.modules():
            for pname, p in list(mod._parameters.items()):
                if p is None or id(p) not in index:
                    continue
                attr = pname + "_lp"
                if hasattr(mod, attr):
                    i = index[id(p)]
                    off = self.offsets[i]
                    setattr(mod, attr, self.lp[off:off + p.numel()].view(p.shape))
                    if self.lp_t is not None and pname in self._twin_names(mod) and p.dim() in (2, 4):
                        if p.dim() == 4:
                            K, R, S, C = p.shape
                            setattr(mod, pname + "_lp_t", self.lp_t[off:off + p.numel()].view(C, R, S, K))
                        else:
                            setattr(mod, pname + "_lp_t", self.lp_t[off:off + p.numel()].view(p.shape[1], p.shape[0]))

    def refresh_lp(self) -> None:
        if self.lp is None:
            return
        from ..ops import _ext
        _ext.load().cast_f32_to_bf16(self.master, self.lp)
        self.refresh_transposed()

    def refresh_transposed(self) -> None:
        """Re-derive the transposed conv filters from the bf16 shadow (one launch for the whole model)."""
        if self.lp_t is None:
            return
        from ..ops import _ext
        _ext.load().transpose_filters_batched(self.lp, self.lp_t, self._tr_table, self._tr_tiles)

    def refresh_transposed_range(self, start: int, end: int) -> None:
        """The same for the filters that lie in ``flat[start:end)`` only (a gradient bucket: ``optim.FusedSGD``'s
        overlapped mode refreshes a bucket's twins while backward is still running).  Sub-tables are cached."""
        if self.lp_t is None:
            return
        key = (start, end)
        sub = self._tr_sub.get(key)
        if sub is None:
            rows, tiles, tt = [], 0, self._tr_tile_edge
            for src, dst, K, RS, C, _ in self._tr_rows:
                if start <= src < end:
                    rows.append([src, dst, K, RS, C, tiles])
                    tiles += ((K + tt - 1) // tt) * ((C + tt - 1) // tt) * RS
            sub = self._tr_sub[key] = (torch.tensor(rows, dtype=torch.int64, device=self.device) if rows else None, tiles)
        if sub[0] is not None:
            from ..ops import _ext
            _ext.load().transpose_filters_batched(self.lp, self.lp_t, sub[0], sub[1])

    # ------------------------------------------------------------------ grads
    def zero_grad(self) -> None:
        self.grad.zero_()

    def restore_grad_views(self) -> None:
        """Re-attach ``p.grad`` views (e.g. after something set them to None)."""
        for p in self.params:
            if p.grad is None or p.grad.data_ptr() != self._grad_views[id(p)].data_ptr():
                p.grad = self._grad_views[id(p)]

    def install_sink(self, ready_cb=None) -> None:
        """Let wgrad kernels accumulate straight into THIS flat gradient buffer (ops/sink.py: the sink object hangs off
        the parameters it covers, so other FlatParams in the process keep theirs)."""
        self.uninstall_sink()
        self.sink = _sink.GradSink(self.params, self._grad_views, ready_cb)

    def uninstall_sink(self) -> None:
        s = getattr(self, "sink", None)
        if s is not None:
            s.uninstall()
        self.sink = None

    # ------------------------------------------------------------------ buckets
    def plan_buckets(self, bucket_bytes: int, first_bucket_bytes: Optional[int] = None) -> List["Bucket"]:
        """Contiguous slices of the flat gradient buffer in backward-readiness order.
        Buckets only close on parameter boundaries.  ``first_bucket_bytes`` bounds the
        FIRST bucket (ready earliest - starts communication early) and the LAST one
        (first layers: nothing is left to overlap it with, so its collective is
        exposed): the trailing parameters that fit ``first_bucket_bytes`` are split off
        the final bucket when it is larger than that."""
        return plan_buckets([self.params[i].numel() for i in self.order],
                            [self.offsets[i] for i in self.order], self.order, self.total,
                            bucket_bytes, first_bucket_bytes)


class Bucket:
    __slots__ = ("start", "end", "param_indices", "pending", "work")

    def __init__(self, start: int, end: int, param_indices: List[int]):
        self.start, self.end, self.param_indices = start, end, param_indices
        self.pending = 0
        self.work = None

    @property
    def numel(self) -> int:
        return self.end - self.start

    def __repr__(self):
        return f"Bucket([{self.start}:{self.end}], params={len(self.param_indices)})"


def plan_buckets(numels: List[int], offsets: List[int], order: List[int], total: int, bucket_bytes: int,
                 first_bucket_bytes: Optional[int] = None, elem_bytes: int = 4) -> List[Bucket]:
    """Pure function (unit-tested on CPU): greedy packing in flat order."""
    cap = max(1, int(bucket_bytes) // elem_bytes)
    first_cap = max(1, int(first_bucket_bytes) // elem_bytes) if first_bucket_bytes else cap
    buckets: List[Bucket] = []
    cur_start, cur_params, cur_elems = 0, [], 0
    for pos, idx in enumerate(order):
        limit = first_cap if not buckets else cap
        n = numels[pos]
        if cur_params and cur_elems + n > limit:
            buckets.append(Bucket(cur_start, offsets[pos], cur_params))
            cur_start, cur_params, cur_elems = offsets[pos], [], 0
        cur_params.append(idx)
        cur_elems += n
    if cur_params:
        buckets.append(Bucket(cur_start, total, cur_params))
    # cap the LAST bucket (the un-overlappable tail): split the trailing parameters that fit first_cap off it
    if first_bucket_bytes and buckets and len(buckets[-1].param_indices) > 1:
        last = buckets[-1]
        n_last = len(last.param_indices)
        first_pos = len(order) - n_last                     # flat position of the bucket's first parameter
        if sum(numels[first_pos:]) > first_cap:
            tail, k = 0, len(order)
            while k - 1 > first_pos and tail + numels[k - 1] <= first_cap:
                k -= 1
                tail += numels[k]
            if first_pos < k < len(order):
                buckets[-1] = Bucket(last.start, offsets[k], last.param_indices[:k - first_pos])
                buckets.append(Bucket(offsets[k], total, last.param_indices[k - first_pos:]))
    return buckets
