"""Weight gradients of several Linear layers in ONE launch (``csrc/wgrad_wide.hip``).

A transformer block's backward produces four weight gradients within a few hundred microseconds of each other
(GPT-2: fc2, fc, attention projection, qkv - 36 + 36 + 9 + 27 output tiles of 256 x 256 over the same 16,384 tokens).
Launched one by one each needs a 7-9-way token split to occupy 256 CUs, and every split flushes a 256 KiB fp32 tile per
CU; together they are 108 tiles, a 2-way split gives every CU one item of 128 K steps, and the four layers cost 232 us
instead of 380 (``scripts/bench_wgrad_wide.py``; 1.0 vs 0.61 PFLOP/s).  Nobody needs a weight gradient before its
bucket's all-reduce, so ``ops/conv.py`` hands eligible problems to the batcher of their device instead of launching them:

* problems are held (operands referenced) until the held tiles reach ``DPA_WGRAD_GROUP_TILES`` (default 96), eight
  problems are held, or a problem over a different number of rows arrives; then one launch goes to the side stream and
  every parameter of the group is reported ready to its gradient sink (``ops/sink.py``) - readiness moves from "its
  kernel was enqueued" to "its group's kernel was enqueued", so bucket launches see a consistent picture;
* anybody who joins or borrows the side stream (``ops/side.py``: bucket launch, optimizer, ``DataParallel.finish``, the
  end-of-backward callback) flushes first, so a held-back gradient can never be missed by its consumer;
* the end of the backward pass flushes too (autograd engine callback), whoever reads gradients next.

``DPA_WGRAD_GROUP=0`` turns the batching off (each problem then goes through ``conv_wgrad``, which still routes large
ones to the wide kernel in a launch of their own).  Reference op: the cuBLAS weight-gradient GEMMs behind
``nn.Linear`` (multinode-MP/multinodeMP.py:94-95), at transformer scale.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Tuple

import torch

from . import _ext
from . import side as _side
from . import sink as _sink

ENABLED = os.environ.get("DPA_WGRAD_GROUP", "1") != "0"
FLUSH_TILES = int(os.environ.get("DPA_WGRAD_GROUP_TILES", "96"))
MAX_PROBLEMS = 8                                   # csrc/kernels.h WGRAD_WIDE_MAX

_query_cache: Dict[Tuple[int, int, int], Tuple[bool, int]] = {}


def _graph_task_id() -> int:
    """Id of the running backward pass (-1 outside one; 0 on a torch without the query: the guard is then inert)."""
    fn = getattr(torch._C, "_current_graph_task_id", None)
    return int(fn()) if fn is not None else 0


def query(rows: int, cout: int, cin: int) -> Tuple[bool, int]:
    """(can and should this Linear-shaped problem go to the wide kernel, its number of 256 x 256 tiles)."""
    key = (rows, cout, cin)
    hit = _query_cache.get(key)
    if hit is None:
        ok, tiles = _ext.load().wgrad_wide_query(rows, cout, cin)
        hit = _query_cache[key] = (bool(ok), int(tiles))
    return hit


class WgradBatcher:
    """The held-back problems of one device."""

    def __init__(self, index: int):
        self.index = index
        self.items: List[tuple] = []           # (x2d, dy2d, dw_view, dbias_view or None, [params to report ready])
        self.tiles = 0
        self.rows = -1
        self.callback_queued = False
        self.task_id = -1                      # autograd graph task the held problems (and the queued callback) belong to
        self.launches = 0                      # statistics: launches / problems so far
        self.problems = 0
        self.dropped = 0                       # problems of a backward pass that died before its end-of-backward callback

    def add(self, x: torch.Tensor, dy: torch.Tensor, dw: torch.Tensor, dbias: Optional[torch.Tensor], params, tiles: int) -> None:
        rows = x.shape[0]
        tid = _graph_task_id()
        if tid != self.task_id:
            # another backward pass than the one the state belongs to: that one ended without its callback (an exception
            # inside backward).  Its held problems are gradients of a pass the caller has abandoned - adding them to the
            # buffers of THIS pass would be wrong - and its callback will never run.
            if self.items:
                self.dropped += len(self.items)
                self.items, self.tiles, self.rows = [], 0, -1
            self.callback_queued = False
            self.task_id = tid
        if self.items and (rows != self.rows or len(self.items) >= MAX_PROBLEMS):
            self.flush()
        self.items.append((x, dy, dw, dbias, list(params)))
        self.rows = rows
        self.tiles += tiles
        if not self.callback_queued:
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
                self.callback_queued = True
            except RuntimeError:               # not inside a backward pass: nobody would flush for us
                self.flush()
                return
        if self.tiles >= FLUSH_TILES or len(self.items) >= MAX_PROBLEMS:
            self.flush()

    def flush(self) -> None:
        if not self.items:
            return
        items, self.items, self.tiles, self.rows = self.items, [], 0, -1
        C = _ext.load()
        xs = [it[0] for it in items]
        dys = [it[1] for it in items]
        dws = [it[2] for it in items]
        dbs = [it[3] for it in items]
        self.launches += 1
        self.problems += len(items)
        if _side.active(xs[0]):
            _side.run(lambda: C.wgrad_wide_group(xs, dys, dws, dbs), tuple(xs) + tuple(dys))
        else:
            C.wgrad_wide_group(xs, dys, dws, dbs)
        for it in items:                       # only now are these gradients on their way
            for p in it[4]:
                _sink.ready(p)

    def _end_of_backward(self) -> None:
        self.callback_queued = False
        self.flush()


_per_device: Dict[int, WgradBatcher] = {}


def for_device(index: int) -> WgradBatcher:
    b = _per_device.get(index)
    if b is None:
        b = _per_device[index] = WgradBatcher(index)
    return b


def _flush_device(index: int) -> None:
    b = _per_device.get(index)
    if b is not None:
        b.flush()


_side.add_flush_hook(_flush_device)


def flush_all() -> None:
    for b in _per_device.values():
        b.flush()
