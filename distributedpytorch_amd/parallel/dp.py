"""Data-parallel engine: bucketed gradient all-reduce over RCCL, overlapped with backward.

Semantics are the reference's synchronous SGD (multinode-DP/multinodeDP.py:118-129,155):
after backward every rank holds the MEAN of the per-rank gradients.  Mechanism is
MI355X-first instead of the reference's 8 blocking per-tensor all-reduces after
backward:

* gradients live in one flat fp32 buffer (``FlatParams``) laid out in backward order;
  a bucket is a contiguous slice of it, so RCCL reduces in place with no pack/unpack;
* a bucket's all-reduce is launched asynchronously the moment its last gradient has
  been produced (autograd post-accumulate hooks / the wgrad sink), on RCCL's own HIP
  stream, while backward of earlier layers keeps the compute stream busy;
* the ``1 / world_size`` of DP.py:129 is not a kernel: it is folded into the fused
  optimizer's ``grad_scale``;
* bucket sizes are chosen for the xGMI mesh (7 point-to-point links per GPU): few,
  large messages (default 32 MiB) so each link's share dwarfs per-message latency,
  with a small FIRST bucket so communication starts early; the small CNN's 87 KB of
  gradients collapse into ONE all-reduce (the reference issues 8).
* replicas start identical: same seed (as the reference, DP.py:138) AND a broadcast
  of rank 0's flat master buffer.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from .. import comm as _comm
from ..ops import side as _side
from ..ops import sink as _sink
from .flat import Bucket, FlatParams

_AR_FROM_SIDE = os.environ.get("DPA_AR_FROM_SIDE", "1") != "0"


def shard_bounds(total: int, world_size: int, align: int = 64) -> List[int]:
    """``world_size + 1`` offsets cutting ``[0, total)`` into contiguous, ``align``-element aligned, near-equal shards
    (the last one takes the remainder; trailing shards may be empty for tiny models)."""
    per = -(-total // world_size)
    per = -(-per // align) * align
    return [min(total, r * per) for r in range(world_size)] + [total]


class _Works:
    def __init__(self, works):
        self.works = works

    def wait(self):
        for w in self.works:
            w.wait()
        return True


class DataParallel:
    def __init__(self, module: nn.Module, comm: Optional[_comm.Communicator] = None, flat: Optional[FlatParams] = None,
                 bucket_mb: float = 32.0, first_bucket_mb: float = 4.0, broadcast_params: bool = True,
                 use_sink: bool = True, use_lp: bool = True, shard_optimizer: bool = False):
        self.module = module
        self.comm = comm if comm is not None else _comm.Communicator()
        self.flat = flat if flat is not None else FlatParams(module, use_lp=use_lp)
        self.world_size = self.comm.world_size
        self.grad_scale = 1.0 / self.world_size
        self.buckets: List[Bucket] = self.flat.plan_buckets(int(bucket_mb * 2 ** 20), int(first_bucket_mb * 2 ** 20))
        # shard_optimizer (ZeRO stage 1; not in the reference): rank r owns flat[shards[r]:shards[r+1]) - its gradient
        # is reduced ONTO r only, r alone keeps optimizer state for it and updates it, and the updated parameters are
        # broadcast back (optim.ShardedSGD).  Same bytes on the links as the all-reduce it replaces (a reduce plus a
        # broadcast per shard), 1/N of the optimizer's memory and HBM traffic per GPU.
        self.shards: Optional[List[int]] = shard_bounds(self.flat.total, self.world_size) if shard_optimizer else None
        self._bucket_of: Dict[int, Bucket] = {}
        self._uses: Dict[int, int] = {}
        for b in self.buckets:
            for idx in b.param_indices:
                p = self.flat.params[idx]
                self._bucket_of[id(p)] = b
                self._uses[id(p)] = int(getattr(p, "_dpa_uses", 1))
        self._remaining: Dict[int, int] = {}
        self._sync = True
        self._armed = False
        self._launched: List[Bucket] = []
        self._next_bucket = 0
        self.stats = {"allreduce_calls": 0, "allreduce_bytes": 0, "bucket_launches": 0}
        # called with a bucket right after its reduction has been ISSUED (b.work is the handle): optim.FusedSGD's
        # overlapped mode updates the bucket's parameters behind it, while backward is still running
        self.bucket_hook = None
        for p in self.flat.params:
            p.register_post_accumulate_grad_hook(self._on_grad)
        if use_sink and self.flat.device.type == "cuda":
            self.flat.install_sink(self._on_sink_ready)
        if broadcast_params and self.world_size > 1:
            self.comm.broadcast(self.flat.master, src=0)
            self.flat.refresh_lp()
            for buf in module.buffers():
                self.comm.broadcast(buf, src=0)
        self._arm()

    # ------------------------------------------------------------------ forward passthrough
    def __call__(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def __getattr__(self, name):
        return getattr(self.__dict__["module"], name)

    # ------------------------------------------------------------------ readiness tracking
    def _arm(self) -> None:
        self._remaining = dict(self._uses)
        for b in self.buckets:
            b.pending = sum(1 for _ in b.param_indices)
            b.work = None
        self._launched = []
        self._next_bucket = 0
        self._armed = True

    def no_sync(self):
        """Context manager: accumulate gradients locally (micro-batches) without communicating."""
        engine = self

        class _NoSync:
            def __enter__(self):
                engine._sync = False

            def __exit__(self, *exc):
                engine._sync = True
                return False
        return _NoSync()

    def _mark(self, p: torch.Tensor) -> None:
        if not self._sync:
            return
        if not self._armed:
            self._arm()
        pid = id(p)
        left = self._remaining.get(pid)
        if left is None or left <= 0:
            return
        self._remaining[pid] = left - 1
        if left - 1 > 0:
            return
        b = self._bucket_of[pid]
        b.pending -= 1
        if b.pending == 0:
            self._launch_in_order()

    def _launch_in_order(self) -> None:
        """Launch every complete bucket whose predecessors have all been launched.  Collectives are matched across
        ranks by call order, so the order must not depend on the order in which gradients happen to become ready
        on a rank: buckets always go out in index order (= backward order, so nothing is held back in practice)."""
        while self._next_bucket < len(self.buckets) and self.buckets[self._next_bucket].pending == 0:
            b = self.buckets[self._next_bucket]
            self._next_bucket += 1
            if b.work is None:
                self._launch(b)

    def _on_grad(self, p: torch.Tensor) -> None:          # autograd wrote p.grad (in place into the flat view)
        self._mark(p)

    def _on_sink_ready(self, p: torch.Tensor) -> None:    # a wgrad kernel accumulated straight into the flat buffer
        self._mark(p)

    def _launch(self, b: Bucket) -> None:
        view = self.flat.grad[b.start:b.end]
        if view.is_cuda and _AR_FROM_SIDE and _side.active(view):
            # The bucket's last gradients come from the side stream (ops/side.py).  Issued from THAT stream (which
            # first waits for the compute stream, a device-side wait), the collective is ordered behind both streams
            # and the compute stream itself never stalls on outstanding weight-gradient kernels.
            with _side.on_side_stream(view.device):
                self._issue(b, view)
            return
        if view.is_cuda:
            _side.join(view.device)       # weight gradients are produced on the side stream (ops/side.py)
        self._issue(b, view)

    def _issue(self, b: Bucket, view: torch.Tensor) -> None:
        if self.shards is not None and self.world_size > 1:
            works = []
            for r in range(self.world_size):            # every rank walks the owners in the same order
                lo, hi = max(b.start, self.shards[r]), min(b.end, self.shards[r + 1])
                if lo < hi:
                    works.append(self.comm.reduce(self.flat.grad[lo:hi], dst=r, async_op=True))
            b.work = _Works(works)
        else:
            b.work = self.comm.all_reduce(view, async_op=True)
        self._launched.append(b)
        self.stats["bucket_launches"] += 1
        # only collectives that were really issued count: a single process without a process group (bench.py at N = 1)
        # gets stand-in handles from Communicator.all_reduce and moves no bytes
        done = b.work.works if isinstance(b.work, _Works) else [b.work]
        if any(not isinstance(w, _comm._DoneWork) for w in done):
            self.stats["allreduce_calls"] += 1
            self.stats["allreduce_bytes"] += view.numel() * 4
        if self.bucket_hook is not None:
            self.bucket_hook(b)

    # ------------------------------------------------------------------ step boundary
    def finish(self) -> None:
        """Call after backward, before the optimizer: launches any bucket that did not
        complete through hooks (unused parameters) and waits for all reductions.  The
        wait only orders the compute stream behind RCCL's stream - no host block."""
        if self.flat.device.type == "cuda":
            _side.join(self.flat.device)
        if self._sync:
            for b in self.buckets:
                if b.work is None:
                    self._launch(b)
            for b in self._launched:
                b.work.wait()
        self._armed = False

    def reduce_all(self) -> None:
        """All-reduce every bucket now (no overlap).  For steps replayed from a hipGraph: no Python
        hook fires during a replay, so the engine launches the buckets itself afterwards."""
        self._arm()
        for b in self.buckets:
            self._launch(b)
        for b in self._launched:
            b.work.wait()
        self._armed = False

    def average_gradients(self) -> None:
        """Reference-shaped API (DP.py:118-129): reduce everything now, blocking, and scale."""
        self.finish()
        if self.world_size > 1:
            self.flat.grad.mul_(self.grad_scale)

    def state_dict(self):
        return self.module.state_dict()
