"""A stand-in for csrc/rccl_comm.cpp's ``RcclComm`` that moves the data over a c10d gloo group.

Why: everything in comm.py / parallel/ that runs on the native layer AT WORLD SIZE > 1 - sub-communicators made by
``ncclCommSplit`` (build_grid), ZeRO-1's reduce / broadcast, the pipeline engine's grouped send / recv, the start-up
self-test's collective phases - can only execute for real on a box with several GPUs, which the builder of this repo never
had.  With ``DPA_FAKE_NATIVE=1`` ``comm.init_process_group`` installs this class as ``Communicator.native`` under a gloo
process group, so the multi-process tests drive exactly that Python code (same methods, same tickets, same rank
conventions: peers and roots are ranks OF THE COMMUNICATOR) on the CPU or with every rank on one GPU.  It says nothing about
RCCL itself: tests/test_multi_gpu.py does that where the hardware exists.

Semantics kept from the C++ class: every call returns a ticket at once, ``wait`` / ``query`` / ``synchronize`` exist,
``split`` is collective over the parent and returns None for a negative colour (NCCL_SPLIT_NOCOLOR), device tensors are
taken as they are (staged through the host here).  Not kept: ncclSend / ncclRecv's rendezvous semantics (gloo buffers) -
``parallel/schedule.validate`` models those.
"""
from __future__ import annotations

from typing import List, Optional, Sequence

import torch
import torch.distributed as dist


class GlooNative:
    accepts_host = True                 # comm.Communicator._on_native lets CPU tensors through to this layer

    def __init__(self, group=None, ranks: Optional[Sequence[int]] = None):
        self.group = group              # c10d gloo group over the same ranks (None = the default group)
        self.ranks = list(ranks) if ranks is not None else list(range(dist.get_world_size()))
        self.rank = self.ranks.index(dist.get_rank())
        self.world = len(self.ranks)
        self.calls: List[str] = []      # what ran here, for the tests
        self._ticket = 0
        self.aborted = False

    # ---- plumbing ----
    def _g(self, r: int) -> int:
        return self.ranks[r]

    def _done(self, what: str) -> int:
        self.calls.append(what)
        self._ticket += 1
        return self._ticket

    @staticmethod
    def _host(t: torch.Tensor) -> torch.Tensor:
        return t.detach().cpu() if t.is_cuda else t.detach()

    @staticmethod
    def _back(t: torch.Tensor, h: torch.Tensor) -> None:
        if t.is_cuda:
            t.copy_(h)

    @staticmethod
    def version() -> int:
        return 0

    def wait(self, ticket: int) -> None:
        return None

    def query(self, ticket: int) -> bool:
        return True

    def synchronize(self) -> None:
        return None

    def abort(self) -> None:
        self.aborted = True

    # ---- collectives (roots / peers are ranks of THIS communicator) ----
    def all_reduce(self, t: torch.Tensor, op: int = 0) -> int:
        h = self._host(t)
        wide = h.float() if h.dtype == torch.bfloat16 else h              # gloo has no bf16 arithmetic
        dist.all_reduce(wide, op=(dist.ReduceOp.SUM, dist.ReduceOp.MAX, dist.ReduceOp.MIN)[op], group=self.group)
        if wide is not h:
            h.copy_(wide)
        self._back(t, h)
        return self._done("all_reduce")

    def all_reduce_direct(self, t: torch.Tensor, scratch: torch.Tensor) -> int:
        self.all_reduce(t, 0)
        self.calls[-1] = "all_reduce_direct"
        return self._ticket

    def reduce(self, t: torch.Tensor, dst: int) -> int:
        h = self._host(t)
        dist.reduce(h, dst=self._g(dst), group=self.group)
        if self.rank == dst:
            self._back(t, h)
        return self._done("reduce")

    def broadcast(self, t: torch.Tensor, src: int) -> int:
        h = self._host(t)
        dist.broadcast(h, src=self._g(src), group=self.group)
        self._back(t, h)
        return self._done("broadcast")

    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor) -> int:
        h = self._host(inp).clone()
        dist.all_reduce(h, group=self.group)                              # gloo has no reduce_scatter
        n = out.numel()
        out.copy_(h.view(-1)[self.rank * n:(self.rank + 1) * n].view(out.shape))
        return self._done("reduce_scatter")

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor) -> int:
        h = self._host(inp).contiguous()
        parts = [torch.empty_like(h) for _ in range(self.world)]
        dist.all_gather(parts, h, group=self.group)
        out.copy_(torch.cat([p.view(-1) for p in parts]).view(out.shape))
        return self._done("all_gather")

    def batch_p2p(self, ops) -> int:
        """[(is_send, tensor, peer), ...] as one group: every transfer posted before any is waited for."""
        posted = []
        for is_send, t, peer in ops:
            if is_send:
                posted.append((dist.isend(self._host(t).contiguous(), dst=self._g(peer), group=self.group), None, None))
            else:
                h = torch.empty(t.shape, dtype=t.dtype)
                posted.append((dist.irecv(h, src=self._g(peer), group=self.group), t, h))
        for work, t, h in posted:
            work.wait()
            if t is not None:
                t.copy_(h)
        return self._done("batch_p2p")

    def split(self, color: int, key: int, new_rank: int, new_world: int):
        """Collective over this communicator, which must span the default group (dist.new_group is collective over the
        world): one gloo group per colour, ranks ordered by key as ncclCommSplit orders them."""
        if len(self.ranks) != dist.get_world_size():
            raise RuntimeError("GlooNative.split: only the world communicator can be split (dist.new_group is world-collective)")
        mine = (int(color), int(key), dist.get_rank())
        everyone = [None] * self.world
        dist.all_gather_object(everyone, mine, group=self.group)
        self._done("split")
        sub = None
        for c in sorted({c for c, _, _ in everyone if c >= 0}):
            members = [g for _, _, g in sorted((k, i, g) for i, (cc, k, g) in enumerate(everyone) if cc == c)]
            group = dist.new_group(ranks=members, backend="gloo")
            if c == color:
                sub = GlooNative(group, members)
                if sub.rank != new_rank or sub.world != new_world:
                    raise RuntimeError(f"GlooNative.split: caller expected rank {new_rank} of {new_world}, the split gives "
                                       f"{sub.rank} of {sub.world}")
        return sub
