"""hipGraph capture of a fixed-shape forward+backward.

The reference's own workload (the 21,840-parameter MNIST CNN, multinode-DP/multinodeDP.py:77-95,
per-rank batch 16..128) is launch-bound on an MI355X: ~25 kernels of a few microseconds each,
issued from Python through autograd at ~15 us apiece.  Capturing forward + backward once and
replaying the graph removes the host from the loop; gradient all-reduce and the fused optimizer
stay outside the graph (one collective and one kernel per step for this model), so the captured
region contains no communication and works for any world size.

Requirements on the step: static shapes, no host synchronisation, no host-side randomness
(dropout masks come from a device-resident counter, see ``models.mnist._DropState``).
"""
from __future__ import annotations

from typing import Callable, Sequence

import torch


class GraphedStep:
    """``loss = step(*inputs)`` replays the captured forward+backward on new input values.

    ``loss_fn(*static_inputs)`` must return the scalar loss; ``engine`` is the ``DataParallel``
    wrapper (gradients are accumulated locally during capture/replay; call
    ``engine.reduce_all()`` then the optimizer afterwards).  ``after_backward`` runs inside the
    captured region after backward (e.g. the dropout step-counter bump)."""

    def __init__(self, engine, loss_fn: Callable[..., torch.Tensor], example_inputs: Sequence[torch.Tensor],
                 after_backward: Callable[[], None] = None, warmup: int = 3):
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU")
        self.engine = engine
        self.static_inputs = [t.clone() for t in example_inputs]
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm up allocator pools / lazy init off the capture
            for _ in range(warmup):
                with engine.no_sync():
                    loss = loss_fn(*self.static_inputs)
                    loss.backward()
                if after_backward is not None:
                    after_backward()
            # the seed gradient of backward(): allocated once here, not filled by a kernel of its own at every replay
            self._seed = torch.ones_like(loss)
            engine.flat.grad.zero_()                       # warm-up gradients are discarded
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()                # hipGraph on ROCm
        with torch.cuda.graph(self.graph):
            with engine.no_sync():
                self.static_loss = loss_fn(*self.static_inputs)
                self.static_loss.backward(self._seed)
            if after_backward is not None:
                after_backward()
        engine.flat.grad.zero_()                           # capture does not execute, but keep the invariant explicit

    def __call__(self, *inputs: torch.Tensor) -> torch.Tensor:
        for s, t in zip(self.static_inputs, inputs):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        self.graph.replay()
        return self.static_loss
