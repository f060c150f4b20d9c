"""Side HIP stream for the weight-gradient kernels.

In backward the chain ``BN-backward -> dgrad -> BN-backward -> ...`` is what the next layer waits for; a
filter gradient is needed by nobody before the bucket all-reduce / the optimizer.  The two are also
complementary on the hardware: BatchNorm backward is HBM-bound, wgrad is MFMA/L2-bound, and late layers'
GEMMs fill only part of the 256 CUs.  So every wgrad that accumulates into the flat gradient buffer is
launched on a second stream, ordered after its operands by an event, and the compute stream only waits for
that stream where the result is consumed (``join()``: before a bucket is handed to RCCL, before the
optimizer).  Operand tensors are kept referenced until the side stream has passed them, so the caching
allocator cannot hand their memory to a later compute-stream kernel.

``DPA_WGRAD_STREAM=0`` disables it (everything on the compute stream, the measured baseline).

State lives in one ``SideStream`` object per DEVICE (the stream, the operands it still reads, whether anything is
in flight).  That is deliberately not per engine: stream order is a property of the device queue, ``join`` has to
cover every producer of a gradient whichever engine launched it, and two engines on one GPU gain nothing from two
side queues (HIP maps them onto the same few hardware queues).  What IS per engine - where a gradient lands and
who is told it is ready - lives in ``ops/sink.py``'s ``GradSink`` objects.
"""
from __future__ import annotations

import os
from collections import deque
from typing import Callable, Deque, Dict, List, Sequence, Tuple

import torch

ENABLED = os.environ.get("DPA_WGRAD_STREAM", "1") != "0"

# Producers that HOLD BACK side-stream launches (ops/wgroup.py collects the weight gradients of several layers for one
# launch) register a flush here: it runs, with the device index, before anybody joins or borrows the side stream - i.e.
# before any consumer of a gradient (bucket all-reduce, optimizer, end of backward) can get ahead of a held-back launch.
_flush_hooks: List[Callable[[int], None]] = []


def add_flush_hook(fn: Callable[[int], None]) -> None:
    if fn not in _flush_hooks:
        _flush_hooks.append(fn)


def _run_flush_hooks(index: int) -> None:
    for fn in _flush_hooks:
        fn(index)


class SideStream:
    """The side queue of one device."""

    def __init__(self, index: int):
        self.index = index
        self.stream = torch.cuda.Stream(device=index)
        self.pending: Deque[Tuple["torch.cuda.Event", tuple]] = deque()   # (done, operands kept alive)
        self.dirty = False
        self.callback_queued = False

    def run(self, fn: Callable[[], None], keep: Sequence[torch.Tensor]) -> "torch.cuda.Event":
        self.stream.wait_stream(torch.cuda.current_stream(self.index))
        with torch.cuda.stream(self.stream):
            fn()
            done = torch.cuda.Event()
            done.record(self.stream)
        self.pending.append((done, tuple(keep)))
        self.dirty = True
        if not self.callback_queued:                    # whoever reads gradients right after backward() is covered too
            try:
                torch.autograd.Variable._execution_engine.queue_callback(self._end_of_backward)
                self.callback_queued = True
            except RuntimeError:                        # not inside a backward pass
                pass
        while self.pending and self.pending[0][0].query():      # drop references the side stream is done with
            self.pending.popleft()
        return done

    def join(self) -> None:
        _run_flush_hooks(self.index)
        if not self.dirty:
            return
        self.dirty = False
        torch.cuda.current_stream(self.index).wait_stream(self.stream)
        self.pending.clear()    # whatever those kernels read may be recycled by compute-stream allocations: they run after the join

    def _end_of_backward(self) -> None:
        self.callback_queued = False
        self.join()


_per_device: Dict[int, SideStream] = {}


def for_device(index: int) -> SideStream:
    s = _per_device.get(index)
    if s is None:
        s = _per_device[index] = SideStream(index)
    return s


def active(t: torch.Tensor) -> bool:
    return ENABLED and t.is_cuda and not torch.cuda.is_current_stream_capturing()


def run(fn: Callable[[], None], keep: Sequence[torch.Tensor]) -> "torch.cuda.Event":
    """Launch ``fn`` (which enqueues kernels reading ``keep``) on the side stream of ``keep[0]``'s device, after
    everything enqueued on the current stream so far.  Returns the event marking its completion (see ``wait``)."""
    return for_device(keep[0].device.index).run(fn, keep)


class on_side_stream:
    """Context: the side stream, first made to wait (on the device) for everything enqueued on the current stream."""

    def __init__(self, device):
        self.index = torch.device(device).index if not isinstance(device, int) else device
        if self.index is None:
            self.index = torch.cuda.current_device()
        self.ctx = None

    def __enter__(self):
        _run_flush_hooks(self.index)
        side = for_device(self.index)
        side.stream.wait_stream(torch.cuda.current_stream(self.index))
        side.dirty = True
        self.ctx = torch.cuda.stream(side.stream)
        self.ctx.__enter__()
        return side.stream

    def __exit__(self, *exc):
        return self.ctx.__exit__(*exc)


def wait(done: "torch.cuda.Event", produced: Sequence[torch.Tensor] = ()) -> None:
    """Order the current stream behind one side launch.  ``produced`` are tensors that launch ALLOCATED (they
    belong to the side stream's allocator pool): they are marked as used by the current stream so that their
    memory is not recycled under the compute-stream kernels that read them."""
    cur = torch.cuda.current_stream()
    cur.wait_event(done)
    for t in produced:
        if t is not None:
            t.record_stream(cur)


def join(device=None) -> None:
    """Order the current stream behind all side-stream work enqueued so far on ``device`` (no host block)."""
    index = torch.cuda.current_device() if device is None else torch.device(device).index
    if index is None:
        index = torch.cuda.current_device()
    s = _per_device.get(index)
    if s is not None:
        s.join()
    else:
        _run_flush_hooks(index)
