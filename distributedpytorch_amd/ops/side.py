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
"""
from __future__ import annotations

import os
from collections import deque
from typing import Callable, Deque, Dict, Sequence, Tuple

import torch

ENABLED = os.environ.get("DPA_WGRAD_STREAM", "1") != "0"

_streams: Dict[int, "torch.cuda.Stream"] = {}
_pending: Deque[Tuple["torch.cuda.Event", int, tuple]] = deque()
_dirty: Dict[int, bool] = {}
_callback_queued = False


def active(t: torch.Tensor) -> bool:
    return ENABLED and t.is_cuda and not torch.cuda.is_current_stream_capturing()


def _stream(index: int) -> "torch.cuda.Stream":
    s = _streams.get(index)
    if s is None:
        s = _streams[index] = torch.cuda.Stream(device=index)
    return s


def run(fn: Callable[[], None], keep: Sequence[torch.Tensor]) -> "torch.cuda.Event":
    """Launch ``fn`` (which enqueues kernels reading ``keep``) on the side stream, after everything enqueued
    on the current stream so far.  Returns the event marking its completion (see ``wait``)."""
    index = keep[0].device.index
    side = _stream(index)
    side.wait_stream(torch.cuda.current_stream(index))
    with torch.cuda.stream(side):
        fn()
        done = torch.cuda.Event()
        done.record(side)
    _pending.append((done, index, tuple(keep)))
    _dirty[index] = True
    global _callback_queued
    if not _callback_queued:                        # whoever reads gradients right after backward() is covered too
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_end_of_backward)
            _callback_queued = True
        except RuntimeError:                        # not inside a backward pass
            pass
    while _pending and _pending[0][0].query():      # drop references the side stream is done with
        _pending.popleft()
    return done


class on_side_stream:
    """Context: the side stream, first made to wait (on the device) for everything enqueued on the current stream."""

    def __init__(self, device):
        self.index = torch.device(device).index if not isinstance(device, int) else device
        if self.index is None:
            self.index = torch.cuda.current_device()
        self.ctx = None

    def __enter__(self):
        side = _stream(self.index)
        side.wait_stream(torch.cuda.current_stream(self.index))
        _dirty[self.index] = True
        self.ctx = torch.cuda.stream(side)
        self.ctx.__enter__()
        return side

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
    """Order the current stream behind all side-stream work enqueued so far (no host block)."""
    if not _dirty:
        return
    index = torch.cuda.current_device() if device is None else torch.device(device).index
    if not _dirty.pop(index, False):
        return
    torch.cuda.current_stream(index).wait_stream(_streams[index])
    # whatever those kernels read may now be recycled by compute-stream allocations: they run after the join
    for _ in range(len(_pending)):
        item = _pending.popleft()
        if item[1] != index:
            _pending.append(item)


def _end_of_backward() -> None:
    global _callback_queued
    _callback_queued = False
    for index in list(_dirty):
        join(torch.device("cuda", index))
