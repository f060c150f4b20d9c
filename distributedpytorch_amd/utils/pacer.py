"""Bound on how far the host may run ahead of the GPU.

A training step here is a few hundred kernel launches that the host enqueues in 8-10 ms while the GPU needs 17-64 ms to
run them, and nothing in the step makes the host wait (the loss stays on the device, the optimizer is a kernel).  Left
alone the host ends up a whole benchmark ahead - thousands of launches queued, every step's activations allocated before
the previous steps' were released (reserved memory 130 GB against 64 GB at ResNet-50 / batch 1024) - and on this stack
that is not free: one process in four to eight then runs its steps 3.6 x SLOWER from start to end (ResNet-50 at batch
1024: 231-242 ms per step instead of 64; seen at batch 512 too, less often; never with a host sync per step -
profiles/r3/host_runahead_slow_mode.txt).  ``StepPacer(depth).step()`` at the end of every step records an event and waits
on the host for the event of ``depth`` steps ago: the GPU always has ``depth`` whole steps queued (it never idles for the
host), the queue and the allocator stay shallow, and the slow mode does not occur.

For steps of a fraction of a millisecond (the hipGraph-replayed small CNN) pass ``min_step_ms``: while the measured step is
shorter, only every ``stride``-th step is paced, so the event bookkeeping does not become the step.
"""
from __future__ import annotations

from collections import deque
from typing import Optional

import os
import time

import torch

# How the host waits for the old event.  hipEventSynchronize spins on a core here, with or without the blocking-sync flag
# (host CPU per step = wall time per step with both kinds: profiles/r3/pacer_wait_ab.txt), which is one busy core per rank for
# nothing: the default polls event.query() and sleeps DPA_PACER_POLL_US (200) microseconds in between - the GPU has a whole
# step queued behind the event, so a late wake-up of that size costs nothing.  DPA_PACER_WAIT=sync: the spinning wait.
_WAIT = os.environ.get("DPA_PACER_WAIT", "poll").lower()
_POLL_S = float(os.environ.get("DPA_PACER_POLL_US", "200")) * 1e-6


class StepPacer:
    def __init__(self, device: Optional[torch.device] = None, depth: int = 2, stride: int = 1):
        self.enabled = device is not None and device.type == "cuda" and depth > 0
        self.device, self.depth, self.stride = device, int(depth), max(1, int(stride))
        self._events: deque = deque()
        self._pool: list = []
        self._n = 0
        self.waits = 0

    def step(self) -> None:
        """Call once at the end of every step (after the optimizer has been enqueued)."""
        if not self.enabled:
            return
        self._n += 1
        if self._n % self.stride:
            return
        ev = self._pool.pop() if self._pool else torch.cuda.Event()
        ev.record(torch.cuda.current_stream(self.device))
        self._events.append(ev)
        if len(self._events) > self.depth:
            old = self._events.popleft()
            self._wait(old)                        # host waits; the GPU still has `depth` paced steps queued behind it
            self.waits += 1
            self._pool.append(old)

    @staticmethod
    def _wait(ev) -> None:
        if _WAIT == "sync":
            ev.synchronize()
            return
        while not ev.query():
            time.sleep(_POLL_S)

    def drain(self) -> None:
        while self._events:
            ev = self._events.popleft()
            ev.synchronize()
            self._pool.append(ev)
