"""Keep Python's cyclic garbage collector out of the synchronous-SGD step.

One rank per GPU and a gradient all-reduce per step means every rank waits for the slowest one, every step.  A
generation-2 collection is a 10-50 ms host pause on ONE rank (the autograd graph of a ResNet-152 step is tens of
thousands of objects) and the thresholds trip at different steps on different ranks, so at N ranks the job pays
about N times as many pauses as a single process.  ``GCGuard`` turns the automatic collector off and collects at a
fixed step period instead - the same step on every rank, so the pauses coincide.  Objects alive when the guard is
entered (modules, parameters, buffers) are frozen out of future scans.

The reference (multinode-DP/multinodeDP.py:147-156) leaves the collector alone; its 8-rank job pays this jitter.
"""
from __future__ import annotations

import gc


class GCGuard:
    def __init__(self, period: int = 500, freeze: bool = True):
        self.period, self.freeze = int(period), freeze
        self._was_enabled = False
        self._steps = 0

    def __enter__(self) -> "GCGuard":
        self._was_enabled = gc.isenabled()
        gc.collect()
        if self.freeze:
            gc.freeze()
        gc.disable()
        self._steps = 0
        return self

    def step(self) -> None:
        """Call once per training step, on every rank."""
        self._steps += 1
        if self.period > 0 and self._steps % self.period == 0:
            gc.collect(1)

    def __exit__(self, *exc) -> None:
        if self.freeze:
            gc.unfreeze()
        if self._was_enabled:
            gc.enable()
