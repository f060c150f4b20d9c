"""Metrics / logging / tracing.

The reference only prints a per-epoch loss (multinode-DP/multinodeDP.py:157: every rank,
mean over batches; multinode-MP/multinodeMP.py:169-170: last stage, last batch).  Those two
lines are kept verbatim in shape (``epoch_line_dp`` / ``epoch_line_mp``); added on top:
images/sec (per rank and whole job), a phase breakdown from HIP events (forward,
backward, exposed communication, optimizer), JSON-lines output (so results survive the
gpurun round trip), optional roctx ranges for rocprofv3 timelines, and the
``DPA_SYNC_DEBUG`` cross-stream checker.
"""
from __future__ import annotations

import json
import os
import time
from contextlib import contextmanager
from typing import Dict, List, Optional

import torch


def epoch_line_dp(rank: int, epoch: int, mean_loss) -> str:
    """Reference format: ``print('Rank ', rank, ' , epoch ', epoch, ' : ', loss)`` (DP.py:157)."""
    return f"Rank  {rank}  , epoch  {epoch}  :  {float(mean_loss):.6f}"


def epoch_line_mp(epoch: int, loss) -> str:
    """Reference format: ``print('epoch ', epoch, ' : ', loss)`` (MP.py:170)."""
    return f"epoch  {epoch}  :  {float(loss):.6f}"


class ThroughputMeter:
    """Items/sec over a window; only synchronises when read."""

    def __init__(self, device: Optional[torch.device] = None):
        self.device = device
        self.reset()

    def reset(self):
        self.items = 0
        self.steps = 0
        self._t0 = None

    def _sync(self):
        if self.device is not None and self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def start(self):
        self._sync()
        self._t0 = time.perf_counter()

    def update(self, items: int):
        if self._t0 is None:
            self.start()
        self.items += items
        self.steps += 1

    def rate(self) -> float:
        self._sync()
        dt = time.perf_counter() - (self._t0 or time.perf_counter())
        return self.items / dt if dt > 0 else 0.0

    def ms_per_step(self) -> float:
        self._sync()
        dt = time.perf_counter() - (self._t0 or time.perf_counter())
        return 1000.0 * dt / max(self.steps, 1)


class PhaseTimer:
    """HIP-event (or wall-clock on CPU) timers around named phases of a step.  Events are
    recorded on the current stream and only read back in ``summary`` - no host sync in
    the step."""

    def __init__(self, device: Optional[torch.device] = None, enabled: bool = True):
        self.cuda = device is not None and device.type == "cuda"
        self.enabled = enabled
        self._open: Dict[str, object] = {}
        self._pairs: Dict[str, List] = {}

    @contextmanager
    def phase(self, name: str):
        if not self.enabled:
            yield
            return
        rng = _roctx_push(name)
        if self.cuda:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            try:
                yield
            finally:
                e.record()
                self._pairs.setdefault(name, []).append((s, e))
                _roctx_pop(rng)
        else:
            t0 = time.perf_counter()
            try:
                yield
            finally:
                self._pairs.setdefault(name, []).append(time.perf_counter() - t0)
                _roctx_pop(rng)

    def summary_with_counts(self, reset: bool = True):
        """(mean milliseconds per occurrence, number of occurrences) of each phase."""
        counts = {name: len(vals) for name, vals in self._pairs.items() if vals}
        return self.summary(reset), counts

    def summary(self, reset: bool = True) -> Dict[str, float]:
        """Mean milliseconds per occurrence of each phase."""
        out = {}
        if self.cuda:
            torch.cuda.synchronize()
        for name, vals in self._pairs.items():
            if not vals:
                continue
            if self.cuda:
                ms = [s.elapsed_time(e) for s, e in vals]
            else:
                ms = [v * 1000.0 for v in vals]
            out[name] = sum(ms) / len(ms)
        if reset:
            self._pairs = {}
        return out


class JsonlLogger:
    def __init__(self, path: Optional[str], rank: int = 0, only_rank0: bool = True):
        self.path = path if (path and (rank == 0 or not only_rank0)) else None
        self.rank = rank
        if self.path:
            os.makedirs(os.path.dirname(os.path.abspath(self.path)) or ".", exist_ok=True)

    def log(self, **fields):
        if not self.path:
            return
        fields.setdefault("ts", time.time())
        fields.setdefault("rank", self.rank)
        with open(self.path, "a") as f:
            f.write(json.dumps(fields) + "\n")


def scaling_efficiency(rate_n: float, rate_1: float, n: int) -> float:
    """Weak-scaling efficiency = T(N) / (N * T(1)) with T = throughput."""
    return rate_n / (n * rate_1) if rate_1 > 0 and n > 0 else 0.0


# ---- roctx ranges (visible in rocprofv3 --marker-trace timelines); no-ops when unavailable ----
_roctx = None


def _load_roctx():
    global _roctx
    if _roctx is not None:
        return _roctx
    _roctx = False
    if os.environ.get("DPA_ROCTX", "0") != "1":
        return _roctx
    try:
        import ctypes
        for name in ("libroctx64.so", "libroctx64.so.4"):
            try:
                _roctx = ctypes.CDLL(name)
                break
            except OSError:
                continue
    except Exception:  # noqa: BLE001
        _roctx = False
    return _roctx


def _roctx_push(name: str):
    lib = _load_roctx()
    if lib:
        lib.roctxRangePushA(name.encode())
        return True
    return None


def _roctx_pop(token):
    if token:
        _load_roctx().roctxRangePop()


# ---- DPA_SYNC_DEBUG: serial recomputation check of the reduced gradients ----
def check_reduced_gradients(flat, comm, atol: float = 1e-5) -> float:
    """Debug helper for stream-ordering bugs: after ``DataParallel.finish()`` every rank
    must hold bit-comparable gradients (they are the same all-reduce result).  Returns
    the max absolute deviation from rank 0's buffer; raises when it exceeds ``atol``."""
    g = flat.grad.detach().clone()
    ref = g.clone()
    comm.broadcast(ref, src=0)
    dev = float((g - ref).abs().max().item()) if g.numel() else 0.0
    worst = comm.all_reduce_scalar(dev, "max")
    if worst > atol:
        raise RuntimeError(f"reduced gradients differ across ranks by {worst:g} (> {atol:g}): "
                           "a bucket was reduced before its gradients were written, or consumed before the reduce finished")
    return worst
