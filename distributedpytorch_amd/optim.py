"""Fused optimizers over ``FlatParams`` buffers.

``FusedSGD`` is one kernel launch per parameter group per step:

    g' = g * grad_scale + wd * p ;  m = mu * m + g' ;  p -= lr * m ;
    p_bf16 = bf16(p) ;  g = 0

It absorbs the reference's ``param.grad /= world_size`` (multinode-DP/multinodeDP.py:129,
via ``grad_scale``), ``optimizer.zero_grad()`` (DP.py:150) and ``optimizer.step()``
(DP.py:156, SGD lr 0.01 momentum 0.5; MP.py:120,124 lr 0.01 no momentum), and emits
the bf16 shadow the MFMA kernels read.  ``state_dict`` follows ``torch.optim.SGD``
(``state[i]['momentum_buffer']``, ``param_groups``) so checkpoints interoperate.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional

import torch

from .parallel.flat import FlatParams


def _no_decay(name: str, p: torch.Tensor) -> bool:
    return p.dim() <= 1  # biases, BN / LN affine


class FusedSGD:
    def __init__(self, flat: FlatParams, lr: float = 0.01, momentum: float = 0.0, weight_decay: float = 0.0,
                 nesterov: bool = False, no_decay: Optional[Callable[[str, torch.Tensor], bool]] = _no_decay):
        self.flat = flat
        self.lr, self.momentum, self.weight_decay, self.nesterov = lr, momentum, weight_decay, nesterov
        self.grad_scale = 1.0
        self.clip_coef: Optional[torch.Tensor] = None     # device scalar from clip_grad_norm, consumed by the next step()
        self.momentum_buf = torch.zeros_like(flat.master) if momentum != 0.0 else None
        self.steps = 0
        # contiguous runs of the flat buffer sharing the same weight decay -> one launch each
        self.segments = self._segments(no_decay)
        self._seg_ends = self._seg_wd = None
        self._ov = None                    # overlapped mode (overlap_with_backward)

    # ---- optimizer overlapped with backward -------------------------------------------------------------------
    def overlap_with_backward(self, dp, enabled: bool = True) -> bool:
        """Update every gradient bucket's parameters as soon as its reduction has been issued - on a stream of its own,
        ordered behind that reduction - instead of in one launch after backward.  The update is pure HBM traffic
        (26 bytes per parameter: 0.7 ms for GPT-2's 124M) that otherwise runs alone at the end of the step; backward's
        GEMMs leave the bandwidth for it.  Safe because nothing enqueued after a bucket's last gradient reads that
        bucket's fp32 masters or bf16 shadows: input-gradient kernels read the TRANSPOSED twins, which are refreshed
        once, in ``step()``.  Conditions (else this returns False and ``step()`` works as before): GPU, replicated
        optimizer state, a device-ordered reduction handle (RCCL, or a single rank), no conv3 recompute.  The caller
        keeps ``lr`` / ``grad_scale`` fixed from ``backward()`` to ``step()`` and does not use ``clip_grad_norm`` (the
        global norm needs every gradient before any update): ``train.run_dp`` enables this mode only without clipping
        and without an LR schedule, ``bench.py`` for GPT-2 (``--opt-overlap on|off|auto``)."""
        from .ops import block as _block
        ok = (enabled and self.flat.device.type == "cuda" and dp.shards is None and type(self) is FusedSGD
              and (dp.world_size == 1 or getattr(dp.comm, "backend", "") == "nccl")
              and _block.RECOMPUTE_MIN_BYTES > (1 << 50))
        if not ok:
            dp.bucket_hook = None
            self._ov = None
            return False
        self._ov = {"stream": torch.cuda.Stream(device=self.flat.device), "done": [], "tables": {}, "twins": 0}
        dp.bucket_hook = self._step_bucket
        return True

    def _range_args(self, start: int, end: int):
        """Weight-decay table of flat[start:end), relative to start (cached per range)."""
        key = (start, end)
        t = self._ov["tables"].get(key) if self._ov is not None else None
        if t is None:
            segs = [(max(a, start) - start, min(b, end) - start, w) for a, b, w in self.segments if max(a, start) < min(b, end)]
            ends = wds = None
            if len(segs) > 1:
                ends = torch.tensor([b for _, b, _ in segs], dtype=torch.int64, device=self.flat.device)
                wds = torch.tensor([w for _, _, w in segs], dtype=torch.float32, device=self.flat.device)
            t = (segs[0][2], ends, wds)
            if self._ov is not None:
                self._ov["tables"][key] = t
        return t

    @torch.no_grad()
    def _launch_range(self, C, start: int, end: int, gs: float, scale_dev=None) -> None:
        f = self.flat
        wd0, ends, wds = self._range_args(start, end)
        C.sgd_momentum(f.master[start:end], f.grad[start:end],
                       None if self.momentum_buf is None else self.momentum_buf[start:end],
                       None if f.lp is None else f.lp[start:end], self.lr, self.momentum, wd0, gs, True, self.nesterov,
                       ends, wds, scale_dev)

    def _step_bucket(self, b) -> None:
        if self._ov is None or self.clip_coef is not None:
            return
        from .ops import _ext, side
        f = self.flat
        st = self._ov["stream"]
        st.wait_stream(torch.cuda.current_stream(f.device))    # everything enqueued on the compute stream so far
        s = side.for_device(f.device.index if f.device.index is not None else torch.cuda.current_device())
        if s is not None and getattr(s, "stream", None) is not None:
            st.wait_stream(s.stream)                           # ... and the weight-gradient kernels on the side stream
        with torch.cuda.stream(st):
            # the transposed twins of the buckets updated EARLIER: the last kernel of this backward that reads them (the
            # input-gradient GEMM of a bucket's earliest layer, enqueued after that layer's weight-gradient kernel, i.e.
            # possibly after the bucket's hook) is on the compute stream by now, and this stream has just been ordered
            # behind the compute stream
            for a, e in self._ov["done"][self._ov["twins"]:]:
                f.refresh_transposed_range(a, e)
            self._ov["twins"] = len(self._ov["done"])
            if b.work is not None:
                b.work.wait()                                  # the bucket's reduction (device-side wait on RCCL's stream)
            self._launch_range(_ext.load(), b.start, b.end, self.grad_scale)
        self._ov["done"].append((b.start, b.end))

    def _segments(self, no_decay):
        f = self.flat
        runs: List[List[float]] = []
        for pos, idx in enumerate(f.order):
            start = f.offsets[idx]
            end = f.offsets[f.order[pos + 1]] if pos + 1 < len(f.order) else f.total
            wd = 0.0 if (self.weight_decay == 0.0 or (no_decay and no_decay(f.names[idx], f.params[idx]))) else self.weight_decay
            if runs and runs[-1][2] == wd:
                runs[-1][1] = end
            else:
                runs.append([start, end, wd])
        return [(int(a), int(b), float(w)) for a, b, w in runs]

    def zero_grad(self) -> None:
        self.flat.zero_grad()

    @torch.no_grad()
    def step(self, grad_scale: Optional[float] = None, zero_grad: bool = True) -> None:
        f = self.flat
        gs = self.grad_scale if grad_scale is None else grad_scale
        if f.device.type == "cuda":
            from .ops import _ext, side
            side.join(f.device)             # weight gradients may still be in flight on the side stream
            C = _ext.load()
            if self._ov is not None and self._ov["done"]:
                # overlapped mode: most (usually all) buckets were updated during backward; the rest now
                torch.cuda.current_stream(f.device).wait_stream(self._ov["stream"])
                pos = 0
                for a, b in sorted(self._ov["done"]) + [(f.total, f.total)]:
                    if a > pos:
                        self._launch_range(C, pos, a, gs, self.clip_coef)
                        f.refresh_transposed_range(pos, a)
                    pos = max(pos, b)
                for a, b in self._ov["done"][self._ov["twins"]:]:      # buckets updated during backward, twins still old
                    f.refresh_transposed_range(a, b)
                self._ov["done"], self._ov["twins"] = [], 0
                self.clip_coef = None
                self.steps += 1
                return
            if len(self.segments) > 1 and self._seg_ends is None:       # device-side weight-decay table, built once
                self._seg_ends = torch.tensor([b for _, b, _ in self.segments], dtype=torch.int64, device=f.device)
                self._seg_wd = torch.tensor([w for _, _, w in self.segments], dtype=torch.float32, device=f.device)
            # ONE launch over the whole flat buffer
            C.sgd_momentum(f.master, f.grad, self.momentum_buf, f.lp, self.lr, self.momentum, self.segments[0][2], gs,
                           zero_grad, self.nesterov, self._seg_ends, self._seg_wd, self.clip_coef)
            f.refresh_transposed()          # conv filters' [C,R,S,K] twins for the next backward (one launch)
        else:
            if self.clip_coef is not None:
                gs = gs * float(self.clip_coef)
            for a, b, wd in self.segments:
                p, g = f.master[a:b], f.grad[a:b]
                d = g * gs
                if wd != 0.0:
                    d = d + wd * p
                if self.momentum_buf is not None:
                    m = self.momentum_buf[a:b]
                    m.mul_(self.momentum).add_(d)
                    d = d + self.momentum * m if self.nesterov else m
                p.add_(d, alpha=-self.lr)
                if zero_grad:
                    g.zero_()
        self.clip_coef = None
        self.steps += 1

    # ---- torch.optim.SGD-compatible (de)serialisation, parameters in registration order ----
    def state_dict(self) -> Dict:
        f = self.flat
        state = {}
        if self.momentum_buf is not None and self.steps > 0:
            for i, p in enumerate(f.params):
                off = f.offsets[i]
                state[i] = {"momentum_buffer": self.momentum_buf[off:off + p.numel()].view(p.shape).detach().cpu().clone()}
        group = dict(lr=self.lr, momentum=self.momentum, dampening=0, weight_decay=self.weight_decay,
                     nesterov=self.nesterov, maximize=False, foreach=None, differentiable=False, fused=None,
                     params=list(range(len(f.params))))
        return {"state": state, "param_groups": [group], "steps": self.steps}

    def load_state_dict(self, sd: Dict) -> None:
        f = self.flat
        g = sd["param_groups"][0]
        self.lr, self.momentum = g["lr"], g["momentum"]
        self.weight_decay, self.nesterov = g.get("weight_decay", 0.0), g.get("nesterov", False)
        self.steps = sd.get("steps", 1 if sd["state"] else 0)
        if self.momentum != 0.0 and self.momentum_buf is None:
            self.momentum_buf = torch.zeros_like(f.master)
        for i, st in sd["state"].items():
            i = int(i)
            p = f.params[i]
            off = f.offsets[i]
            self.momentum_buf[off:off + p.numel()].copy_(st["momentum_buffer"].reshape(-1).to(f.device))


class ShardedSGD(FusedSGD):
    """SGD-momentum with the optimizer state sharded over the data-parallel ranks (ZeRO stage 1) - for
    ``DataParallel(..., shard_optimizer=True)``.  Rank r holds the momentum of its shard only and runs the fused
    update kernel over that slice of the flat buffers (same kernel, same per-segment weight decay); the updated fp32
    parameters of every shard are then broadcast from their owner and the bf16 shadows re-derived locally.
    ``state_dict()`` gathers the shards so the checkpoint is the same stock ``torch.optim.SGD`` layout as FusedSGD's.
    """

    def __init__(self, dp, lr: float = 0.01, momentum: float = 0.0, weight_decay: float = 0.0, nesterov: bool = False,
                 no_decay: Optional[Callable[[str, torch.Tensor], bool]] = _no_decay):
        if dp.shards is None:
            raise ValueError("ShardedSGD needs DataParallel(..., shard_optimizer=True)")
        self.dp, self.comm = dp, dp.comm
        self.lo, self.hi = dp.shards[dp.comm.rank], dp.shards[dp.comm.rank + 1]
        super().__init__(dp.flat, lr, 0.0, weight_decay, nesterov, no_decay)     # (0.0: no full-size momentum buffer)
        self.momentum = momentum
        self.grad_scale = dp.grad_scale
        if momentum != 0.0:                                    # 1/N of the replicated optimizer's state
            self.momentum_buf = torch.zeros(self.hi - self.lo, dtype=torch.float32, device=self.flat.device)
        # the weight-decay runs clipped to this shard, relative to its start
        self.shard_segments = [(max(a, self.lo) - self.lo, min(b, self.hi) - self.lo, w)
                               for a, b, w in self.segments if max(a, self.lo) < min(b, self.hi)]

    @torch.no_grad()
    def step(self, grad_scale: Optional[float] = None, zero_grad: bool = True) -> None:
        f = self.flat
        gs = self.grad_scale if grad_scale is None else grad_scale
        lo, hi = self.lo, self.hi
        if hi > lo:
            p, g = f.master[lo:hi], f.grad[lo:hi]
            if f.device.type == "cuda":
                from .ops import _ext, side
                side.join(f.device)
                C = _ext.load()
                if len(self.shard_segments) > 1 and self._seg_ends is None:
                    self._seg_ends = torch.tensor([b for _, b, _ in self.shard_segments], dtype=torch.int64, device=f.device)
                    self._seg_wd = torch.tensor([w for _, _, w in self.shard_segments], dtype=torch.float32, device=f.device)
                C.sgd_momentum(p, g, self.momentum_buf, None, self.lr, self.momentum,
                               self.shard_segments[0][2], gs, False, self.nesterov, self._seg_ends, self._seg_wd, self.clip_coef)
            else:
                if self.clip_coef is not None:
                    gs = gs * float(self.clip_coef)
                for a, b, wd in self.shard_segments:
                    d = g[a:b] * gs
                    if wd != 0.0:
                        d = d + wd * p[a:b]
                    if self.momentum_buf is not None:
                        m = self.momentum_buf[a:b]
                        m.mul_(self.momentum).add_(d)
                        d = d + self.momentum * m if self.nesterov else m
                    p[a:b].add_(d, alpha=-self.lr)
        # updated parameters back to everyone: one broadcast per shard, all in flight together
        works = [self.comm.broadcast(f.master[self.dp.shards[r]:self.dp.shards[r + 1]], src=r, async_op=True)
                 for r in range(self.comm.world_size) if self.dp.shards[r + 1] > self.dp.shards[r]]
        if zero_grad:
            f.grad.zero_()
        for w in works:
            w.wait()
        f.refresh_lp()                  # bf16 shadows (and the transposed filter twins) from the gathered masters
        self.clip_coef = None
        self.steps += 1

    def _full_momentum(self) -> Optional[torch.Tensor]:
        if self.momentum_buf is None:
            return None
        full = torch.zeros(self.flat.total, dtype=torch.float32, device=self.flat.device)
        full[self.lo:self.hi].copy_(self.momentum_buf)
        for r in range(self.comm.world_size):
            a, b = self.dp.shards[r], self.dp.shards[r + 1]
            if b > a:
                self.comm.broadcast(full[a:b], src=r)
        return full

    def state_dict(self) -> Dict:
        """Collective: every rank must call it (the momentum shards are gathered)."""
        own = self.momentum_buf
        self.momentum_buf = self._full_momentum()
        try:
            return super().state_dict()
        finally:
            self.momentum_buf = own

    def load_state_dict(self, sd: Dict) -> None:
        own = self.momentum_buf
        self.momentum_buf = torch.zeros(self.flat.total, dtype=torch.float32, device=self.flat.device)
        super().load_state_dict(sd)
        shard = self.momentum_buf[self.lo:self.hi].clone()
        self.momentum_buf = shard if (own is not None or self.momentum != 0.0) else None


class FusedAdamW:
    """AdamW over the flat buffer (used for the GPT-2 configuration when asked)."""

    def __init__(self, flat: FlatParams, lr: float = 3e-4, betas=(0.9, 0.95), eps: float = 1e-8, weight_decay: float = 0.0):
        self.flat, self.lr, self.betas, self.eps, self.weight_decay = flat, lr, betas, eps, weight_decay
        self.m = torch.zeros_like(flat.master)
        self.v = torch.zeros_like(flat.master)
        self.steps = 0
        self.grad_scale = 1.0
        self.clip_coef: Optional[torch.Tensor] = None

    def zero_grad(self):
        self.flat.zero_grad()

    # ---- torch.optim.AdamW-compatible (de)serialisation: state[i] = {step, exp_avg, exp_avg_sq}, registration order ----
    def state_dict(self) -> Dict:
        f = self.flat
        state = {}
        if self.steps > 0:
            for i, p in enumerate(f.params):
                off, n = f.offsets[i], p.numel()
                state[i] = {"step": torch.tensor(float(self.steps)),
                            "exp_avg": self.m[off:off + n].view(p.shape).detach().cpu().clone(),
                            "exp_avg_sq": self.v[off:off + n].view(p.shape).detach().cpu().clone()}
        group = dict(lr=self.lr, betas=tuple(self.betas), eps=self.eps, weight_decay=self.weight_decay, amsgrad=False,
                     maximize=False, foreach=None, capturable=False, differentiable=False, fused=None,
                     params=list(range(len(f.params))))
        return {"state": state, "param_groups": [group], "steps": self.steps}

    def load_state_dict(self, sd: Dict) -> None:
        f = self.flat
        g = sd["param_groups"][0]
        self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        self.weight_decay = g.get("weight_decay", 0.0)
        steps = sd.get("steps")
        for i, st in sd["state"].items():
            i = int(i)
            off, n = f.offsets[i], f.params[i].numel()
            self.m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(f.device))
            self.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(f.device))
            if steps is None:
                steps = int(float(st["step"]))
        self.steps = int(steps or 0)

    @torch.no_grad()
    def step(self, grad_scale: Optional[float] = None, zero_grad: bool = True):
        f = self.flat
        gs = self.grad_scale if grad_scale is None else grad_scale
        self.steps += 1
        b1, b2 = self.betas
        if f.device.type == "cuda":
            from .ops import _ext, side
            side.join(f.device)
            _ext.load().adamw(f.master, f.grad, self.m, self.v, f.lp, self.lr, b1, b2, self.eps, self.weight_decay, gs,
                              self.steps, zero_grad, self.clip_coef)
            f.refresh_transposed()
        else:
            if self.clip_coef is not None:
                gs = gs * float(self.clip_coef)
            g = f.grad * gs
            self.m.mul_(b1).add_(g, alpha=1 - b1)
            self.v.mul_(b2).addcmul_(g, g, value=1 - b2)
            f.master.mul_(1 - self.lr * self.weight_decay)
            mh = self.m / (1 - b1 ** self.steps)
            vh = self.v / (1 - b2 ** self.steps)
            f.master.addcdiv_(mh, vh.sqrt().add_(self.eps), value=-self.lr)
            if zero_grad:
                f.grad.zero_()
        self.clip_coef = None


# ---------------------------------------------------------------------------- training-loop utilities
@torch.no_grad()
def clip_grad_norm(flat: FlatParams, max_norm: float, grad_scale: float = 1.0, shard=None, comm=None,
                   optimizer=None) -> torch.Tensor:
    """Global-norm gradient clipping over the flat gradient buffer, after ``DataParallel.finish()`` and before the
    optimizer step.  The buffer holds the SUM over ranks (the ``1 / world`` lives in the optimizer's ``grad_scale``),
    so pass that scale: the norm returned (a device scalar - nothing here synchronises the host) is the norm of the
    mean gradient.  With a sharded optimizer pass ``shard=(lo, hi)`` and the communicator: each rank holds the reduced
    gradient of its shard only, the squared norms are summed across ranks.  With ``optimizer`` (FusedSGD / ShardedSGD /
    FusedAdamW) the clip coefficient is handed to its next ``step()`` as a device scalar and multiplied into the
    kernel's gradient scale - no pass over the buffer to apply it (GPT-2: 496 MB read + written per step saved);
    without, the buffer is scaled in place.  Not in the reference (constant-lr SGD, DP.py:143); GPT-2 training
    needs it."""
    g = flat.grad if shard is None else flat.grad[shard[0]:shard[1]]
    sq = torch.linalg.vector_norm(g).square() if g.numel() else torch.zeros((), device=flat.grad.device)
    if shard is not None and comm is not None and comm.world_size > 1:
        sq = sq.reshape(1)
        comm.all_reduce(sq)
        sq = sq[0]
    norm = sq.sqrt() * grad_scale
    coef = (max_norm / (norm + 1e-6)).clamp(max=1.0)
    if optimizer is not None and hasattr(optimizer, "clip_coef"):
        optimizer.clip_coef = coef.reshape(1).to(torch.float32)
    else:
        g.mul_(coef)
    return norm


class WarmupCosine:
    """Learning-rate schedule: linear warm-up to ``base_lr`` over ``warmup`` steps, cosine decay to ``min_lr`` at
    ``total`` steps.  ``sched.apply(opt, step)`` sets ``opt.lr`` (the fused kernels take the rate as a launch argument,
    so changing it costs nothing)."""

    def __init__(self, base_lr: float, warmup: int, total: int, min_lr: float = 0.0):
        self.base_lr, self.warmup, self.total, self.min_lr = float(base_lr), int(warmup), int(total), float(min_lr)

    def __call__(self, step: int) -> float:
        import math
        if self.warmup > 0 and step < self.warmup:
            return self.base_lr * (step + 1) / self.warmup
        if step >= self.total:
            return self.min_lr
        t = (step - self.warmup) / max(1, self.total - self.warmup)
        return self.min_lr + 0.5 * (self.base_lr - self.min_lr) * (1.0 + math.cos(math.pi * t))

    def apply(self, opt, step: int) -> float:
        opt.lr = self(step)
        return opt.lr
