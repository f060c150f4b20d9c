"""Model-parallel (pipeline) engine: layer stages on different GPUs, device-to-device
send/recv of activations and gradients over RCCL, micro-batched schedules.

What it generalises (reference multinode-MP/multinodeMP.py:118-167): stage k lives on
rank k; forward sends the stage's output to k+1; backward receives d(loss)/d(output),
runs ``output.backward(grad)`` and sends ``input.grad`` to k-1; every stage has its own
optimizer; the autograd graph is cut at the stage boundary (``.data`` at MP.py:139) and
re-rooted on a fresh leaf (MP.py:155); the ragged last batch works because the batch
dimension is taken from the data (MP.py:143,152).

What is different: tensors never touch the host (the reference stages every transfer
through CPU memory because gloo needs it, MP.py:139,145,155,166); any number of stages;
the batch is split into micro-batches and scheduled GPipe- or 1F1B-style so stages
overlap instead of idling (the reference is strictly serial: <= 1/S utilisation);
sends and receives that are adjacent in the schedule go out as one grouped RCCL call;
stages can be replicated (DP x PP) with the data-parallel engine reducing each stage's
gradients inside its replica group.
"""
from __future__ import annotations

import os
import time
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn as nn

from .. import comm as _comm
from . import schedule as _sched

_DTYPES = [torch.float32, torch.bfloat16, torch.float16, torch.int64, torch.float64]


# ---------------------------------------------------------------------------- partitioning
def partition_balanced(costs: Sequence[float], n_stages: int) -> List[Tuple[int, int]]:
    """Split ``len(costs)`` consecutive layers into ``n_stages`` contiguous groups minimising
    the heaviest group (classic linear-partition DP).  Returns [(start, end), ...]."""
    n = len(costs)
    if n_stages < 1 or n_stages > n:
        raise ValueError(f"cannot split {n} layers into {n_stages} stages")
    prefix = [0.0]
    for c in costs:
        prefix.append(prefix[-1] + float(c))
    INF = float("inf")
    best = [[INF] * (n + 1) for _ in range(n_stages + 1)]
    cut = [[0] * (n + 1) for _ in range(n_stages + 1)]
    best[0][0] = 0.0
    for k in range(1, n_stages + 1):
        for i in range(k, n + 1):
            for j in range(k - 1, i):
                v = max(best[k - 1][j], prefix[i] - prefix[j])
                if v < best[k][i]:
                    best[k][i], cut[k][i] = v, j
    bounds = []
    i = n
    for k in range(n_stages, 0, -1):
        j = cut[k][i]
        bounds.append((j, i))
        i = j
    return bounds[::-1]


def param_costs(layers: Sequence[nn.Module]) -> List[float]:
    return [1.0 + sum(p.numel() for p in l.parameters()) for l in layers]


class _PreserveState:
    """Train-mode passes over the REAL layers (timing, shape tracing) put back what they touch besides gradients:
    BatchNorm running statistics / batch counters and the positions of the counter-based dropout streams - measuring
    must not change the run that follows."""

    def __init__(self, layers: Sequence[nn.Module]):
        self.layers = layers

    def __enter__(self):
        self.bufs = [(b, b.detach().clone()) for layer in self.layers for b in layer.buffers()]
        self.drop = [(m._drop, {k: getattr(m._drop, k) for k in ("calls", "step") if hasattr(m._drop, k)})
                     for layer in self.layers for m in layer.modules() if getattr(m, "_drop", None) is not None]
        return self

    def __exit__(self, *exc):
        with torch.no_grad():
            for b, keep in self.bufs:
                b.copy_(keep)
        for drop, fields in self.drop:
            for k, v in fields.items():
                setattr(drop, k, v)
        return False


Meta = Tuple[Tuple[int, ...], torch.dtype]      # trailing shape (without the batch axis) and dtype of a stage boundary


def measure_costs(layers: Sequence[nn.Module], sample: torch.Tensor, iters: int = 3,
                  metas: Optional[List[Meta]] = None) -> List[float]:
    """Forward+backward time of every layer on this device (stages must be balanced by
    TIME, not by layer count: early ResNet stages move far more activation bytes).  ``metas`` (a list) receives the
    trailing shape and dtype of every layer's OUTPUT: what a stage needs to post its receives without any wire
    negotiation (``PipelineStage(in_meta=...)``)."""
    costs = []
    x = sample
    use_cuda = sample.is_cuda

    def once(layer, x, g):
        xin = x.detach().requires_grad_(x.is_floating_point())
        y = layer(xin)
        y.backward(torch.ones_like(y) if g is None else g)
        return y

    with _PreserveState(layers):
        for layer in layers:
            # one untimed forward AND backward first: the first launch of a kernel loads its code object (milliseconds),
            # and whichever layer meets a kernel first would be charged for it - a split measured that way put 49 of
            # ResNet-152's 52 layers on the last of 4 stages
            y = once(layer, x, None)
            g = torch.ones_like(y)
            best = float("inf")
            for _ in range(max(1, iters)):
                if use_cuda:
                    torch.cuda.synchronize()
                t0 = time.perf_counter()
                once(layer, x, g)
                if use_cuda:
                    torch.cuda.synchronize()
                best = min(best, time.perf_counter() - t0)      # min, not mean: a neighbour's burst or a host hiccup only adds
            costs.append(best)
            if metas is not None:
                metas.append((tuple(int(d) for d in y.shape[1:]), y.dtype))
            x = y.detach()
            for p in layer.parameters():
                p.grad = None
    return costs


def trace_shapes(layers: Sequence[nn.Module], sample: torch.Tensor) -> List[Meta]:
    """Trailing shape + dtype of every layer's output, from one no-grad pass of a small sample (where nothing is
    timed: the CPU path, whose stages are cut by parameter count)."""
    metas: List[Meta] = []
    x = sample
    with _PreserveState(layers), torch.no_grad():
        for layer in layers:
            x = layer(x)
            metas.append((tuple(int(d) for d in x.shape[1:]), x.dtype))
    return metas


def stage_in_metas(metas: Sequence[Meta], bounds: Sequence[Tuple[int, int]]) -> List[Optional[Meta]]:
    """Per stage: the meta of what it receives (None for stage 0) = the output of the layer before its first one."""
    return [None if a == 0 else (tuple(metas[a - 1][0]), metas[a - 1][1]) for a, _ in bounds]


def split_microbatches(n: int, n_micro: int) -> List[int]:
    """Sizes of the micro-batches of a batch of ``n`` samples (ragged batches allowed;
    empty micro-batches are dropped, so a batch of 3 with 8 micro-batches runs 3)."""
    n_micro = max(1, min(n_micro, n))
    base, rem = divmod(n, n_micro)
    return [base + (1 if i < rem else 0) for i in range(n_micro)]


# ---------------------------------------------------------------------------- engine
class _SlotPool:
    """Pre-allocated receive slots.  A receive takes a slot of its (shape, dtype); the slot goes back when the
    micro-batch it carried has been consumed (its backward has run / its gradient has been sent on), so in steady
    state no receive allocates: a stage holds as many slots per direction as micro-batches are live on it
    (``schedule.peak_live_microbatches``: 2 for the serial schedule of the reference, S - stage under 1F1B).
    Re-use is ordered on the device: the communicator makes its stream wait for the compute stream before every
    grouped call, so a slot is never overwritten under a kernel that still reads it.

    Receives are issued in schedule order on the ONE communication stream, grouped with the send that is adjacent in
    the schedule (``schedule.group_comm``); posting a receive ahead of the compute before it would queue it in front of
    this stage's own next send on that stream - a chain of stages each waiting to receive before it may send."""

    def __init__(self):
        self._free: Dict[Tuple, List[torch.Tensor]] = {}
        self.allocated = 0

    def take(self, shape: Tuple[int, ...], dtype: torch.dtype, device) -> torch.Tensor:
        free = self._free.get((tuple(shape), dtype))
        if free:
            return free.pop()
        self.allocated += 1
        return torch.empty(tuple(shape), dtype=dtype, device=device)

    def give(self, t: Optional[torch.Tensor]) -> None:
        if t is None or not getattr(t, "_dpa_slot", False):
            return
        if t.requires_grad:
            t.grad = None
            t.requires_grad_(False)
        self._free.setdefault((tuple(t.shape), t.dtype), []).append(t)


class PipelineStage:
    """One stage of the pipeline, bound to this rank.

    ``comm``      pipeline-group communicator (rank inside it == stage index)
    ``module``    this stage's layers
    ``loss_fn``   ``(output, target) -> scalar`` (mean over the micro-batch); last stage only
    ``dp``        optional ``DataParallel`` engine over the stage's replica group
    """

    def __init__(self, module: nn.Module, comm: _comm.Communicator, loss_fn: Optional[Callable] = None,
                 n_micro: int = 1, schedule: str = "1f1b", dp=None, send_dtype: Optional[torch.dtype] = None,
                 in_meta: Optional[Meta] = None, negotiate: bool = True, tied_wire: Optional[str] = None):
        """``negotiate`` (the same on EVERY stage of a pipeline): True = each stage learns the trailing shape and dtype
        of its input from one small transfer in front of the first activation; False = nothing is negotiated on the
        wire, as in the reference (both of its ranks know ``[B, 320]`` fp32, MP.py:143,152): every receiving stage is
        given ``in_meta`` by the caller (``measure_costs`` / ``trace_shapes`` + ``stage_in_metas``)."""
        self.module, self.comm, self.loss_fn, self.dp = module, comm, loss_fn, dp
        self.stage, self.n_stages = comm.rank, comm.world_size
        self.n_micro, self.schedule = n_micro, schedule
        self.first, self.last = self.stage == 0, self.stage == self.n_stages - 1
        self.send_dtype = send_dtype
        self.device = next(module.parameters()).device
        self._negotiate = bool(negotiate)
        if not self._negotiate and not self.first and in_meta is None:
            raise ValueError("PipelineStage(negotiate=False): a receiving stage needs in_meta")
        self._in_meta: Optional[Meta] = (tuple(in_meta[0]), in_meta[1]) if (in_meta is not None and not self._negotiate) else None
        self._out_meta_sent = False
        self._pending_sends: List = []
        self._slots = _SlotPool()
        self.last_loss: Optional[torch.Tensor] = None
        self.timer = None                      # metrics.PhaseTimer: forward / backward / p2p phases per micro-batch
        self._tied_wire = (tied_wire or os.environ.get("DPA_TIED_WIRE", "auto")).lower()
        self._tied_buf: Dict[str, torch.Tensor] = {}
        self.peak_live = _sched.peak_live_microbatches(schedule, self.n_stages, self.stage, max(1, n_micro))
        # parameters tied across the first and the last stage (GPT-2's wte: embedding + LM head).  A stage holding only
        # one end exchanges that parameter's gradient with the other end after backward (see _sync_tied)
        ends: Dict[str, Dict[str, nn.Parameter]] = {}
        for mod in module.modules():
            tie = getattr(mod, "_dpa_tie", None)
            if tie is not None:
                ends.setdefault(tie[0], {})[tie[1]] = getattr(mod, tie[0])
        self._tied = [(name, roles["first" if "first" in roles else "last"], "first" if "first" in roles else "last")
                      for name, roles in sorted(ends.items()) if len(roles) == 1 and self.n_stages > 1]

    # ---- shape negotiation (only when the caller gave no ``in_meta``): once, trailing dims only ----
    def _send_meta(self, y: torch.Tensor) -> None:
        meta = torch.zeros(8, dtype=torch.int64)
        trailing = list(y.shape[1:])
        meta[0] = len(trailing)
        for i, d in enumerate(trailing):
            meta[1 + i] = d
        meta[7] = _DTYPES.index(y.dtype if self.send_dtype is None else self.send_dtype)
        buf = meta.to(self.device)
        for w in self.comm.batch_p2p([("send", buf, self.stage + 1)]):
            w.wait()
        self._out_meta_sent = True

    def _recv_meta(self) -> None:
        meta = torch.zeros(8, dtype=torch.int64, device=self.device)
        for w in self.comm.batch_p2p([("recv", meta, self.stage - 1)]):
            w.wait()
        meta = meta.cpu()                      # one host read, once per run
        nd = int(meta[0])
        self._in_meta = (tuple(int(v) for v in meta[1:1 + nd]), _DTYPES[int(meta[7])])

    def _phase(self, name: str):
        return self.timer.phase(name) if self.timer is not None else _NullCtx()

    # ---- one training step over a (possibly ragged) batch ----
    def train_step(self, batch_size: int, x: Optional[torch.Tensor] = None, target: Optional[torch.Tensor] = None):
        """``x`` is needed on the first stage, ``target`` on the last; every stage passes
        the same ``batch_size`` (each rank walks its own loader, as in the reference).
        Returns the mean loss over the batch on the last stage, else None.  Gradients are
        accumulated into ``.grad``; the caller steps its own optimizer afterwards."""
        sizes = split_microbatches(batch_size, self.n_micro)
        n_micro = len(sizes)
        offs = [0]
        for s in sizes:
            offs.append(offs[-1] + s)
        # compute operations alone, communication in groups that are deadlock-free under rendezvous semantics on one
        # in-order stream per rank, even if whole groups had to pair up (schedule.validate_rendezvous, strict)
        steps = _sched.build_steps(self.schedule, self.n_stages, self.stage, n_micro)
        inputs: Dict[int, torch.Tensor] = {}
        outputs: Dict[int, torch.Tensor] = {}
        grads_in: Dict[int, torch.Tensor] = {}
        loss_sum = None
        n_b_done = 0
        if self._negotiate and not self.first and self._in_meta is None:
            self._recv_meta()
        send_meta = (not self.last) and (not self._out_meta_sent) and self._sends_meta()

        for step in steps:
            kind = step[0][0]
            if kind in _sched.COMM_OPS:
                with self._phase("p2p"):
                    self._run_comm(step, sizes, inputs, outputs, grads_in)
                continue
            op, mb = step[0]
            if op == "F":
                with self._phase("forward"):
                    if self.first:
                        xin = x[offs[mb]:offs[mb + 1]]
                    else:
                        xin = inputs[mb]
                        if xin.is_floating_point():
                            xin.requires_grad_(True)
                        inputs[mb] = xin
                    y = self.module(xin)
                    if self.last:
                        t = target[offs[mb]:offs[mb + 1]]
                        loss = self.loss_fn(y, t) * (sizes[mb] / float(batch_size))
                        outputs[mb] = loss
                        loss_sum = loss.detach() if loss_sum is None else loss_sum + loss.detach()
                    else:
                        outputs[mb] = y
                if send_meta:
                    self._send_meta(y)
                    send_meta = False
            else:  # "B"
                n_b_done += 1
                sync_now = n_b_done == n_micro
                ctx = self.dp.no_sync() if (self.dp is not None and not sync_now) else _NullCtx()
                with self._phase("backward"), ctx:
                    if self.last:
                        outputs.pop(mb).backward()
                    else:
                        g_in = grads_in.pop(mb)
                        outputs.pop(mb).backward(g_in)
                        self._slots.give(g_in)
                if self.first:
                    inputs.pop(mb, None)
        with self._phase("p2p"):
            for w in self._pending_sends:
                w[0].wait()
            self._pending_sends.clear()
        if self.dp is not None:
            with self._phase("dp_exposed"):
                self.dp.finish()
        with self._phase("tied_sync"):
            self._sync_tied()
        self.last_loss = loss_sum
        return loss_sum

    def _sends_meta(self) -> bool:
        return self._negotiate

    def _tied_wire_dtype(self, g: torch.Tensor) -> torch.dtype:
        if self._tied_wire in ("bf16", "bfloat16"):
            return torch.bfloat16
        if self._tied_wire in ("fp32", "float32"):
            return torch.float32
        # auto: bf16 on the links when the model computes in bf16 on a GPU (the gradient of a bf16 GEMM carries no more
        # than that), the parameter's own dtype elsewhere (CPU oracles compare to fp32 rounding)
        return torch.bfloat16 if (g.is_cuda and g.dtype == torch.float32) else g.dtype

    def _sync_tied(self) -> None:
        """Sum the gradients of a parameter whose two uses sit on the first and the last stage (Megatron's embedding
        all-reduce, as two transfers over the pipeline group: last -> first, sum, first -> last).  After the replica
        all-reduce (both are sums, the order is free), before the optimizers - which then apply the same update to both
        copies.  The wire buffer is persistent (GPT-2's wte: 154 MB in fp32 - not allocated per step) and carries bf16
        with fp32 accumulation on the GPU; BOTH ends keep the value that travelled back (the rounded sum), so the two
        copies of the parameter stay bit-identical."""
        for name, p, role in self._tied:
            g = p.grad
            if g is None:
                g = p.grad = torch.zeros_like(p)
            wdt = self._tied_wire_dtype(g)
            wire = self._tied_buf.get(name)
            if wire is None or wire.shape != g.shape or wire.dtype != wdt:
                wire = self._tied_buf[name] = torch.empty(g.shape, dtype=wdt, device=g.device)
            other = self.n_stages - 1 if role == "first" else 0
            if role == "last":
                wire.copy_(g)
                for w in self.comm.batch_p2p([("send", wire, other)]):
                    w.wait()
                for w in self.comm.batch_p2p([("recv", wire, other)]):
                    w.wait()
                g.copy_(wire)
            else:
                for w in self.comm.batch_p2p([("recv", wire, other)]):
                    w.wait()
                g.add_(wire)                   # fp32 accumulate of the widened wire values
                wire.copy_(g)
                if wdt != g.dtype:
                    g.copy_(wire)              # keep exactly what the other end will hold
                for w in self.comm.batch_p2p([("send", wire, other)]):
                    w.wait()

    def stats(self) -> Dict[str, float]:
        """What the last steps looked like from this stage (``timer`` set): mean busy milliseconds per step (forward +
        backward kernels of all micro-batches), time in point-to-point waits, the schedule's theoretical bubble and
        its peak number of live micro-batches."""
        out: Dict[str, float] = {"stage": self.stage, "peak_live_microbatches": self.peak_live,
                                 "bubble_fraction_theory": _sched.bubble_fraction(self.n_stages, max(1, self.n_micro)),
                                 "slots_allocated": self._slots.allocated}
        if self.timer is not None:
            per_occ, counts = self.timer.summary_with_counts()
            steps = max(1, counts.get("tied_sync", 1))
            for k in ("forward", "backward", "p2p", "dp_exposed", "tied_sync"):
                if k in per_occ:
                    out[k + "_ms_per_step"] = round(per_occ[k] * counts[k] / steps, 4)
            out["busy_ms_per_step"] = round(out.get("forward_ms_per_step", 0.0) + out.get("backward_ms_per_step", 0.0), 4)
        if self.device.type == "cuda":
            out["max_memory_allocated_gb"] = round(torch.cuda.max_memory_allocated(self.device) / 2 ** 30, 3)
        return out

    def _run_comm(self, step, sizes, inputs, outputs, grads_in) -> None:
        batch_ops = []
        recvs = []
        for op, mb in step:
            if op == "send_fwd":
                y = outputs[mb].detach()
                if self.send_dtype is not None and y.dtype != self.send_dtype:
                    y = y.to(self.send_dtype)
                y = y.contiguous()
                batch_ops.append(("send", y, self.stage + 1))
            elif op == "recv_fwd":
                shape, dtype = self._in_meta
                buf = self._slots.take((sizes[mb],) + shape, dtype, self.device)
                buf._dpa_slot = True
                batch_ops.append(("recv", buf, self.stage - 1))
                recvs.append(("in", mb, buf))
            elif op == "send_bwd":
                xin = inputs.pop(mb)
                g = xin.grad if xin.grad is not None else torch.zeros_like(xin)
                batch_ops.append(("send", g.contiguous(), self.stage - 1))
                self._slots.give(xin)         # its backward has run; the slot carries a later micro-batch next
            elif op == "recv_bwd":
                y = outputs[mb]
                buf = self._slots.take(tuple(y.shape), y.dtype if self.send_dtype is None else self.send_dtype, self.device)
                buf._dpa_slot = True
                batch_ops.append(("recv", buf, self.stage + 1))
                recvs.append(("grad", mb, buf))
        works = self.comm.batch_p2p(batch_ops)
        # receives must complete before the compute that follows; sends complete lazily
        if len(works) == len(batch_ops):
            for (kind, t, _), w in zip(batch_ops, works):
                if kind == "recv":
                    w.wait()
                else:
                    self._pending_sends.append((w, t))
        else:  # backend coalesced the group into fewer handles
            for w in works:
                w.wait()
        for kind, mb, buf in recvs:
            if kind == "in":
                inputs[mb] = buf
            else:
                y = outputs[mb]
                if buf.dtype == y.dtype:
                    grads_in[mb] = buf
                else:
                    grads_in[mb] = buf.to(y.dtype)
                    self._slots.give(buf)


class _NullCtx:
    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


def build_stage_module(layers: Sequence[nn.Module], bounds: Sequence[Tuple[int, int]], stage: int,
                       model: Optional[nn.Module] = None) -> nn.Module:
    """This stage's layers as one module.  When the full ``model`` is given, the stage
    remembers the model-level name of every state_dict entry (``_dpa_key_map``) so stage
    checkpoints merge back into ONE state_dict with the original keys."""
    a, b = bounds[stage]
    mod = layers[a] if b - a == 1 else nn.Sequential(*layers[a:b])
    if model is not None:
        global_names = {id(t): k for k, t in model.state_dict(keep_vars=True).items()}
        mod._dpa_key_map = {k: global_names[id(t)] for k, t in mod.state_dict(keep_vars=True).items()
                            if id(t) in global_names}
    return mod


# ---------------------------------------------------------------------------- benchmark driver
def run_pipeline_bench(args, world: _comm.Communicator, device: torch.device) -> dict:
    """``bench.py --pipeline S``: S-stage micro-batched pipeline (x replicas when
    world_size > S), same images/sec metric.  Stages are balanced by measured time."""
    from ..models import resnet
    from ..optim import FusedSGD
    from .dp import DataParallel
    from .flat import FlatParams
    from .. import ops

    S = args.pipeline
    n = world.world_size
    if n % S != 0:
        raise SystemExit(f"world size {n} is not a multiple of --pipeline {S}")
    replicas = n // S
    pp_comm, dp_comm, replica, stage = _comm.build_grid(world, replicas, S)
    name = args.model
    tokens = name == "gpt2"
    small = name == "mnist"
    g = torch.Generator(device=device).manual_seed(1234 + replica)
    if small:
        # BASELINE config 2 = the reference's model-parallel example (multinode-MP/multinodeMP.py:118-167): Net1 on
        # stage 0, Net2 on stage 1, SGD lr 0.01 without momentum per stage, batch 256 (the README's run command)
        from ..models import mnist
        if S != 2:
            raise SystemExit("--model mnist --pipeline: the reference split has exactly 2 stages (Net1 | Net2)")
        batch = args.batch or 256
        hw, classes = 28, 10
        x = torch.randn(batch, 1, 28, 28, generator=g, device=device)
        y = torch.randint(0, 10, (batch,), generator=g, device=device)
        layers = mnist.mnist_stages()
        stage_mod = layers[stage].to(device).train()
        bounds = [(0, 1), (1, 2)]
        in_metas = [None, ((320,), torch.float32)]             # [B, 320] fp32: both ranks of the reference know it (MP.py:143,152)
        model = None
    elif tokens:
        from ..models import gpt2
        model = gpt2.gpt2_small()
        classes, seq = model.cfg.vocab_size, args.seq_len
        batch = args.batch or 8 * args.micro_batches      # >= 8 sequences of 1024 tokens per micro-batch
        t = torch.randint(0, classes, (batch, seq + 1), generator=g, device=device)
        x, y = t[:, :-1].contiguous(), t[:, 1:].contiguous()
    else:
        model = resnet.tiny_resnet() if name == "tiny_resnet" else getattr(resnet, name)()
        hw, classes = (64, 16) if name == "tiny_resnet" else (224, 1000)
        # per-replica batch sized so that a MICRO-batch is >= 128 images: below that the conv kernels run under-filled
        # grids and pay their fixed costs per launch (docs/DESIGN.md section 10); 288 GB of HBM per stage allow it
        batch = args.batch or 128 * args.micro_batches
        x = torch.randn(batch, 3, hw, hw, generator=g, device=device)
        y = torch.randint(0, classes, (batch,), generator=g, device=device)
    if not small:
        model = model.to(device).train()
        layers = model.pipeline_layers()
        mb = max(1, batch // args.micro_batches)
        metas: List[Meta] = []
        if device.type == "cuda":
            costs = measure_costs(layers, x[:mb], metas=metas)
        else:
            costs, metas = param_costs(layers), trace_shapes(layers, x[:2])
        # every rank must agree on the split: use stage 0 / replica 0's measurement (and its boundary shapes)
        costs, metas = world.broadcast_object((costs, [(list(sh), _DTYPES.index(dt)) for sh, dt in metas]), src=0)
        metas = [(tuple(sh), _DTYPES[di]) for sh, di in metas]
        bounds = partition_balanced(costs, S)
        in_metas = stage_in_metas(metas, bounds)
        stage_mod = build_stage_module(layers, bounds, stage, model)
        if device.type == "cuda":        # the other stages' layers leave this GPU
            keep = {id(p) for p in stage_mod.parameters()} | {id(b) for b in stage_mod.buffers()}
            for t_ in list(model.parameters()) + list(model.buffers()):
                if id(t_) not in keep:
                    t_.data = torch.empty(0, dtype=t_.dtype)
            torch.cuda.empty_cache()
        del model, layers
    dp = DataParallel(stage_mod, dp_comm, bucket_mb=args.bucket_mb, first_bucket_mb=args.first_bucket_mb,
                      use_sink=not args.no_sink, use_lp=not small)
    if small:
        from ..models import mnist
        opt = FusedSGD(dp.flat, lr=0.01, momentum=0.0)                 # MP.py:120,124
        loss_fn = mnist.nll_loss
    else:
        opt = FusedSGD(dp.flat, lr=0.1, momentum=0.9, weight_decay=1e-4)
        loss_fn = (lambda out, t: ops.softmax_cross_entropy(out.reshape(-1, out.shape[-1]), t.reshape(-1), classes))
    opt.grad_scale = dp.grad_scale
    eng = PipelineStage(stage_mod, pp_comm, loss_fn, n_micro=args.micro_batches, schedule=args.schedule, dp=dp,
                        in_meta=in_metas[stage], negotiate=False)
    from ..metrics import PhaseTimer
    eng.timer = PhaseTimer(device)       # a few dozen HIP events per step: forward / backward / p2p per micro-batch

    from ..utils.pacer import StepPacer
    pacer = StepPacer(device, depth=getattr(args, "run_ahead", 2))       # utils/pacer.py: the host at most two steps ahead

    def step():
        loss = eng.train_step(batch, x if eng.first else None, y if eng.last else None)
        opt.step()
        pacer.step()
        return loss

    for _ in range(args.warmup):
        step()
    eng.timer.summary()                  # drop the warm-up's samples
    if device.type == "cuda":
        torch.cuda.reset_peak_memory_stats(device)
    world.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = None
    for _ in range(args.steps):
        loss = step()
    world.barrier()
    if device.type == "cuda":
        torch.cuda.synchronize()
    elapsed = world.all_reduce_scalar(time.perf_counter() - t0, "max")
    lossv = float(loss.float().cpu()) if loss is not None else float("nan")
    # the loss lives on the last stage; make rank 0 report it
    lossv = world.all_reduce_scalar(lossv if eng.last and replica == 0 else 0.0, "sum")
    shape = {"seq_len": seq} if tokens else {"image": f"{1 if small else 3}x{hw}x{hw}"}
    # one line per stage of replica 0 explains the run: busy time (forward + backward kernels of all micro-batches),
    # point-to-point waits, peak live micro-batches, peak memory - and the measured bubble beside the schedule's
    st = eng.stats()
    st["replica"] = replica
    all_stats = world.gather_object(st, dst=0)
    ms_step = 1000.0 * elapsed / max(1, args.steps)
    if all_stats is not None:
        per_stage = sorted((d for d in all_stats if d.get("replica") == 0), key=lambda d: d["stage"])
        busiest = max((d.get("busy_ms_per_step", 0.0) for d in per_stage), default=0.0)
        pipeline_stats = {"stages": per_stage, "bubble_fraction_theory": _sched.bubble_fraction(S, max(1, args.micro_batches)),
                          "bubble_fraction_measured": (round(1.0 - busiest / ms_step, 4) if busiest > 0 and ms_step > 0 else None),
                          "peak_live_microbatches": [d["peak_live_microbatches"] for d in per_stage],
                          "micro_batch_size": max(1, batch // args.micro_batches)}
    else:
        pipeline_stats = None
    shape["pipeline"] = pipeline_stats
    return {"elapsed": elapsed, "items_per_rank_step": batch * (seq if tokens else 1), "dp_world": replicas, "loss": lossv,
            "config": dict({"model": "gpt2-small-124M" if tokens else ("mnist_cnn (Net1 | Net2)" if small else name),
                            "global_batch": batch * replicas, "slots_allocated": eng._slots.allocated,
                            "parallelism": f"pp{S}xdp{replicas}", "micro_batches": args.micro_batches,
                            "schedule": args.schedule, "stage_bounds": bounds}, **shape)}
