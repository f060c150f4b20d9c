"""Training loops: the reference's two ``run()`` functions, re-built on the engines.

``run_dp``  = multinode-DP/multinodeDP.py:131-157 (synchronous data-parallel SGD)
``run_mp``  = multinode-MP/multinodeMP.py:102-170 (model parallel; here an S-stage,
              micro-batched pipeline, optionally replicated)

Both are ``fn(local_rank, global_rank, world_size, cfg)`` targets for ``launch.spawn``
(reference: ``init_processes`` then ``run``, DP.py:159-162 / MP.py:172-178).
"""
from __future__ import annotations

import os
from math import ceil

import torch

from . import comm as dcomm
from . import data as ddata
from . import metrics as dmetrics
from .checkpoint import load_checkpoint, save_checkpoint
from .config import TrainConfig
from .optim import FusedAdamW, FusedSGD, ShardedSGD, WarmupCosine, clip_grad_norm
from .parallel.dp import DataParallel
from .parallel.flat import FlatParams
from .parallel.pp import PipelineStage, build_stage_module, partition_balanced, param_costs
from .utils.gcguard import GCGuard
from .utils.pacer import StepPacer


def pick_device(cfg: TrainConfig, local_rank: int) -> torch.device:
    want = cfg.device
    if want == "auto":
        want = "cuda" if (torch.cuda.is_available() and cfg.backend not in ("gloo",)) else "cpu"
    if want == "cuda":
        index = 0 if os.environ.get("DPA_SHARE_GPU", "0") == "1" else local_rank    # rehearsal: all ranks on GPU 0 (gloo)
        torch.cuda.set_device(index)
        return torch.device("cuda", index)
    return torch.device("cpu")


def build_model(name: str, dropout: float = 0.0):
    """Returns (model, dataset_kind, loss(model, x, y), use_lp, classes)."""
    from . import ops
    if name == "mnist_cnn":
        from .models import mnist
        m = mnist.Net()
        return m, "mnist", (lambda mod, x, y: mod.loss(x, y)), False, 10
    if name in ("resnet50", "resnet101", "resnet152", "tiny_resnet"):
        from .models import resnet
        m = getattr(resnet, name)()
        return m, ("tiny_image" if name == "tiny_resnet" else "imagenet"), (lambda mod, x, y: mod.loss(x, y)), True, m.num_classes
    if name in ("gpt2", "gpt2_small", "gpt2_tiny"):
        from .models import gpt2
        m = gpt2.gpt2_tiny(dropout=dropout) if name == "gpt2_tiny" else gpt2.GPT2(gpt2.GPT2Config(dropout=dropout))
        return m, "tokens", (lambda mod, x, y: mod.loss(x, y)), True, m.cfg.vocab_size
    raise ValueError(f"unknown model {name!r}")


def _rank0_print(msg: str) -> None:
    if int(os.environ.get("RANK", "0")) == 0 and os.environ.get("DPA_QUIET", "0") != "1":
        print(f"[data] {msg}", flush=True)


def build_dataset(kind: str, cfg: TrainConfig, classes: int, model=None, device=None):
    # the dataset lives on the GPU only with --device-data; otherwise it stays on the host and data.PrefetchLoader
    # pins each batch and copies it on its own stream (a device handed to a host dataset made every batch a pageable
    # copy on the compute stream and switched the loader's pinned staging off)
    on_device = device is not None and device.type == "cuda" and cfg.device_data
    gen = dict(device=device if on_device else None, on_device=on_device)
    if kind == "mnist":
        # the reference's dataset (DP.py:100-107): MNIST IDX files under --data-dir when present, never downloaded
        return ddata.mnist_or_synthetic(cfg.data_dir, cfg.dataset_size, cfg.seed, log=_rank0_print, **gen)
    if kind == "imagenet":
        return ddata.SyntheticImageDataset(cfg.dataset_size, (3, 224, 224), classes, cfg.seed, **gen)
    if kind == "tiny_image":
        return ddata.SyntheticImageDataset(cfg.dataset_size, (3, 64, 64), classes, cfg.seed, **gen)
    if kind == "tokens":
        c = model.cfg
        return ddata.SyntheticTokenDataset(cfg.dataset_size, min(c.n_positions, 1024), c.vocab_size, cfg.seed)
    raise ValueError(kind)


def run_dp(local_rank: int, global_rank: int, world_size: int, cfg: TrainConfig):
    """Distributed synchronous SGD (reference DP.py:131-157)."""
    device = pick_device(cfg, local_rank)
    dcomm.set_sync_debug(cfg.sync_debug)
    comm = dcomm.init_process_group(cfg.backend, cfg.dist_url, global_rank, world_size, cfg.timeout_s, device)
    last = None
    try:
        torch.manual_seed(cfg.seed)                      # identical initial replicas (DP.py:138) ...
        model, kind, loss_fn, use_lp, classes = build_model(cfg.model, cfg.dropout)
        model = model.to(device).train()
        dataset = build_dataset(kind, cfg, classes, model, device)
        loader, bsz = ddata.partition_dataset(dataset, comm.rank, comm.world_size, cfg.batch_size, cfg.seed, device,
                                              workers=max(1, cfg.workers))
        dp = DataParallel(model, comm, bucket_mb=cfg.bucket_mb, first_bucket_mb=cfg.first_bucket_mb,
                          use_lp=use_lp, shard_optimizer=cfg.shard_optimizer)   # ... and a broadcast from rank 0 for safety
        if cfg.optimizer == "adamw":
            if cfg.shard_optimizer:
                raise ValueError("--shard-optimizer is implemented for SGD (optim.ShardedSGD)")
            opt = FusedAdamW(dp.flat, lr=cfg.lr, weight_decay=cfg.weight_decay)
        elif cfg.shard_optimizer:
            opt = ShardedSGD(dp, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
        else:
            opt = FusedSGD(dp.flat, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)
        opt.grad_scale = dp.grad_scale                   # the `/= world_size` of DP.py:129, fused
        start_epoch = 0
        if cfg.resume:
            payload = load_checkpoint(cfg.resume, model, opt, flat=dp.flat)
            start_epoch = int(payload.get("epoch", -1)) + 1
        log = dmetrics.JsonlLogger(cfg.log_json, global_rank)
        meter = dmetrics.ThroughputMeter(device)
        num_batches = ceil(len(loader.dataset) / float(bsz))          # DP.py:145
        if cfg.steps_per_epoch:
            num_batches = min(num_batches, cfg.steps_per_epoch)
        last = None
        graphed = None
        use_graph = cfg.graph and device.type == "cuda" and cfg.model == "mnist_cnn" and hasattr(model, "_drop")
        if use_graph:
            # launch-bound regime (the reference's CNN: ~20 kernels of a few microseconds): forward + backward are
            # captured once into a hipGraph and replayed; ragged batches keep the eager path.  Captured here, on a
            # placeholder batch, BEFORE the loader's prefetch thread exists: a capture must not race another
            # thread's allocations and copies
            from .parallel.graph import GraphedStep
            x0, y0 = dataset.batch(list(range(min(int(bsz), len(dataset)))))
            if x0.shape[0] == int(bsz):
                model._drop.use_device_counter(device)
                # (Net.loss's kernel bumps the device-resident dropout counter: no after_backward hook)
                graphed = GraphedStep(dp, lambda a, b: loss_fn(model, a, b), (x0.to(device), y0.to(device)))
        accum = max(1, cfg.accum_steps)
        if accum > 1:
            graphed = None                                   # (the captured step has no accumulate-only variant)
            opt.grad_scale = dp.grad_scale / accum           # mean over ranks AND over the accumulated micro-steps
        steps_this_epoch = min(len(loader), cfg.steps_per_epoch) if cfg.steps_per_epoch else len(loader)
        sched = None
        if cfg.warmup_steps > 0 or cfg.cosine_steps > 0:
            total_steps = cfg.cosine_steps or (cfg.epochs * -(-steps_this_epoch // accum))
            sched = WarmupCosine(cfg.lr, cfg.warmup_steps, max(total_steps, cfg.warmup_steps + 1), cfg.min_lr)
        want_overlap = cfg.opt_overlap == "on" or (cfg.opt_overlap == "auto" and getattr(model, "optimizer_overlap_pays", False))
        if (type(opt) is FusedSGD and graphed is None and sched is None and cfg.clip_grad_norm <= 0 and not cfg.sync_debug
                and want_overlap):                       # measured: +0.9 % GPT-2, -0.3 % ResNet-50 (HBM-bound throughout)
            # constant-lr SGD without clipping: every bucket is updated behind its all-reduce, during backward
            opt.overlap_with_backward(dp)
        guard = GCGuard(period=500)      # collector at a fixed step period, the same step on every rank (utils/gcguard.py)
        guard.__enter__()
        # the host may be two steps ahead of the GPU, not an epoch (utils/pacer.py); the graph-replayed small CNN every 64th step
        pacer = StepPacer(device, depth=2, stride=64 if graphed is not None else 1)
        # --breakdown: HIP events around forward / backward / dp.finish() (= the part of the gradient reduction that backward
        # did NOT hide: "exposed communication") / optimizer; read back once per epoch, no host sync inside the step
        phases = dmetrics.PhaseTimer(device, enabled=cfg.breakdown)
        for epoch in range(start_epoch, cfg.epochs):
            loader.set_epoch(epoch)
            epoch_loss = torch.zeros((), device=device)
            meter.reset(); meter.start()
            for step, (x, y) in enumerate(loader):
                if cfg.steps_per_epoch and step >= cfg.steps_per_epoch:
                    break
                if graphed is not None and x.shape == graphed.static_inputs[0].shape:
                    with phases.phase("forward_backward_graph"):
                        loss = graphed(x, y)
                    epoch_loss += loss.detach()
                    with phases.phase("comm_exposed"):
                        dp.reduce_all()                  # no hook fires in a replay: the engine launches the bucket(s)
                elif accum > 1 and (step + 1) % accum != 0 and step + 1 < steps_this_epoch:
                    # gradient accumulation: local backward only, nothing on the wire until the last micro-step
                    with dp.no_sync():
                        loss = loss_fn(model, x, y)
                        epoch_loss += loss.detach()
                        loss.backward()
                    meter.update(x.shape[0])
                    guard.step()
                    pacer.step()
                    continue
                else:
                    with phases.phase("forward"):
                        loss = loss_fn(model, x, y)
                    epoch_loss += loss.detach()          # stays on device: no host sync (DP.py:153)
                    with phases.phase("backward"):
                        loss.backward()
                    with phases.phase("comm_exposed"):
                        dp.finish()                      # bucketed all-reduce, overlapped with backward
                if cfg.sync_debug and not cfg.shard_optimizer:   # (sharded: only the owner holds a shard's reduced gradient)
                    dmetrics.check_reduced_gradients(dp.flat, comm)
                if cfg.clip_grad_norm > 0:
                    shard = (opt.lo, opt.hi) if cfg.shard_optimizer else None
                    clip_grad_norm(dp.flat, cfg.clip_grad_norm, opt.grad_scale, shard, comm, optimizer=opt)
                if sched is not None:
                    sched.apply(opt, opt.steps)
                with phases.phase("optimizer"):
                    opt.step()                           # fused: scale, momentum, update, bf16 shadow, zero grad
                meter.update(x.shape[0])
                guard.step()
                pacer.step()
            last = float(epoch_loss / num_batches)
            comm.check_peer()                            # DPA_ALLREDUCE=ipc: no bounded poll of the peer-mapped kernel ran out
            breakdown = {k + "_ms": round(v, 4) for k, v in phases.summary().items()} if cfg.breakdown else {}
            rate = meter.rate()
            print(dmetrics.epoch_line_dp(comm.rank, epoch, last), flush=True)            # DP.py:157
            total = comm.all_reduce_scalar(rate, "sum")
            if comm.rank == 0:
                print(f"[dp] epoch {epoch}: {total:.1f} samples/s whole job ({rate:.1f}/rank, {meter.ms_per_step():.2f} ms/step)", flush=True)
            if breakdown and comm.rank == 0:
                print("[dp] epoch %d breakdown (mean ms/step): %s" % (epoch, ", ".join(f"{k[:-3]} {v:.3f}" for k, v in breakdown.items())), flush=True)
            log.log(kind="dp_epoch", epoch=epoch, loss=last, samples_per_sec=total, world_size=world_size,
                    ms_per_step=meter.ms_per_step(), allreduce_calls=dp.stats["allreduce_calls"],
                    allreduce_bytes=dp.stats["allreduce_bytes"], **breakdown)
            if cfg.checkpoint:
                save_checkpoint(cfg.checkpoint, model, opt, epoch, comm=comm)
        guard.__exit__(None, None, None)
    except BaseException:
        # a failing rank must EXIT, at once and with its own traceback: its peers are blocked in a collective that waits
        # for it, so a barrier here would hang until the process-group timeout (and mask the error); the launcher sees
        # the non-zero exit and terminates the rest (launch.SpawnContext.join)
        raise
    comm.barrier()
    dcomm.destroy_process_group()
    return last


def run_mp(local_rank: int, global_rank: int, world_size: int, cfg: TrainConfig):
    """Model-parallel training (reference MP.py:102-170), generalised to S micro-batched stages."""
    device = pick_device(cfg, local_rank)
    dcomm.set_sync_debug(cfg.sync_debug)
    world = dcomm.init_process_group(cfg.backend, cfg.dist_url, global_rank, world_size, cfg.timeout_s, device)
    loss = None
    try:
        stages = cfg.stages
        replicas = max(1, cfg.dp_replicas)
        if stages * replicas > world_size:
            raise ValueError(f"{stages} stages x {replicas} replicas need {stages * replicas} ranks, have {world_size}")
        # ranks beyond stages*replicas idle, like ranks >= 2 in the reference (MP.py:118-124 only uses 0 and 1)
        pp_comm, dp_comm, replica, stage = dcomm.build_grid(world, replicas, stages)
        if stage < 0:
            # an idle rank leaves now and takes part in nothing timed: parked in a barrier it would hit the process-group
            # timeout in any run longer than cfg.timeout_s and take the working ranks down with it
            return None
        active = getattr(world, "active", None) or world     # the ranks that train (== world when none idles)
        torch.manual_seed(cfg.seed)
        from . import ops
        dataset = None
        if cfg.model == "mnist_cnn":
            from .models import mnist
            full = None
            if stages != 2:
                raise ValueError("the MNIST model-parallel example has exactly 2 stages (Net1 | Net2)")
            layers = mnist.mnist_stages()                 # stage k on rank k (MP.py:118-124)
            stage_mod = layers[stage]
            bounds = [(0, 1), (1, 2)]
            loss_fn = mnist.nll_loss
            kind, classes, use_lp = "mnist", 10, False
        else:
            full, kind, _, use_lp, classes = build_model(cfg.model, cfg.dropout)
            layers = full.pipeline_layers()
            # a resumed run cuts the stages where the checkpointed run did: the split below comes from wall-clock
            # measurements that differ from run to run, and optimizer state is stored per stage by stage-local index
            saved_bounds = None
            if cfg.resume:
                from .checkpoint import peek_extra
                ex = peek_extra(cfg.resume)
                if ex.get("stage_bounds") is not None and int(ex.get("stages", -1)) == stages:
                    saved_bounds = [tuple(int(v) for v in b) for b in ex["stage_bounds"]]
            # stages balanced by measured forward+backward TIME on the GPU (rank 0's measurement, so every rank cuts
            # at the same layers); by parameter count only where nothing can be timed (CPU)
            if saved_bounds is not None:
                costs = None
            elif device.type == "cuda":
                from .parallel.pp import measure_costs
                full = full.to(device).train()
                mb = max(1, cfg.batch_size // max(1, cfg.micro_batches))
                dataset = build_dataset(kind, cfg, classes, full, device)
                sample = dataset.batch(list(range(min(mb, len(dataset)))))[0].to(device)
                costs = measure_costs(layers, sample)
            else:
                costs = param_costs(layers)
            if saved_bounds is not None:
                bounds = saved_bounds
            else:
                costs = active.broadcast_object(costs, src=0)
                bounds = partition_balanced(costs, stages)
            stage_mod = build_stage_module(layers, bounds, stage, full)
            if device.type == "cuda":
                # the other stages' layers leave this GPU: only this stage's parameters stay resident
                keep = {id(p) for p in stage_mod.parameters()} | {id(b) for b in stage_mod.buffers()}
                for t in list(full.parameters()) + list(full.buffers()):
                    if id(t) not in keep:
                        t.data = torch.empty(0, dtype=t.dtype)
            if dataset is None:
                dataset = build_dataset(kind, cfg, classes, full, device)
            del full, layers
            full = None
            loss_fn = (lambda out, t: ops.softmax_cross_entropy(out.reshape(-1, out.shape[-1]), t.reshape(-1), classes))
        stage_mod = stage_mod.to(device).train()
        if dataset is None:
            dataset = build_dataset(kind, cfg, classes, full, device)
        # every rank walks the SAME un-shuffled loader (MP.py:104-116): stage 0 consumes the data, the
        # last stage the targets; replicas take disjoint shards
        if replicas > 1:
            part = ddata.DataPartitioner(dataset, [1.0 / replicas] * replicas, cfg.seed).use(replica)
        else:
            part = ddata.Partition(dataset, list(range(len(dataset))))
        loader = ddata.BatchLoader(part, cfg.batch_size, shuffle=False, device=None if device.type == "cuda" else device)
        if device.type == "cuda":
            loader = ddata.PrefetchLoader(loader, device, workers=max(1, cfg.workers))
        dp = DataParallel(stage_mod, dp_comm, bucket_mb=cfg.bucket_mb, first_bucket_mb=cfg.first_bucket_mb,
                          use_lp=use_lp) if (dp_comm is not None) else None
        flat = dp.flat if dp is not None else FlatParams(stage_mod, use_lp=use_lp)
        opt = FusedSGD(flat, lr=cfg.lr, momentum=cfg.momentum, weight_decay=cfg.weight_decay)   # one optimizer per stage
        if dp is not None:
            opt.grad_scale = dp.grad_scale
        eng = PipelineStage(stage_mod, pp_comm, loss_fn, n_micro=cfg.micro_batches, schedule=cfg.schedule, dp=dp)
        start_epoch = 0
        if cfg.resume:
            payload = load_checkpoint(cfg.resume, stage_mod, opt, strict=False, flat=flat, stage=stage)
            start_epoch = int(payload.get("epoch", -1)) + 1
        log = dmetrics.JsonlLogger(cfg.log_json, global_rank, only_rank0=False)
        meter = dmetrics.ThroughputMeter(device)
        loss = None
        phases = dmetrics.PhaseTimer(device, enabled=cfg.breakdown)
        eng.timer = phases if cfg.breakdown else None          # forward / backward / p2p-wait per micro-batch (parallel/pp.py)
        pacer = StepPacer(device, depth=2)                      # utils/pacer.py
        for epoch in range(start_epoch, cfg.epochs):
            meter.reset(); meter.start()
            for step, (x, y) in enumerate(loader):
                if cfg.steps_per_epoch and step >= cfg.steps_per_epoch:
                    break
                with phases.phase("pipeline_step"):
                    loss = eng.train_step(x.shape[0], x if eng.first else None, y if eng.last else None)
                with phases.phase("optimizer"):
                    opt.step()
                meter.update(x.shape[0])
                pacer.step()
            breakdown = {k + "_ms": round(v, 4) for k, v in phases.summary().items()} if cfg.breakdown else {}
            if breakdown:
                print("[mp] stage %d replica %d epoch %d breakdown (mean ms per occurrence): %s" % (
                    stage, replica, epoch, ", ".join(f"{k[:-3]} {v:.3f}" for k, v in breakdown.items())), flush=True)
                if not (eng.last and replica == 0):
                    log.log(kind="mp_stage", epoch=epoch, stage=stage, replica=replica, **breakdown)
            if eng.last and replica == 0:
                print(dmetrics.epoch_line_mp(epoch, loss), flush=True)                      # MP.py:169-170
                log.log(kind="mp_epoch", epoch=epoch, loss=float(loss), samples_per_sec=meter.rate() * replicas,
                        stage=stage, ms_per_step=meter.ms_per_step(), **breakdown)
            if dp_comm is not None:
                dp_comm.check_peer()
            if cfg.checkpoint:
                save_checkpoint(cfg.checkpoint, stage_mod, opt, epoch, comm=dp_comm, pipeline_comm=pp_comm,
                                extra={"stage_bounds": [list(b) for b in bounds], "stages": stages})
    except BaseException:
        raise                    # exit with the traceback; no barrier on the error path (see run_dp)
    active.barrier()
    dcomm.destroy_process_group()
    return float(loss) if loss is not None else None
