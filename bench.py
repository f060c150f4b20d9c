#!/usr/bin/env python
"""Headline benchmark: ResNet-50 data-parallel training throughput (images/sec, whole job).

    python bench.py --gpus N --steps K --warmup W

N > 1 is launched by the driver as
``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`` (one rank
per GPU over RCCL; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* come from the env).
Every rank runs the FULL training step on its own synthetic 3x224x224 batch
(weak scaling): layout conversion, forward, loss, backward, bucketed all-reduce
overlapped with backward, fused SGD-momentum update (bf16 compute, fp32 master
weights / gradients).  Rank 0 prints ONE JSON line.

Other workloads: ``--model resnet152|gpt2|mnist``; ``--pipeline S`` runs an S-stage
micro-batched pipeline (ranks = stages x replicas) and reports the same metric.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
# Three streams carry the step (compute, weight-gradient side stream, RCCL's): with HIP's default of 4 hardware queues
# per process two of them end up sharing a queue and serialise (measured with real RCCL collectives at world size 1:
# 11,246 img/s with 4 queues, 12,314 with 8).  Must be set before the HIP runtime initialises.
if int(os.environ.get("GPU_MAX_HW_QUEUES", "4")) < 8:
    os.environ["GPU_MAX_HW_QUEUES"] = "8"

import torch  # noqa: E402

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=20)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--model", default="resnet50", choices=["resnet50", "resnet152", "gpt2", "mnist", "tiny_resnet"])
    p.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: 1024 resnet50 - the largest whose tensors stay under the kernels' 2^31-byte operand limit, "
                        "+3-4 %% over 512 -, 1024 resnet152 too, 16 gpt2: sized for 288 GB of HBM; BASELINE.md has 512 and 256 beside them, each "
                        "against stock PyTorch at the SAME batch; 128/N mnist, the reference's global batch)")
    p.add_argument("--seq-len", type=int, default=1024)
    p.add_argument("--bucket-mb", type=float, default=32.0)
    p.add_argument("--first-bucket-mb", type=float, default=4.0)
    p.add_argument("--no-graph", action="store_true", help="mnist: do not capture forward+backward in a hipGraph")
    p.add_argument("--pipeline", type=int, default=0, help="number of pipeline stages (0 = pure data parallel)")
    p.add_argument("--micro-batches", type=int, default=8)
    p.add_argument("--schedule", default="1f1b")
    p.add_argument("--no-sink", action="store_true")
    p.add_argument("--shard-optimizer", action="store_true", help="ZeRO stage 1: optimizer state sharded over the DP ranks")
    p.add_argument("--backend", default="auto")
    p.add_argument("--opt-overlap", default="auto", choices=["auto", "on", "off"],
                   help="optimizer per bucket behind its all-reduce, during backward (optim.FusedSGD.overlap_with_backward); "
                        "auto = on for gpt2 (+0.9 %% on one GPU), off for the ResNets (-0.3 %%: their step is HBM-bound throughout)")
    p.add_argument("--breakdown", action="store_true",
                   help="HIP-event phase timers: forward / backward / exposed communication / optimizer per step, added to the JSON")
    p.add_argument("--comm", default=None, choices=["native", "c10d"], help="collective layer under RCCL (default: DPA_COMM or native)")
    p.add_argument("--allreduce", default=None, choices=["ring", "direct", "ipc"],
                   help="RCCL's ring, the one-shot direct algorithm over grouped send/recv, or ONE kernel over hipIpc-mapped peer memory (csrc/peer_allreduce.hip)")
    p.add_argument("--grad-wire", default=None, choices=["fp32", "bf16"], help="direct all-reduce: gradient format on the links")
    p.add_argument("--run-ahead", type=int, default=2,
                   help="steps the host may be ahead of the GPU (utils/pacer.py; 0 = unbounded, the behaviour until the last day of round 3)")
    return p.parse_args()


def load_baseline(model: str, n: int, per_gpu_batch: int):
    """Stock PyTorch-ROCm numbers measured by scripts/baseline_torch.py on MI355X.
    The reference itself publishes none.  For N > 1 the comparison line is N x the measured
    1-GPU stock number (i.e. stock PyTorch given perfect scaling)."""
    path = os.path.join(ROOT, "BASELINE_MEASURED.json")
    if not os.path.exists(path):
        return None
    try:
        with open(path) as f:
            data = json.load(f)
        entry = data.get(model, {})
        if isinstance(entry, dict) and str(per_gpu_batch) in entry.get("by_batch", {}):
            entry = entry["by_batch"][str(per_gpu_batch)]
        elif not isinstance(entry, dict) or ("batch" in entry and int(entry["batch"]) != per_gpu_batch):
            return None          # the comparison line was measured at a different per-GPU batch
        if str(n) in entry:
            return float(entry[str(n)])
        if "1" in entry:
            return float(entry["1"]) * n
    except Exception:  # noqa: BLE001
        return None
    return None


def peak_flops(dtype: str) -> float:
    """Dense MFMA peak of one MI355X (MI355X_MICROARCH.md): 2.5 PFLOP/s bf16; fp32 through the f32-input MFMA 157 TFLOP/s."""
    return 2.5e15 if dtype == "bf16" else 157.3e12


def _rccl_version():
    """RCCL's own version number (ncclGetVersion through the extension; torch's record as a fallback)."""
    try:
        from distributedpytorch_amd.ops import _ext
        return int(_ext.load().RcclComm.version())
    except Exception:  # noqa: BLE001
        try:
            v = torch.cuda.nccl.version()
            return int(v[0]) * 10000 + int(v[1]) * 100 + int(v[2])
        except Exception:  # noqa: BLE001
            return None


def main():
    args = parse()
    # collective-layer choices must be in the environment before distributedpytorch_amd.comm is imported
    if args.comm:
        os.environ["DPA_COMM"] = args.comm
    if args.allreduce:
        os.environ["DPA_ALLREDUCE"] = args.allreduce
    if args.grad_wire:
        os.environ["DPA_GRAD_WIRE"] = args.grad_wire
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (MI355X); none visible")
    # DPA_SHARE_GPU=1: every rank on GPU 0 (rehearsal of the multi-rank flows on a one-GPU box, with --backend gloo:
    # RCCL refuses two ranks on one device); the number it prints is not a benchmark result
    if os.environ.get("DPA_HANG_DUMP"):     # dump every thread's Python stack and exit if the run is still going after N s
        import faulthandler
        faulthandler.dump_traceback_later(float(os.environ["DPA_HANG_DUMP"]), exit=True)
    dev_index = 0 if os.environ.get("DPA_SHARE_GPU", "0") == "1" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)

    from distributedpytorch_amd import comm as dcomm
    from distributedpytorch_amd.optim import FusedSGD
    from distributedpytorch_amd.parallel.dp import DataParallel
    from distributedpytorch_amd.data import synthetic_batch

    if world > 1 or (dcomm._FORCE_COLLECTIVES and "MASTER_ADDR" in os.environ):
        # DPA_FORCE_COLLECTIVES=1 under torchrun with ONE rank: the full RCCL call path (process group, every bucket
        # all-reduced for real on RCCL's stream) on a one-GPU box
        communicator = dcomm.init_process_group(args.backend, init_method=None, device=device)
    else:
        communicator = dcomm.Communicator(device=device)
    if args.gpus != world and rank == 0:
        print(f"[bench] note: --gpus {args.gpus} but WORLD_SIZE={world}; using WORLD_SIZE", file=sys.stderr)

    torch.manual_seed(1234)
    unit = "images/sec"
    if args.pipeline and args.pipeline > 1:
        if args.model == "gpt2":
            unit = "tokens/sec"
        from distributedpytorch_amd.parallel.pp import run_pipeline_bench
        result = run_pipeline_bench(args, communicator, device)
    else:
        if args.model in ("resnet50", "resnet152", "tiny_resnet"):
            from distributedpytorch_amd.models import resnet
            batch = args.batch or {"resnet50": 1024, "resnet152": 1024}.get(args.model, 256)
            if args.model == "tiny_resnet":
                model = resnet.tiny_resnet(); hw, classes = 64, 16
            else:
                model = getattr(resnet, args.model)(); hw, classes = 224, 1000
            model = model.to(device).train()
            g = torch.Generator(device=device).manual_seed(1234 + rank)
            x = torch.randn(batch, 3, hw, hw, generator=g, device=device)
            y = torch.randint(0, classes, (batch,), generator=g, device=device)
            lr, momentum, wd, use_lp = 0.1, 0.9, 1e-4, True
            loss_fn = lambda: model.loss(x, y)  # noqa: E731
            items = batch
            # training FLOPs per image = 3 x forward (forward, input gradients, weight gradients)
            flops_item = 3.0 * resnet.flops_per_image(model, hw) if isinstance(model, resnet.ResNet) else None
            cfg = {"model": args.model, "global_batch": batch * world, "image": f"3x{hw}x{hw}"}
        elif args.model == "gpt2":
            from distributedpytorch_amd.models import gpt2
            batch = args.batch or 16
            model = gpt2.gpt2_small().to(device).train()
            x, y = synthetic_batch("tokens", batch, device, seed=1234 + rank, seq_len=args.seq_len)
            lr, momentum, wd, use_lp = 0.01, 0.9, 0.0, True
            loss_fn = lambda: model.loss(x, y)  # noqa: E731
            items = batch * args.seq_len
            flops_item = gpt2.flops_per_token(model.cfg, args.seq_len)
            unit = "tokens/sec"
            cfg = {"model": "gpt2-small-124M", "global_batch": batch * world, "seq_len": args.seq_len}
        else:  # mnist: the reference's own DP config (global batch 128, SGD lr .01 momentum .5)
            from distributedpytorch_amd.models import mnist
            batch = args.batch or max(1, 128 // world)
            model = mnist.Net().to(device).train()
            x, y = synthetic_batch("mnist", batch, device, seed=1234 + rank)
            lr, momentum, wd, use_lp = 0.01, 0.5, 0.0, False
            loss_fn = lambda: model.loss(x, y)  # noqa: E731   (head + log-softmax + NLL fused: models/mnist.py)
            items = batch
            flops_item = 3.0 * 2.0 * 480500.0           # 480,500 MACs forward per sample, x3 for fwd + bwd
            cfg = {"model": "mnist_cnn", "global_batch": batch * world, "image": "1x28x28"}

        dp = DataParallel(model, communicator, bucket_mb=args.bucket_mb, first_bucket_mb=args.first_bucket_mb,
                          use_sink=not args.no_sink, use_lp=use_lp, shard_optimizer=args.shard_optimizer)
        if args.shard_optimizer:
            from distributedpytorch_amd.optim import ShardedSGD
            opt = ShardedSGD(dp, lr=lr, momentum=momentum, weight_decay=wd)
            cfg["optimizer_state"] = "sharded (ZeRO-1)"
        else:
            opt = FusedSGD(dp.flat, lr=lr, momentum=momentum, weight_decay=wd)
        opt.grad_scale = dp.grad_scale
        if not args.shard_optimizer and args.model != "mnist":
            # every bucket's update runs behind its all-reduce while backward continues (optim.FusedSGD.overlap_with_backward)
            want = args.opt_overlap == "on" or (args.opt_overlap == "auto" and args.model == "gpt2")
            cfg["optimizer_overlap"] = opt.overlap_with_backward(dp, enabled=want)

        graphed = None
        if args.model == "mnist" and not args.no_graph:
            # launch-bound regime: forward+backward replayed from one hipGraph (parallel/graph.py);
            # the single gradient all-reduce and the fused optimizer kernel stay outside the graph
            from distributedpytorch_amd.parallel.graph import GraphedStep
            model._drop.use_device_counter(device)
            graphed = GraphedStep(dp, lambda xx, yy: model.loss(xx, yy), (x, y))   # Net.loss bumps the dropout counter itself
            # the synthetic batch lives in the graph's input buffers, as the other workloads' batch lives in one device
            # tensor that every step reads: no per-step device-to-device copy of data that never changes
            x, y = graphed.static_inputs
            cfg["hipgraph"] = True

        from distributedpytorch_amd.metrics import PhaseTimer
        timer = PhaseTimer(device, enabled=args.breakdown)

        def step():
            if graphed is not None:
                with timer.phase("forward+backward (hipGraph)"):
                    loss = graphed(x, y)
                with timer.phase("exposed_comm"):
                    dp.reduce_all()
            else:
                with timer.phase("forward"):
                    loss = loss_fn()
                with timer.phase("backward"):          # includes the bucket launches: communication overlapped with it
                    loss.backward()
                with timer.phase("exposed_comm"):      # what the compute stream still has to wait for after backward
                    dp.finish()
            with timer.phase("optimizer"):
                opt.step()
            return loss

        from distributedpytorch_amd.utils.gcguard import GCGuard
        from distributedpytorch_amd.utils.pacer import StepPacer
        # the host stays at most two steps ahead of the GPU (utils/pacer.py: unbounded run-ahead has a slow mode here);
        # the small CNN's 0.13 ms steps are paced every 64th step
        pacer = StepPacer(device, depth=0 if args.run_ahead == 0 else args.run_ahead, stride=64 if args.model == "mnist" else 1)
        for _ in range(args.warmup):
            step()
            pacer.step()
        timer.summary()                      # drop the warm-up's samples
        dp.stats["allreduce_calls"] = dp.stats["allreduce_bytes"] = 0
        # the collector runs at a fixed step period, the same step on every rank (as in train.py): a generation-2
        # pause on one rank is a pause of the whole synchronous job
        with GCGuard(period=500) as guard:
            communicator.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            c0 = time.process_time()
            for _ in range(args.steps):
                loss = step()
                guard.step()
                pacer.step()
            host_cpu = time.process_time() - c0        # CPU time of this process while enqueueing (all its threads)
            communicator.barrier()
            torch.cuda.synchronize()
            elapsed = time.perf_counter() - t0
        elapsed = communicator.all_reduce_scalar(elapsed, "max")
        if getattr(communicator.native, "accepts_host", False):
            layer = "native layer's Python side over a gloo stand-in (utils/fake_native.py): a rehearsal, not RCCL"
        else:
            layer = "native (csrc/rccl_comm.cpp)" if communicator.native is not None else \
                ("c10d" if torch.distributed.is_initialized() else "none (single process, no collectives)")
        no_comm = communicator.native is None and not torch.distributed.is_initialized()
        communicator.check_peer()     # a bounded poll of the peer-mapped kernel that ran out would have left wrong sums behind
        if no_comm:
            algo = None               # nothing is reduced: no algorithm to name
        elif communicator.peer is not None:
            algo = "ipc/" + communicator.grad_wire + " (one kernel over peer-mapped memory)"
        elif communicator.native is not None and communicator._direct:
            algo = "direct/" + communicator.grad_wire
        else:
            algo = "ring"
        result = {"elapsed": elapsed, "items_per_rank_step": items, "loss": float(loss.detach().float().cpu()),
                  "flops_per_item": flops_item,
                  "allreduce_bytes_per_step": dp.stats["allreduce_bytes"] // max(1, args.steps),
                  "allreduce_calls_per_step": dp.stats["allreduce_calls"] / max(1, args.steps),
                  "breakdown_ms": ({k: round(v, 3) for k, v in timer.summary().items()} if args.breakdown else None),
                  "config": dict(cfg, parallelism=f"dp{world}", per_gpu_batch=batch,
                                 buckets=len(dp.buckets), optimizer=f"fused SGD momentum {momentum}", comm_layer=layer,
                                 allreduce=algo, gpu_max_hw_queues=int(os.environ.get("GPU_MAX_HW_QUEUES", "4")),
                                 rccl_version=_rccl_version(),
                                 host_run_ahead_steps=args.run_ahead,
                                 host_cpu_ms_per_step=round(1000.0 * host_cpu / max(1, args.steps), 2),
                                 max_memory_allocated_gb=round(torch.cuda.max_memory_allocated(device) / 1e9, 2),
                                 max_memory_reserved_gb=round(torch.cuda.max_memory_reserved(device) / 1e9, 2),
                                 device_allocs=int(torch.cuda.memory_stats(device).get("num_device_alloc", 0)))}

    if rank == 0:
        elapsed = result["elapsed"]
        value = result["items_per_rank_step"] * result.get("dp_world", world) * args.steps / elapsed
        pgb = result["config"].get("per_gpu_batch", 0)
        base = load_baseline(args.model, world, pgb) if not (args.pipeline and args.pipeline > 1) else None
        strong = args.model == "mnist" and not args.batch      # the reference's config fixes the GLOBAL batch at 128
        metric = "images/sec (whole node) ResNet-50 DP" if args.model == "resnet50" and not args.pipeline else \
            f"{unit} (whole job) {args.model}"
        out = {
            "metric": metric, "value": round(value, 2), "unit": unit, "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1000.0 * elapsed / args.steps, 3), "higher_is_better": True,
            "scaling": "strong" if strong else "weak", "vs_baseline": (round(value / base, 4) if base else None), "dtype": "bf16" if args.model != "mnist" else "fp32",
            "data": "synthetic (random-init weights, on-device random inputs of the named shape)",
            "final_loss": result["loss"], "config": result["config"],
        }
        dtype = out["dtype"]
        if result.get("flops_per_item"):
            # model FLOP utilisation against the dense MFMA peak of the GPUs used (no sparsity figures)
            out["mfu"] = round(value * result["flops_per_item"] / (world * peak_flops(dtype)), 4)
        # (scaling efficiency is for whoever runs N = 1, 2, 4, 8 to compute from the per-N values: scripts/scale_table.py)
        for k in ("allreduce_bytes_per_step", "allreduce_calls_per_step", "breakdown_ms"):
            if result.get(k) is not None:
                out[k] = result[k]
        print(json.dumps(out), flush=True)
    if torch.distributed.is_initialized():
        communicator.barrier()
        dcomm.destroy_process_group()


if __name__ == "__main__":
    main()
