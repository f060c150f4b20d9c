"""Worker functions for the multi-process CPU (gloo) tests.  They live in the package so
that ``spawn`` children can import them by module path."""
from __future__ import annotations

import os
import time

import torch

from .. import comm as dcomm
from ..optim import FusedSGD
from ..parallel.dp import DataParallel
from ..parallel.flat import FlatParams
from ..parallel.pp import PipelineStage, build_stage_module, partition_balanced, param_costs


_ports_handed_out = set()


def free_port() -> int:
    """A port for a rendezvous server that is started a moment later by ANOTHER process.  Not ``bind(0)``: that answers
    with a port of the kernel's ephemeral range (32768-60999 here), which the outgoing connections of the test before -
    gloo pairs, store clients - draw from too; between this probe and the server's ``listen`` one of them took the port
    once in ~700 spawns (EADDRINUSE in the GPU tier).  So: a random port BELOW the ephemeral range that binds now, and
    never the same one twice in a process."""
    import random
    import socket
    rng = random.Random()
    for _ in range(200):
        port = rng.randrange(20000, 32000)
        if port in _ports_handed_out:
            continue
        s = socket.socket()
        try:
            s.bind(("127.0.0.1", port))
        except OSError:
            continue
        finally:
            s.close()
        _ports_handed_out.add(port)
        return port
    raise RuntimeError("no free rendezvous port between 20000 and 32000")


def _make_model(name: str):
    torch.manual_seed(1234)
    if name == "mnist":
        from ..models import mnist
        m = mnist.Net()
        m.conv2_drop.p = 0.0
        m.eval()       # dropout off so single-process and distributed runs are comparable
        return m, (lambda mod, x, y: mnist.nll_loss(mod(x), y)), (1, 28, 28), 10
    from ..models import resnet
    m = resnet.tiny_resnet()
    return m.train(), (lambda mod, x, y: mod.loss(x, y)), (3, 32, 32), 16


def _data(shape, classes, n, seed=7):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(n, *shape, generator=g), torch.randint(0, classes, (n,), generator=g)


def _tokens(n, seq=128, vocab=500, seed=7):
    g = torch.Generator().manual_seed(seed)
    t = torch.randint(0, vocab, (n, seq + 1), generator=g)
    return t[:, :-1].contiguous(), t[:, 1:].contiguous()


def dp_worker(local_rank, global_rank, world_size, url, model_name, out_dir, bucket_kb, steps, mode="hooks", backend="gloo"):
    """Each rank takes its shard of ONE global batch; writes reduced grads + params.
    backend "nccl": one GPU per rank over RCCL (multi-GPU boxes only; tests/test_multi_gpu.py);
    "gloo_gpu": all ranks share GPU 0 and reduce over gloo (tests/test_gpu_e2e.py)."""
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    elif backend == "gloo_gpu":             # every rank on GPU 0, collectives over gloo: the GPU engine (HIP kernels,
        torch.cuda.set_device(0)            # side stream, bucket launches) at world size > 1 on a one-GPU box
        device = torch.device("cuda", 0)
        backend = "gloo"
    else:
        device = torch.device("cpu")
    # two processes time-share one GPU under gloo_gpu: a cold box can stall a rank for tens of seconds
    comm = dcomm.init_process_group(backend, url, global_rank, world_size, 180.0 if device.type == "cuda" else 60.0, device)
    model, loss_fn, shape, classes = _make_model(model_name)
    model = model.to(device)
    n_global = 8 * world_size
    x, y = _data(shape, classes, n_global)
    per = n_global // world_size
    xs, ys = x[global_rank * per:(global_rank + 1) * per].to(device), y[global_rank * per:(global_rank + 1) * per].to(device)
    dp = DataParallel(model, comm, bucket_mb=bucket_kb / 1024.0, first_bucket_mb=bucket_kb / 4096.0, use_lp=(model_name != "mnist"),
                      shard_optimizer=(mode == "zero"))
    if mode == "zero":                       # optimizer state sharded over the ranks (optim.ShardedSGD)
        from ..optim import ShardedSGD
        opt = ShardedSGD(dp, lr=0.05, momentum=0.5, weight_decay=1e-3)
    else:
        opt = FusedSGD(dp.flat, lr=0.05, momentum=0.5, weight_decay=1e-3 if mode == "zero_ref" else 0.0)
    opt.grad_scale = dp.grad_scale
    losses = []
    for s in range(steps):
        if mode == "reduce_all":            # the hipGraph-replay flow: no hook-driven launches, reduce afterwards
            with dp.no_sync():
                loss = loss_fn(model, xs, ys)
                loss.backward()
            dp.reduce_all()
        else:
            loss = loss_fn(model, xs, ys)
            loss.backward()
            dp.finish()
        if s == 0 and mode != "zero":        # (sharded: a rank only holds the reduced gradient of its own shard)
            if device.type == "cuda":
                torch.cuda.synchronize()
            g = dp.flat.grad.clone().cpu() * dp.grad_scale
            torch.save({n: g[dp.flat.offsets[i]:dp.flat.offsets[i] + p.numel()].view(p.shape).clone()
                        for i, (n, p) in enumerate(zip(dp.flat.names, dp.flat.params))},
                       os.path.join(out_dir, f"grads_{global_rank}.pt"))
        opt.step()
        losses.append(float(loss.detach()))
    torch.save({"losses": losses, "params": {n: p.detach().cpu().clone() for n, p in model.named_parameters()},
                "buckets": len(dp.buckets), "calls": dp.stats["allreduce_calls"], "opt": opt.state_dict(),
                "peer_launches": int(comm.peer.launches) if comm.peer is not None else 0,
                "native_calls": list(getattr(comm.native, "calls", [])),       # utils/fake_native.py rehearsals
                "momentum_numel": 0 if opt.momentum_buf is None else opt.momentum_buf.numel()},
               os.path.join(out_dir, f"final_{global_rank}.pt"))
    comm.barrier()
    dcomm.destroy_process_group()


def pp_worker(local_rank, global_rank, world_size, url, model_name, out_dir, stages, replicas, n_micro, schedule, batch, steps,
              share_gpu=False, backend="gloo", negotiate=True):
    """Pipeline (x replicas); writes per-stage grads after the first step and losses.
    ``backend`` "gloo": CPU tensors over gloo (the CPU tier).  ``share_gpu`` (or backend "gloo_gpu"): every stage runs
    the HIP engine on GPU 0 (fused blocks with several micro-batches in flight, gradient sinks accumulating over
    micro-batches, side stream), activations travel host-staged over gloo.  "nccl": one GPU per rank, activations and
    gradients device to device over RCCL (multi-GPU boxes only; tests/test_multi_gpu.py) - the transport the pipeline
    engine is built for (the reference stages through the host because gloo must, MP.py:139-145).
    ``negotiate`` False: boundary shapes come from a local trace (``trace_shapes``), nothing is negotiated on the wire."""
    device = torch.device("cpu")
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    elif share_gpu or backend == "gloo_gpu":
        share_gpu, backend = True, "gloo"
        torch.cuda.set_device(0)
        device = torch.device("cuda", 0)
    on_gpu = device.type == "cuda"
    world = dcomm.init_process_group(backend, url, global_rank, world_size, 180.0 if on_gpu else 60.0, device)
    pp_comm, dp_comm, replica, stage = dcomm.build_grid(world, replicas, stages)
    if stage >= 0:
        torch.manual_seed(1234)
        if model_name == "mnist":
            from ..models import mnist
            layers = mnist.mnist_stages()
            stage_mod = layers[stage]
            loss_fn, shape, classes = mnist.nll_loss, (1, 28, 28), 10
        elif model_name == "gpt2_tiny":
            from ..models import gpt2
            from .. import ops
            full = gpt2.gpt2_tiny().train()
            layers = full.pipeline_layers()
            bounds = partition_balanced(param_costs(layers), stages)
            stage_mod = build_stage_module(layers, bounds, stage, full)
            loss_fn = (lambda o, t: ops.softmax_cross_entropy(o.reshape(-1, o.shape[-1]), t.reshape(-1), 500))
            shape, classes = None, 500
        else:
            from ..models import resnet
            from .. import ops
            full = resnet.tiny_resnet().train()
            layers = full.pipeline_layers()
            bounds = partition_balanced(param_costs(layers), stages)
            stage_mod = build_stage_module(layers, bounds, stage, full)
            loss_fn, shape, classes = (lambda o, t: ops.softmax_cross_entropy(o, t, 16)), (3, 32, 32), 16
        x, y = _data(shape, classes, batch * replicas) if shape is not None else _tokens(batch * replicas)
        x, y = x[replica * batch:(replica + 1) * batch].to(device), y[replica * batch:(replica + 1) * batch].to(device)
        in_meta = None
        if not negotiate:
            # boundary shapes without any wire traffic: the reference's split is known statically ([B, 320] fp32,
            # MP.py:143,152); other models are traced locally on the CPU - every rank builds the same seeded model.
            # (On a GPU the boundary is whatever the HIP engine emits - NHWC bf16 - which run_pipeline_bench gets from
            # measure_costs; the test workers negotiate there.)
            from ..parallel.pp import trace_shapes, stage_in_metas
            if model_name == "mnist":
                in_meta = [None, ((320,), torch.float32)][stage]
            elif not on_gpu:
                in_meta = stage_in_metas(trace_shapes(layers, x[:2]), bounds)[stage]
            else:
                negotiate = True
        stage_mod = stage_mod.to(device)
        dp = DataParallel(stage_mod, dp_comm) if (replicas > 1 or on_gpu) else None
        flat = dp.flat if dp is not None else FlatParams(stage_mod)
        opt = FusedSGD(flat, lr=0.05, momentum=0.0)
        if dp is not None:
            opt.grad_scale = dp.grad_scale
        eng = PipelineStage(stage_mod, pp_comm, loss_fn, n_micro=n_micro, schedule=schedule, dp=dp,
                            in_meta=in_meta, negotiate=negotiate)
        losses, slots = [], []
        for s in range(steps):
            loss = eng.train_step(batch, x if eng.first else None, y if eng.last else None)
            slots.append(eng._slots.allocated)
            if s == 0:
                scale = opt.grad_scale if dp is not None else 1.0
                if on_gpu:
                    torch.cuda.synchronize()
                torch.save({n: (p.grad * scale).detach().cpu().clone() for n, p in stage_mod.named_parameters()},
                           os.path.join(out_dir, f"grads_r{replica}_s{stage}.pt"))
            opt.step()
            if loss is not None:
                losses.append(float(loss))
        torch.save({"losses": losses, "slots": slots,
                    "native_calls": {"world": list(getattr(world.native, "calls", [])),
                                     "pp": list(getattr(pp_comm.native, "calls", [])) if pp_comm is not None else [],
                                     "dp": list(getattr(dp_comm.native, "calls", [])) if dp_comm is not None else []}},
                   os.path.join(out_dir, f"final_r{replica}_s{stage}.pt"))
        from ..checkpoint import save_checkpoint
        save_checkpoint(os.path.join(out_dir, f"ckpt_r{replica}.pt"), stage_mod, opt, 0, comm=dp_comm, pipeline_comm=pp_comm)
    world.barrier()
    dcomm.destroy_process_group()


def peer_worker(local_rank, global_rank, world_size, url, out_dir, backend="gloo_gpu", wire="fp32", sizes=(1003, 21840, 3_000_017)):
    """All-reduce over hipIpc-mapped peer memory (csrc/peer_allreduce.hip; the environment carries DPA_ALLREDUCE=ipc).
    "gloo_gpu": every rank on GPU 0 (a peer mapping is then a second mapping of the same HBM: the protocol - slicing,
    flags, staging halves, chunking - is the one that runs between GPUs); "nccl": one GPU per rank beside the native
    RCCL communicator.  Writes every rank's results and the inputs' seeds."""
    os.environ["DPA_GRAD_WIRE"] = wire
    if backend == "nccl":
        torch.cuda.set_device(local_rank)
        device = torch.device("cuda", local_rank)
    else:
        torch.cuda.set_device(0)
        device, backend = torch.device("cuda", 0), "gloo"
    comm = dcomm.init_process_group(backend, url, global_rank, world_size, 180.0, device)
    out = {"attached": comm.peer is not None, "results": {}, "burst": None, "error_seq": -1}
    if comm.peer is not None:
        for n in sizes:
            g = torch.Generator().manual_seed(1000 * n % 7919 + global_rank)
            x = torch.randn(n, generator=g)
            t = None
            for _ in range(3):                       # the same input three times: launches alternate staging halves
                t = x.to(device)
                comm.all_reduce(t)
            torch.cuda.synchronize()
            out["results"][n] = t.cpu()
        # a burst of asynchronous reductions on separate tensors (what the bucket hooks issue during backward)
        g = torch.Generator().manual_seed(77 + global_rank)
        xs = [torch.randn(50_000 + 64 * i, generator=g).to(device) for i in range(24)]
        works = [comm.all_reduce(t, async_op=True) for t in xs]
        for w in works:
            w.wait()
        torch.cuda.synchronize()
        out["burst"] = torch.stack([t[:1000] for t in xs]).cpu()
        out["error_seq"] = int(comm.peer.error_seq())
        out["launches"] = int(comm.peer.launches)
    torch.save(out, os.path.join(out_dir, f"peer_{global_rank}.pt"))
    comm.barrier()
    comm.close()
    dcomm.destroy_process_group()


def peer_refusal_worker(local_rank, global_rank, world_size, url, out_dir):
    """CPU ranks ask for the peer-mapped all-reduce: every rank must decline together (one agreement on the control
    plane) and keep reducing over gloo."""
    comm = dcomm.init_process_group("gloo", url, global_rank, world_size, 60.0, torch.device("cpu"))
    comm.attach_peer()
    t = torch.full((5,), float(global_rank + 1))
    comm.all_reduce(t)
    torch.save({"attached": comm.peer is not None, "sum": t}, os.path.join(out_dir, f"refusal_{global_rank}.pt"))
    comm.check_peer()                       # no layer, nothing to report
    comm.barrier()
    dcomm.destroy_process_group()


def crash_worker(local_rank, global_rank, world_size, url):
    """Rank 1 dies before the collective; the others would hang forever without the launcher's monitor."""
    comm = dcomm.init_process_group("gloo", url, global_rank, world_size, 120.0, torch.device("cpu"))
    if global_rank == 1:
        raise RuntimeError("injected failure on rank 1")
    time.sleep(0.5)
    t = torch.ones(4)
    comm.all_reduce(t)          # blocks: rank 1 never arrives
    time.sleep(600)


def ok_worker(local_rank, global_rank, world_size, out_dir):
    with open(os.path.join(out_dir, f"rank_{global_rank}.txt"), "w") as f:
        f.write(f"{local_rank} {global_rank} {world_size} {os.environ.get('RANK')} {os.environ.get('LOCAL_RANK')}")
