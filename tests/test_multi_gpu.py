"""T4 tier: the gloo-proven distributed logic again over RCCL, one GPU per rank.

Needs >= 2 GPUs in one box, which neither the dev container (0) nor a `gpurun` box (1) has: every test
here self-skips below that, and runs unchanged on a multi-GPU node (`python -m pytest tests -m gpu`)."""
import os

import pytest
import torch

from distributedpytorch_amd import launch
from distributedpytorch_amd.utils import testing as T

pytestmark = pytest.mark.gpu


def _need(n):
    if not torch.cuda.is_available() or torch.cuda.device_count() < n:
        pytest.skip(f"needs {n} GPUs, found {torch.cuda.device_count() if torch.cuda.is_available() else 0}")


def _cpu_reference_mean_grads(world):
    from distributedpytorch_amd.models import mnist
    model, loss_fn, shape, classes = T._make_model("mnist")
    x, y = T._data(shape, classes, 8 * world)
    per = 8
    acc = None
    for r in range(world):
        model.zero_grad()
        loss_fn(model, x[r * per:(r + 1) * per], y[r * per:(r + 1) * per]).backward()
        g = {n: p.grad.clone() for n, p in model.named_parameters()}
        acc = g if acc is None else {n: acc[n] + g[n] for n in g}
    return {n: v / world for n, v in acc.items()}


@pytest.mark.parametrize("native,direct,wire", [(False, False, "fp32"), (True, False, "fp32"), (True, True, "fp32"),
                                                (True, True, "bf16")])
def test_dp_over_rccl_matches_reference(tmp_path, native, direct, wire, monkeypatch):
    """Mean-of-gradients semantics (reference DP.py:118-129) over RCCL: c10d path, native communicator
    (ring) and native communicator with the direct one-shot all-reduce."""
    _need(2)
    world = min(4, torch.cuda.device_count())
    monkeypatch.setenv("DPA_COMM", "native" if native else "c10d")
    monkeypatch.setenv("DPA_ALLREDUCE", "direct" if direct else "ring")
    monkeypatch.setenv("DPA_GRAD_WIRE", wire)
    tol = dict(rtol=1e-3, atol=1e-5) if wire == "fp32" else dict(rtol=2e-2, atol=1e-3)     # bf16 on the wire
    url = f"tcp://127.0.0.1:{T.free_port()}"
    launch.spawn(T.dp_worker, [world], 0, args=(url, "mnist", str(tmp_path), 1024, 3, "hooks", "nccl"))
    ref = _cpu_reference_mean_grads(world)
    for r in range(world):
        got = torch.load(tmp_path / f"grads_{r}.pt", weights_only=True)
        for name, g in ref.items():
            assert torch.allclose(got[name], g, **tol), f"{name} on rank {r}"
    finals = [torch.load(tmp_path / f"final_{r}.pt", weights_only=True) for r in range(world)]
    for r in range(1, world):
        for name in finals[0]["params"]:
            assert torch.equal(finals[0]["params"][name], finals[r]["params"][name]), f"replicas diverged at {name}"


def test_bench_two_ranks(tmp_path):
    """bench.py under torchrun on 2 GPUs prints one valid JSON line (the driver's contract)."""
    _need(2)
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(T.free_port()), os.path.join(root, "bench.py"),
                          "--gpus", "2", "--model", "tiny_resnet", "--batch", "32", "--steps", "3", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = [ln for ln in out.stdout.splitlines() if ln.startswith("{")][-1]
    rec = json.loads(line)
    assert rec["n_gpus"] == 2 and rec["value"] > 0 and rec["config"]["parallelism"] == "dp2"


def _native_group_worker(local_rank, rank, world, url, out_dir):
    """A job WITHOUT torch.distributed: store rendezvous, native RCCL collectives, ncclCommSplit sub-groups, P2P."""
    import torch
    from distributedpytorch_amd import comm as dcomm
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    c = dcomm.init_native_group(url, rank, world, dev)
    t = torch.full((1 << 20,), float(rank + 1), device=dev)
    c.all_reduce(t)
    assert float(t[0]) == world * (world + 1) / 2
    b = torch.full((1000,), float(rank), device=dev)
    c.broadcast(b, src=world - 1)
    assert float(b[0]) == world - 1
    objs = c.gather_object({"rank": rank}, dst=0)
    if rank == 0:
        assert [o["rank"] for o in objs] == list(range(world))
    assert c.broadcast_object("hello" if rank == 0 else None, src=0) == "hello"
    # sub-groups of two neighbours by ncclCommSplit; ring exchange inside each by grouped send/recv
    pairs = [[i, i + 1] for i in range(0, world - world % 2, 2)]
    mine = None
    for pr in pairs:
        sub = c.split(pr)
        if sub is not None:
            mine = sub
    if mine is not None:
        peer = 1 - mine.rank
        send = torch.full((4096,), float(rank), device=dev)
        recv = torch.empty(4096, device=dev)
        for w in mine.batch_p2p([("send", send, peer), ("recv", recv, peer)]):
            w.wait()
        assert float(recv[0]) == mine.to_global(peer)
        s2 = torch.ones(8, device=dev)
        mine.all_reduce(s2)
        assert float(s2[0]) == 2.0
    c.barrier()
    torch.save({"ok": True}, f"{out_dir}/native_{rank}.pt")
    c.close()


def test_native_group_without_torch_distributed(tmp_path):
    """init_native_group: TCP store rendezvous (store.py) + csrc/rccl_comm.cpp only - all-reduce, broadcast, object
    helpers over the store, ncclCommSplit sub-communicators and grouped point-to-point."""
    _need(2)
    world = min(4, torch.cuda.device_count())
    url = f"tcp://127.0.0.1:{T.free_port()}"
    launch.spawn(_native_group_worker, [world], 0, args=(url, str(tmp_path)))
    for r in range(world):
        assert (tmp_path / f"native_{r}.pt").exists()
