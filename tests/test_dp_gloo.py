"""T1: data-parallel engine over the gloo loopback backend, world_size 2 and 4.
Asserts the reference's semantics (DP.py:118-129): after the all-reduce every rank holds
the MEAN of the per-rank gradients, replicas stay identical, loss goes down."""
import os

import pytest
import torch

from distributedpytorch_amd import launch
from distributedpytorch_amd.utils import testing as T


def _reference_mean_grads(model_name, world):
    model, loss_fn, shape, classes = T._make_model(model_name)
    x, y = T._data(shape, classes, 8 * world)
    per = 8
    acc = None
    for r in range(world):
        model.zero_grad()
        loss_fn(model, x[r * per:(r + 1) * per], y[r * per:(r + 1) * per]).backward()
        g = {n: p.grad.clone() for n, p in model.named_parameters()}
        acc = g if acc is None else {n: acc[n] + g[n] for n in g}
    return {n: v / world for n, v in acc.items()}


@pytest.mark.parametrize("model_name,world,bucket_kb,mode", [("mnist", 2, 1024, "hooks"), ("mnist", 4, 8, "hooks"),
                                                             ("tiny_resnet", 2, 16, "hooks"), ("mnist", 2, 1024, "reduce_all")])
def test_dp_gradients_are_the_mean_and_replicas_agree(tmp_path, model_name, world, bucket_kb, mode):
    url = f"tcp://127.0.0.1:{T.free_port()}"
    launch.spawn(T.dp_worker, [world], 0, args=(url, model_name, str(tmp_path), bucket_kb, 4, mode))
    ref = _reference_mean_grads(model_name, world)
    grads = [torch.load(tmp_path / f"grads_{r}.pt", weights_only=True) for r in range(world)]
    for name, g in ref.items():
        for r in range(world):
            assert torch.allclose(grads[r][name], g, rtol=1e-4, atol=1e-6), f"{name} on rank {r}"
    finals = [torch.load(tmp_path / f"final_{r}.pt", weights_only=True) for r in range(world)]
    for r in range(1, world):
        for name in finals[0]["params"]:
            assert torch.equal(finals[0]["params"][name], finals[r]["params"][name]), f"replicas diverged at {name}"
    if model_name == "mnist":
        if bucket_kb >= 1024:
            assert finals[0]["buckets"] == 1 and finals[0]["calls"] == 4      # ONE all-reduce per step (reference: 8)
        else:
            assert finals[0]["buckets"] > 1
        # per-rank loss on a fixed shard must go down over 4 SGD steps
        assert finals[0]["losses"][-1] < finals[0]["losses"][0]


def test_baseline_config1_small_cnn_dp_gloo_world2(tmp_path, capfd):
    """BASELINE.json config 1: small CNN, synthetic MNIST-shape data, DP via spawn on CPU/gloo, world 2."""
    from distributedpytorch_amd.examples import dp
    port = T.free_port()
    ck = tmp_path / "ck.pt"
    dp.main([f"--dist-url=tcp://127.0.0.1:{port}", "--backend=gloo", "--node-index=0", "--GPU-list=2", "--epochs=2",
             "--dataset-size=512", "--steps-per-epoch=4", f"--checkpoint={ck}", f"--log-json={tmp_path / 'log.jsonl'}"])
    out = capfd.readouterr().out
    assert "Rank  0  , epoch  0  :" in out and "Rank  1  , epoch  1  :" in out       # the reference's per-rank epoch line
    sd = torch.load(str(ck) + ".model", weights_only=True)
    assert list(sd.keys()) == ["conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"]
    assert os.path.exists(tmp_path / "log.jsonl")


@pytest.mark.parametrize("model_name,world,bucket_kb", [("mnist", 2, 16), ("mnist", 3, 64), ("tiny_resnet", 2, 256)])
def test_sharded_optimizer_matches_the_replicated_one(tmp_path, model_name, world, bucket_kb):
    """ZeRO stage 1 (DataParallel(shard_optimizer=True) + ShardedSGD): gradients reduced onto the shard owner, the
    owner updates its slice, parameters broadcast back - after 4 steps with momentum and per-segment weight decay the
    replicas equal those of the replicated FusedSGD run, each rank holds 1/N of the momentum, and the gathered
    optimizer state_dict equals the replicated one."""
    a, b = tmp_path / "zero", tmp_path / "ref"
    a.mkdir(); b.mkdir()
    launch.spawn(T.dp_worker, [world], 0, args=(f"tcp://127.0.0.1:{T.free_port()}", model_name, str(a), bucket_kb, 4, "zero"))
    launch.spawn(T.dp_worker, [world], 0, args=(f"tcp://127.0.0.1:{T.free_port()}", model_name, str(b), bucket_kb, 4, "zero_ref"))
    z = [torch.load(a / f"final_{r}.pt", weights_only=True) for r in range(world)]
    ref = torch.load(b / "final_0.pt", weights_only=True)
    total = ref["momentum_numel"]
    assert sum(f["momentum_numel"] for f in z) == total and max(f["momentum_numel"] for f in z) <= -(-total // world) + 64
    for name, p in ref["params"].items():
        for r in range(world):
            # a reduce tree and an all-reduce ring may add the ranks' contributions in different orders
            assert torch.allclose(z[r]["params"][name], p, rtol=1e-5, atol=1e-7), (name, r)
        assert torch.equal(z[0]["params"][name], z[world - 1]["params"][name])       # replicas bit-identical
    for i, st in ref["opt"]["state"].items():
        assert torch.allclose(z[0]["opt"]["state"][i]["momentum_buffer"], st["momentum_buffer"], rtol=1e-5, atol=1e-7)
    assert z[0]["losses"][-1] < z[0]["losses"][0]


@pytest.mark.parametrize("extra", [["--accum-steps=2", "--clip-grad-norm=1.0", "--warmup-steps=3"],
                                   ["--optimizer=adamw", "--lr=0.002", "--accum-steps=3", "--clip-grad-norm=0.5"],
                                   ["--shard-optimizer", "--clip-grad-norm=1.0", "--cosine-steps=16", "--min-lr=0.001"]])
def test_training_loop_options_world2(tmp_path, capfd, extra):
    """Gradient accumulation (no traffic on the accumulate-only micro-steps), global-norm clipping (shard-aware),
    warm-up / cosine LR, AdamW and the sharded optimizer through the example CLI on gloo, world 2: the loop runs, the
    loss goes down, and an accumulating run issues 1/accum of the all-reduces."""
    import json
    import re
    from distributedpytorch_amd.examples import dp
    log = tmp_path / "log.jsonl"
    dp.main([f"--dist-url=tcp://127.0.0.1:{T.free_port()}", "--backend=gloo", "--node-index=0", "--GPU-list=2", "--epochs=3",
             "--dataset-size=1024", "--steps-per-epoch=8", "--lr=0.05", "--momentum=0.9", f"--log-json={log}"] + extra)
    out = capfd.readouterr().out
    losses = [float(v) for v in re.findall(r"Rank  0  , epoch  \d+  :  ([0-9.]+)", out)]
    assert len(losses) == 3 and losses[-1] < losses[0], losses
    assert all(json.loads(l)["kind"] == "dp_epoch" for l in open(log))


def test_peer_mapped_allreduce_declines_together_without_a_gpu(tmp_path):
    """Communicator.attach_peer (DPA_ALLREDUCE=ipc, csrc/peer_comm.cpp) on CPU ranks: no GPU, so every rank must fall
    back TOGETHER - decided on the control plane - and collectives keep working over gloo."""
    world = 2
    launch.spawn(T.peer_refusal_worker, [world], 0, args=(f"tcp://127.0.0.1:{T.free_port()}", str(tmp_path)))
    outs = [torch.load(tmp_path / f"refusal_{r}.pt", weights_only=True) for r in range(world)]
    assert not any(o["attached"] for o in outs)
    assert all(torch.equal(o["sum"], torch.full((5,), 3.0)) for o in outs)
