"""Failure and resume paths of the training loops: a rank that raises mid-epoch exits at once with
its own traceback (no barrier on the error path), idle ranks of a model-parallel job do not sit in a timed
collective, model-parallel and AdamW runs resume with their optimizer state."""
import time

import pytest
import torch

from distributedpytorch_amd import launch
from distributedpytorch_amd.config import TrainConfig
from distributedpytorch_amd.utils import testing as T


def _cfg(url, **kw):
    base = dict(model="mnist_cnn", backend="gloo", dist_url=url, device="cpu", epochs=1, batch_size=16, dataset_size=64,
                lr=0.05, momentum=0.5, graph=False, workers=1)
    base.update(kw)
    return TrainConfig(**base)


def _failing_dp(local_rank, global_rank, world_size, cfg):
    from distributedpytorch_amd import train
    from distributedpytorch_amd.parallel import dp as dpmod
    if global_rank == 1:
        orig, calls = dpmod.DataParallel.finish, {"n": 0}

        def finish(self):
            calls["n"] += 1
            if calls["n"] == 2:
                raise RuntimeError("injected failure in step 2")
            return orig(self)
        dpmod.DataParallel.finish = finish
    train.run_dp(local_rank, global_rank, world_size, cfg)


def test_a_rank_that_raises_mid_epoch_takes_the_job_down_promptly_with_its_traceback():
    url = f"tcp://127.0.0.1:{T.free_port()}"
    t0 = time.time()
    with pytest.raises(launch.ChildFailed) as e:
        launch.spawn(_failing_dp, [2], 0, args=(_cfg(url, timeout_s=120.0),))
    assert time.time() - t0 < 60, "the failing rank must not wait in a barrier for the process-group timeout"
    assert "injected failure in step 2" in str(e.value)


def _slow_mp(local_rank, global_rank, world_size, cfg):
    from distributedpytorch_amd import train
    from distributedpytorch_amd.parallel import pp as ppmod
    orig = ppmod.PipelineStage.train_step

    def slow(self, *a, **k):
        time.sleep(1.5)
        return orig(self, *a, **k)
    ppmod.PipelineStage.train_step = slow
    train.run_mp(local_rank, global_rank, world_size, cfg)


def test_idle_ranks_of_a_model_parallel_job_do_not_time_out_the_run():
    """--GPU-list=3 with 2 stages (the reference leaves ranks >= 2 idle, MP.py:118-124): the run outlasts the
    process-group timeout; the idle rank must have left instead of waiting in a timed barrier."""
    url = f"tcp://127.0.0.1:{T.free_port()}"
    cfg = _cfg(url, timeout_s=4.0, stages=2, micro_batches=2, schedule="1f1b", momentum=0.0)
    launch.spawn(_slow_mp, [3], 0, args=(cfg,))          # 4 steps x 1.5 s > 4 s; raises ChildFailed on a timeout


def test_model_parallel_run_resumes_epoch_and_per_stage_optimizer_state(tmp_path):
    url = f"tcp://127.0.0.1:{T.free_port()}"
    ck = str(tmp_path / "mp.pt")
    from distributedpytorch_amd import train
    cfg = _cfg(url, stages=2, micro_batches=2, momentum=0.5, checkpoint=ck, epochs=2)
    launch.spawn(train.run_mp, [2], 0, args=(cfg,))
    payload = torch.load(ck, weights_only=True)
    assert payload["epoch"] == 1 and isinstance(payload["optimizer"], list) and len(payload["optimizer"]) == 2
    assert all(len(o["state"]) > 0 for o in payload["optimizer"])
    # resume for one more epoch: starts at epoch 2 (not 0) with the momentum restored
    url2 = f"tcp://127.0.0.1:{T.free_port()}"
    cfg2 = _cfg(url2, stages=2, micro_batches=2, momentum=0.5, checkpoint=ck, resume=ck, epochs=3)
    launch.spawn(train.run_mp, [2], 0, args=(cfg2,))
    payload2 = torch.load(ck, weights_only=True)
    assert payload2["epoch"] == 2
    s1, s2 = payload["optimizer"][0]["steps"], payload2["optimizer"][0]["steps"]
    assert s2 == s1 + s1 // 2, f"optimizer step count must continue across the resume ({s1} -> {s2})"


def test_adamw_state_dict_round_trip_matches_torch_layout_and_continues_identically():
    from distributedpytorch_amd.models import mnist
    from distributedpytorch_amd.optim import FusedAdamW
    from distributedpytorch_amd.parallel.flat import FlatParams

    def make():
        torch.manual_seed(3)
        m = mnist.Net().eval()
        f = FlatParams(m, use_lp=False)
        return m, f, FusedAdamW(f, lr=1e-2, weight_decay=0.01)
    x = torch.randn(8, 1, 28, 28); y = torch.randint(0, 10, (8,))
    m1, f1, o1 = make()
    for _ in range(3):
        mnist.nll_loss(m1(x), y).backward(); o1.step()
    sd = o1.state_dict()
    assert set(sd["state"][0]) == {"step", "exp_avg", "exp_avg_sq"} and sd["param_groups"][0]["betas"] == (0.9, 0.95)
    m2, f2, o2 = make()
    f2.master.copy_(f1.master)
    o2.load_state_dict(sd)
    assert o2.steps == 3
    for m, o in ((m1, o1), (m2, o2)):
        mnist.nll_loss(m(x), y).backward(); o.step()
    assert torch.equal(f1.master, f2.master)          # same bias correction, same moments: bit-identical next step


def test_clip_coefficient_folded_into_the_optimizer_equals_scaling_the_buffer():
    from distributedpytorch_amd.models import mnist
    from distributedpytorch_amd.optim import FusedSGD, clip_grad_norm
    from distributedpytorch_amd.parallel.flat import FlatParams
    x = torch.randn(8, 1, 28, 28); y = torch.randint(0, 10, (8,))
    outs = []
    for folded in (True, False):
        torch.manual_seed(5)
        m = mnist.Net().eval()
        f = FlatParams(m, use_lp=False)
        o = FusedSGD(f, lr=0.1, momentum=0.5)
        for _ in range(2):
            (mnist.nll_loss(m(x), y) * 50.0).backward()
            norm = clip_grad_norm(f, 1.0, 1.0, optimizer=o if folded else None)
            assert float(norm) > 1.0
            o.step()
        outs.append(f.master.clone())
    torch.testing.assert_close(outs[0], outs[1], rtol=1e-6, atol=1e-7)


def test_breakdown_flag_writes_phase_times_to_the_jsonl(tmp_path):
    """the TRAINING loops (not only bench.py) report forward / backward / exposed-communication / optimizer
    milliseconds per step, and the counters of the reductions that were really issued."""
    import json
    from distributedpytorch_amd import train
    url = f"tcp://127.0.0.1:{T.free_port()}"
    logf = str(tmp_path / "dp.jsonl")
    launch.spawn(train.run_dp, [2], 0, args=(_cfg(url, breakdown=True, log_json=logf, epochs=2),))
    rows = [json.loads(l) for l in open(logf)]
    assert len(rows) == 2 and rows[0]["kind"] == "dp_epoch"
    for r in rows:
        for k in ("forward_ms", "backward_ms", "comm_exposed_ms", "optimizer_ms", "ms_per_step"):
            assert r[k] > 0.0, (k, r)
        assert r["allreduce_calls"] >= 2 and r["allreduce_bytes"] > 0          # world 2: one bucket per step, really reduced
    url = f"tcp://127.0.0.1:{T.free_port()}"
    logm = str(tmp_path / "mp.jsonl")
    launch.spawn(train.run_mp, [2], 0, args=(_cfg(url, stages=2, micro_batches=2, momentum=0.0, breakdown=True, log_json=logm),))
    rows = [json.loads(l) for l in open(logm)]
    kinds = {r["kind"] for r in rows}
    assert kinds == {"mp_epoch", "mp_stage"}                                  # the last stage's epoch line + the other stage's times
    for r in rows:
        assert r["pipeline_step_ms"] > 0.0 and r["optimizer_ms"] > 0.0 and r["forward_ms"] > 0.0 and r["backward_ms"] > 0.0, r


def _mp_with_fake_costs(local_rank, global_rank, world_size, cfg, shift):
    """run_mp whose cost model is skewed by ``shift``: stands in for a wall-clock measurement that came out
    differently on the resumed run."""
    from distributedpytorch_amd import train
    from distributedpytorch_amd.parallel import pp as ppmod
    orig = ppmod.param_costs
    train.param_costs = lambda layers: [c * (10.0 if i == shift else 1.0) for i, c in enumerate(orig(layers))]
    train.run_mp(local_rank, global_rank, world_size, cfg)


def test_pipeline_resume_cuts_the_stages_where_the_checkpoint_did(tmp_path):
    """the stage split comes from measurements that vary between runs, optimizer state is stored per
    stage by stage-local index - so the bounds travel in the checkpoint and a resumed run reuses them."""
    ck = str(tmp_path / "pp.pt")
    kw = dict(model="tiny_resnet", stages=2, micro_batches=2, momentum=0.9, batch_size=8, dataset_size=16, checkpoint=ck)
    launch.spawn(_mp_with_fake_costs, [2], 0, args=(_cfg(f"tcp://127.0.0.1:{T.free_port()}", epochs=1, **kw), 0))
    p1 = torch.load(ck, weights_only=True)
    b1 = p1["extra"]["stage_bounds"]
    assert p1["extra"]["stages"] == 2 and b1[0][0] == 0 and b1[0][1] == b1[1][0]
    # the resumed run's own cost model would cut elsewhere ...
    launch.spawn(_mp_with_fake_costs, [2], 0, args=(_cfg(f"tcp://127.0.0.1:{T.free_port()}", epochs=1, **dict(kw, checkpoint=str(tmp_path / "other.pt"))), 2))
    assert torch.load(tmp_path / "other.pt", weights_only=True)["extra"]["stage_bounds"] != b1
    # ... but with --resume it keeps the checkpoint's, and the per-stage momentum lands on the parameters it belongs to
    launch.spawn(_mp_with_fake_costs, [2], 0, args=(_cfg(f"tcp://127.0.0.1:{T.free_port()}", epochs=2, resume=ck, **kw), 2))
    p2 = torch.load(ck, weights_only=True)
    assert p2["extra"]["stage_bounds"] == b1 and p2["epoch"] == 1
    assert p2["optimizer"][0]["steps"] == 2 * p1["optimizer"][0]["steps"]


def test_gpt2_pipeline_checkpoint_with_a_bare_embedding_stage_keeps_the_dropout_streams(tmp_path):
    """a stage that is ONE bare layer (gpt2_tiny over 4 stages: the Embedding) exposes GPT-2's
    per-module dropout call counter as ``module._drop``; the checkpoint writer handled only the MNIST nets' step
    counter there.  All stages' dropout positions are saved and restored."""
    ck = str(tmp_path / "g.pt")
    kw = dict(model="gpt2_tiny", stages=4, micro_batches=2, momentum=0.0, batch_size=4, dataset_size=8, dropout=0.1,
              checkpoint=ck, lr=0.01)
    from distributedpytorch_amd import train
    launch.spawn(train.run_mp, [4], 0, args=(_cfg(f"tcp://127.0.0.1:{T.free_port()}", epochs=1, **kw),))
    p = torch.load(ck, weights_only=True)
    calls = p["dropout_calls"]
    assert len(calls) >= 3 and all(v > 0 for v in calls.values()), calls      # embedding + attention / MLP sites of every block
    launch.spawn(train.run_mp, [4], 0, args=(_cfg(f"tcp://127.0.0.1:{T.free_port()}", epochs=2, resume=ck, **kw),))
    p2 = torch.load(ck, weights_only=True)
    assert p2["epoch"] == 1 and all(p2["dropout_calls"][k] == 2 * v for k, v in calls.items()), (calls, p2["dropout_calls"])
