"""The NATIVE layer's Python side at world size > 1, rehearsed without several GPUs: ``DPA_FAKE_NATIVE=1`` installs
utils/fake_native.GlooNative (the interface of csrc/rccl_comm.cpp's RcclComm, data carried by gloo) as
``Communicator.native``, so the start-up self-test with all its collective phases, the bucket all-reduces of the DP engine,
ZeRO-1's reduce / broadcast, ``ncclCommSplit``-style sub-communicators of build_grid and the pipeline engine's grouped
send / recv run through exactly the code paths a multi-GPU job takes - and must give what the c10d paths give."""
import torch

import test_dp_gloo as D
import test_pp_gloo as P


def _calls(path):
    return torch.load(path, weights_only=True)["native_calls"]


def test_dp_buckets_ride_the_native_layer(tmp_path, monkeypatch):
    monkeypatch.setenv("DPA_FAKE_NATIVE", "1")
    D.test_dp_gradients_are_the_mean_and_replicas_agree(tmp_path, "mnist", 4, 8, "hooks")
    calls = _calls(tmp_path / "final_1.pt")
    # self-test: two all-reduces, broadcast, reduce, reduce-scatter, all-gather, the send/recv ring, two splits at W = 4
    assert calls[:7] == ["all_reduce", "all_reduce", "broadcast", "reduce", "reduce_scatter", "all_gather", "batch_p2p"]
    assert calls.count("split") == 2
    assert calls.count("all_reduce") > 2 + 4                      # + the buckets of four steps


def test_zero1_reduce_and_broadcast_ride_the_native_layer(tmp_path, monkeypatch):
    monkeypatch.setenv("DPA_FAKE_NATIVE", "1")
    D.test_sharded_optimizer_matches_the_replicated_one(tmp_path, "mnist", 3, 64)
    calls = _calls(tmp_path / "zero" / "final_2.pt")
    assert calls.count("reduce") > 1 and calls.count("broadcast") > 1


def test_two_stage_pipeline_sends_through_grouped_native_p2p(tmp_path, monkeypatch):
    monkeypatch.setenv("DPA_FAKE_NATIVE", "1")
    P.test_two_stage_mnist_split_matches_unsplit(tmp_path, "1f1b", 4, 12)
    calls = _calls(tmp_path / "final_r0_s1.pt")
    assert calls["pp"].count("batch_p2p") >= 3 * 4                 # three steps, four micro-batches, both directions grouped


def test_dp_x_pp_grid_is_built_from_native_splits(tmp_path, monkeypatch):
    monkeypatch.setenv("DPA_FAKE_NATIVE", "1")
    P.test_dp_x_pp_hybrid_2x2(tmp_path)
    calls = _calls(tmp_path / "final_r1_s1.pt")
    assert calls["world"].count("split") >= 2 + 4                  # self-test (2) + two pipeline groups + two replica groups
    assert "batch_p2p" in calls["pp"] and "all_reduce" in calls["dp"]


def test_tied_embedding_pipeline_on_the_native_layer(tmp_path, monkeypatch):
    monkeypatch.setenv("DPA_FAKE_NATIVE", "1")
    P.test_gpt2_pipeline_keeps_the_embedding_and_the_lm_head_tied(tmp_path)
    assert "batch_p2p" in _calls(tmp_path / "final_r0_s1.pt")["pp"]
