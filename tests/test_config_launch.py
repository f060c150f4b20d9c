"""T0/T1: flag parsing, node-topology rank math (reference DP.py:170-178), launcher."""
import os

import pytest

from distributedpytorch_amd import config, launch
from distributedpytorch_amd.utils import testing as T


def test_rank_math_matches_reference_formula():
    for gpu_list in ([4, 4], [4, 2, 1], [1], [2, 0, 3], [8]):
        for node in range(len(gpu_list)):
            topo = launch.topology(gpu_list, node)
            # reference: DP.py:171-174
            assert topo.local_size == gpu_list[node]
            assert topo.world_size == sum(gpu_list)
            assert topo.offset == sum(gpu_list[:node])
            for lr in range(gpu_list[node]):
                assert topo.global_rank(lr) == sum(gpu_list[:node]) + lr
                assert topo.node_of(topo.global_rank(lr)) == node
    ranks = [launch.topology([4, 2, 1], n).global_rank(l) for n, c in enumerate([4, 2, 1]) for l in range(c)]
    assert ranks == list(range(7))          # heterogeneous nodes tile the world exactly once


def test_rank_math_errors():
    with pytest.raises(ValueError):
        launch.topology([2, 2], 2)
    with pytest.raises(ValueError):
        launch.topology([2, 2], 0).global_rank(2)


def test_dp_flags_keep_reference_names():
    cfg = config.parse_dp(["--dist-url=tcp://10.0.0.1:8885", "--backend=nccl", "--node-index=1", "--GPU-list=4,4"])
    assert cfg.gpu_list == [4, 4] and cfg.node_index == 1 and cfg.world_size == 8 and cfg.rank_offset == 4
    assert cfg.dist_url == "tcp://10.0.0.1:8885" and cfg.backend == "nccl"
    assert (cfg.lr, cfg.momentum, cfg.batch_size, cfg.epochs, cfg.seed) == (0.01, 0.5, 128, 10, 1234)   # DP.py:109,138,143,146
    # flags the reference parses but never uses are accepted
    config.parse_dp(["--workers=4", "--gpu=0", "--rank=0", "--GPU-list=1"])


def test_dp_flag_defects_fixed():
    assert config.parse_dp(["--GPU-list=2", "--dist-url=127.0.0.1:8885"]).dist_url == "tcp://127.0.0.1:8885"  # DP.py:27
    assert config.parse_dp(["--GPU-list=2", "--epochs=3"]).epochs == 3                                          # DP.py:31 vs :146
    assert config.parse_dp([]).gpu_list == [1]                                                                  # DP.py:37 default crashes


def test_mp_flags_keep_reference_names():
    cfg = config.parse_mp(["--batch-size=256", "--backend=gloo", "--node-index=0", "--GPU-list=2", "--IP=127.0.0.1", "--portNum=39990"])
    assert cfg.dist_url == "tcp://127.0.0.1:39990" and cfg.batch_size == 256 and cfg.world_size == 2
    assert (cfg.lr, cfg.momentum, cfg.epochs) == (0.01, 0.0, 100)         # MP.py:120,124,126
    assert config.parse_mp([]).batch_size == 1024                         # MP.py:35


def test_bad_gpu_list():
    with pytest.raises(ValueError):
        config.parse_gpu_list("0")
    with pytest.raises(ValueError):
        config.parse_dp(["--GPU-list=2,2", "--node-index=5"])


def test_spawn_passes_ranks(tmp_path):
    launch.spawn(T.ok_worker, [1, 3], 1, args=(str(tmp_path),))
    got = sorted(open(tmp_path / f).read() for f in os.listdir(tmp_path))
    assert got == ["0 1 4 1 0", "1 2 4 2 1", "2 3 4 3 2"]


def test_spawn_kills_survivors_when_a_rank_dies():
    """Failure detection: the reference's parent would block in join() forever."""
    import time
    url = f"tcp://127.0.0.1:{T.free_port()}"
    t0 = time.time()
    with pytest.raises(launch.ChildFailed) as e:
        launch.spawn(T.crash_worker, [3], 0, args=(url,))
    assert time.time() - t0 < 60
    assert "injected failure" in str(e.value)


def test_gc_guard_disables_collector_and_collects_on_a_fixed_period():
    import gc
    from distributedpytorch_amd.utils.gcguard import GCGuard
    assert gc.isenabled()
    seen = []
    cb = lambda phase, info: seen.append(info["generation"]) if phase == "start" else None  # noqa: E731
    with GCGuard(period=3) as guard:
        assert not gc.isenabled()
        gc.callbacks.append(cb)
        try:
            for _ in range(7):
                guard.step()
        finally:
            gc.callbacks.remove(cb)
    assert gc.isenabled()
    assert seen == [1, 1]          # steps 3 and 6, nothing in between
