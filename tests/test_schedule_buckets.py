"""T0: pure-function planners - pipeline schedules, stage partitioner, bucket planner."""
import pytest

from distributedpytorch_amd.parallel import schedule as S
from distributedpytorch_amd.parallel.flat import plan_buckets
from distributedpytorch_amd.parallel.pp import partition_balanced, split_microbatches


@pytest.mark.parametrize("kind", ["serial", "gpipe", "1f1b"])
@pytest.mark.parametrize("stages", [1, 2, 3, 4, 8])
@pytest.mark.parametrize("micro", [1, 2, 4, 7, 16])
def test_schedules_are_consistent_and_deadlock_free(kind, stages, micro):
    S.validate(kind, stages, micro)


def test_serial_schedule_is_the_reference_order():
    # stage 0: forward, send, wait for grad, backward (MP.py:133-146)
    assert S.build_schedule("serial", 2, 0, 1) == [("F", 0), ("send_fwd", 0), ("recv_bwd", 0), ("B", 0)]
    # stage 1: recv, forward, backward, send grad (MP.py:151-167)
    assert S.build_schedule("serial", 2, 1, 1) == [("recv_fwd", 0), ("F", 0), ("B", 0), ("send_bwd", 0)]


def test_1f1b_bounds_live_activations():
    for stages in (2, 4, 8):
        for stage in range(stages):
            assert S.peak_live_microbatches("1f1b", stages, stage, 32) <= stages - stage
            assert S.peak_live_microbatches("gpipe", stages, stage, 32) == 32
    assert abs(S.bubble_fraction(4, 8) - 3 / 11) < 1e-12


def test_1f1b_groups_send_with_recv():
    steps = S.group_comm(S.build_schedule("1f1b", 4, 1, 8))
    assert [("send_fwd", 2), ("recv_bwd", 0)] in steps or any(len(s) == 2 and s[0][0] == "send_fwd" and s[1][0] == "recv_bwd" for s in steps)
    assert any(len(s) == 2 and s[0][0] == "send_bwd" and s[1][0] == "recv_fwd" for s in steps)


@pytest.mark.parametrize("kind", ["serial", "gpipe", "1f1b"])
def test_grouped_communication_is_deadlock_free_under_rendezvous_semantics(kind):
    """ncclSend / ncclRecv are rendezvous operations on ONE in-order stream per rank (``validate`` models buffered
    sends).  The groups the engine issues (``build_steps``) survive the strictest model - whole groups must pair up -
    for every S in 2..8 and M in 1..16, and so also the model in which operations of a group complete one by one."""
    for stages in range(2, 9):
        for micro in range(1, 17):
            S.validate_rendezvous(kind, stages, micro, strict=True)
            S.validate_rendezvous(kind, stages, micro, strict=False)
            for stage in range(stages):      # same operations in the same order as the plain schedule: only the grouping differs
                flat = [op for g in S.build_steps(kind, stages, stage, micro) for op in g]
                assert flat == S.build_schedule(kind, stages, stage, micro)


def test_rendezvous_simulator_catches_the_groupings_it_is_there_to_catch():
    # round 2's "maximal runs" grouping pairs a send to one neighbour with a receive from the other (and, in the serial
    # schedule, a send with a receive the peer issues two groups apart): fine while operations of a group progress
    # independently, a deadlock if whole groups had to pair up
    steps = [S.group_comm(S.build_schedule("serial", 2, s, 1)) for s in range(2)]
    S.validate_rendezvous("serial", 2, 1, strict=False, steps=steps)
    with pytest.raises(AssertionError, match="rendezvous deadlock"):
        S.validate_rendezvous("serial", 2, 1, strict=True, steps=steps)
    # two neighbours that both send first, as separate calls: a deadlock under either model
    bad = [[[("send_fwd", 0)], [("recv_bwd", 0)]], [[("send_bwd", 0)], [("recv_fwd", 0)]]]
    for strict in (True, False):
        with pytest.raises(AssertionError, match="rendezvous deadlock"):
            S.validate_rendezvous("1f1b", 2, 1, strict=strict, steps=bad)
    # 1F1B's steady-state exchange stays ONE group on both sides (both send first: it must be)
    a = S.build_steps("1f1b", 4, 1, 8)
    b = S.build_steps("1f1b", 4, 2, 8)
    assert [("send_fwd", 2), ("recv_bwd", 0)] in a and [("send_bwd", 0), ("recv_fwd", 2)] in b
    # ... and no group spans two peers
    for st in range(4):
        for g in S.build_steps("1f1b", 4, st, 8):
            if g[0][0] in S.COMM_OPS:
                assert len({S._peer_op(op, st)[0] for op in g}) == 1


def test_partition_balanced():
    assert partition_balanced([1, 1, 1, 1], 2) == [(0, 2), (2, 4)]
    b = partition_balanced([10, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1], 2)
    assert b == [(0, 1), (1, 11)]
    costs = [3, 1, 4, 1, 5, 9, 2, 6, 5, 3, 5]
    for k in (1, 2, 3, 4, 11):
        b = partition_balanced(costs, k)
        assert b[0][0] == 0 and b[-1][1] == len(costs) and all(b[i][1] == b[i + 1][0] for i in range(k - 1))
        assert all(e > s for s, e in b)
    with pytest.raises(ValueError):
        partition_balanced([1, 2], 3)


def test_split_microbatches_ragged():
    assert split_microbatches(256, 8) == [32] * 8
    assert split_microbatches(96, 8) == [12] * 8            # reference's ragged last batch at B=256
    assert split_microbatches(12, 8) == [2, 2, 2, 2, 1, 1, 1, 1]
    assert split_microbatches(3, 8) == [1, 1, 1]
    assert sum(split_microbatches(608, 7)) == 608


def test_plan_buckets():
    numels = [10, 10, 64, 500, 5, 5]
    offsets, cur = [], 0
    for n in numels:
        offsets.append(cur); cur += (n + 63) // 64 * 64
    order = list(range(len(numels)))
    # everything fits: ONE bucket (the small CNN's 87 KB of grads -> one all-reduce)
    b = plan_buckets(numels, offsets, order, cur, bucket_bytes=1 << 20)
    assert len(b) == 1 and b[0].start == 0 and b[0].end == cur and b[0].param_indices == order
    # small cap: buckets close on parameter boundaries, cover the buffer exactly, in order
    b = plan_buckets(numels, offsets, order, cur, bucket_bytes=100 * 4, first_bucket_bytes=12 * 4)
    assert b[0].param_indices == [0]
    assert [x.start for x in b][0] == 0 and b[-1].end == cur
    assert all(b[i].end == b[i + 1].start for i in range(len(b) - 1))
    assert sorted(sum((x.param_indices for x in b), [])) == order


def test_plan_buckets_caps_the_last_bucket_too():
    """The final bucket belongs to the first layers: nothing is left to overlap its collective with, so it is held
    to ``first_bucket_bytes`` like the first one (parallel/flat.py docstring)."""
    numels = [8, 300, 300, 300, 20, 20, 20, 6]           # flat (= backward) order; the tail is small parameters
    offsets, cur = [], 0
    for n in numels:
        offsets.append(cur); cur += (n + 63) // 64 * 64
    order = list(range(len(numels)))
    b = plan_buckets(numels, offsets, order, cur, bucket_bytes=700 * 4, first_bucket_bytes=50 * 4)
    assert b[0].param_indices == [0]
    assert b[-1].param_indices == [5, 6, 7], b           # 46 elements <= 50; adding parameter 4 would exceed the cap
    assert sum(numels[i] for i in b[-1].param_indices) <= 50
    assert b[-1].end == cur and all(b[i].end == b[i + 1].start for i in range(len(b) - 1))
    assert sum((x.param_indices for x in b), []) == order
    # a last parameter larger than the cap cannot be split off usefully: the plan is left alone
    b = plan_buckets([10, 10, 500], [0, 64, 128], [0, 1, 2], 640, bucket_bytes=1 << 20, first_bucket_bytes=12 * 4)
    assert [x.param_indices for x in b] == [[0], [1, 2]]


def test_two_flat_buffers_keep_their_own_gradient_sinks():
    """ops/sink.py holds no process-wide state: a second engine does not steal the first one's sink."""
    import torch
    from distributedpytorch_amd.ops import sink
    from distributedpytorch_amd.parallel.flat import FlatParams
    m1, m2 = torch.nn.Linear(4, 3), torch.nn.Linear(4, 3)
    f1, f2 = FlatParams(m1, use_lp=False), FlatParams(m2, use_lp=False)
    seen = []
    f1.install_sink(lambda p: seen.append(("f1", p)))
    f2.install_sink(lambda p: seen.append(("f2", p)))
    t1, t2 = sink.target(m1.weight), sink.target(m2.weight)
    assert t1 is not None and t2 is not None
    a1, _ = f1.slice_of(m1.weight)
    a2, _ = f2.slice_of(m2.weight)
    assert t1.data_ptr() == f1.grad[a1:].data_ptr() and t2.data_ptr() == f2.grad[a2:].data_ptr()
    sink.ready(m1.weight); sink.ready(m2.bias)
    assert seen == [("f1", m1.weight), ("f2", m2.bias)]
    f2.uninstall_sink()
    assert sink.target(m2.weight) is None and sink.target(m1.weight) is not None
    assert sink.target(torch.nn.Parameter(torch.zeros(2))) is None


def test_clip_grad_norm_and_lr_schedule():
    import math
    import torch
    from distributedpytorch_amd.optim import FusedSGD, WarmupCosine, clip_grad_norm
    from distributedpytorch_amd.parallel.flat import FlatParams
    torch.manual_seed(0)
    m = torch.nn.Sequential(torch.nn.Linear(8, 16), torch.nn.ReLU(), torch.nn.Linear(16, 4))
    flat = FlatParams(m, use_lp=False)
    m(torch.randn(5, 8)).square().sum().backward()
    ref = torch.nn.utils.clip_grad_norm_([p for p in m.parameters()], max_norm=0.1)      # stock, on the same views
    # undo (the stock call scaled in place), then ours with world-size-2 bookkeeping: buffer = 2 x mean gradient
    flat.grad.mul_(float(ref) / 0.1 if float(ref) > 0.1 else 1.0)
    before = flat.grad.clone()
    flat.grad.mul_(2.0)
    norm = clip_grad_norm(flat, 0.1, grad_scale=0.5)
    assert abs(float(norm) - float(before.norm())) < 1e-5
    assert abs(float(flat.grad.norm()) * 0.5 - 0.1) < 1e-4                  # the mean gradient now has norm 0.1
    sched = WarmupCosine(1.0, warmup=4, total=12, min_lr=0.1)
    assert [round(sched(s), 4) for s in range(4)] == [0.25, 0.5, 0.75, 1.0]
    assert abs(sched(4) - 1.0) < 1e-9 and abs(sched(8) - (0.1 + 0.45 * (1 + math.cos(math.pi * 0.5)))) < 1e-9
    assert sched(12) == 0.1 and sched(100) == 0.1
    opt = FusedSGD(flat, lr=1.0)
    assert sched.apply(opt, 1) == 0.5 and opt.lr == 0.5
