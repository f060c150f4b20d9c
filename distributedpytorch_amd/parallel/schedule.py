"""Pipeline schedules as pure functions (unit-tested without any process group).

A schedule is the ordered list of operations ONE stage executes for one training step:

    ("recv_fwd", mb)  receive micro-batch mb's activation from stage-1
    ("F", mb)         forward micro-batch mb
    ("send_fwd", mb)  send its activation to stage+1
    ("recv_bwd", mb)  receive d(loss)/d(activation) from stage+1
    ("B", mb)         backward micro-batch mb
    ("send_bwd", mb)  send d(loss)/d(input) to stage-1

Communication ops that are adjacent in the list are issued as ONE grouped RCCL call
(``ncclGroupStart/End`` through ``batch_isend_irecv``), which is what makes the
send-forward/receive-backward exchanges between neighbouring stages deadlock-free.

* ``serial``  - the reference's schedule (multinode-MP/multinodeMP.py:130-167): one batch,
  stage k idles while stage k+1 works.
* ``gpipe``   - all forwards, then all backwards (bubble (S-1)/(M+S-1), activations of
  all M micro-batches live at once).
* ``1f1b``    - warm-up of S-1-stage forwards, then one-forward-one-backward, then
  cool-down; same bubble as GPipe but at most S-stage micro-batches of activations
  live, which is what lets stages be sized against 288 GB of HBM.
"""
from __future__ import annotations

from typing import List, Tuple

Op = Tuple[str, int]
COMM_OPS = ("recv_fwd", "send_fwd", "recv_bwd", "send_bwd")


def build_schedule(kind: str, n_stages: int, stage: int, n_micro: int) -> List[Op]:
    if not 0 <= stage < n_stages:
        raise ValueError("stage out of range")
    if n_micro < 1:
        raise ValueError("need at least one micro-batch")
    first, last = stage == 0, stage == n_stages - 1
    ops: List[Op] = []

    def fwd(mb):
        if not first:
            ops.append(("recv_fwd", mb))
        ops.append(("F", mb))
        if not last:
            ops.append(("send_fwd", mb))

    def bwd(mb):
        if not last:
            ops.append(("recv_bwd", mb))
        ops.append(("B", mb))
        if not first:
            ops.append(("send_bwd", mb))

    kind = kind.lower()
    if kind == "serial":
        for mb in range(n_micro):
            fwd(mb)
            bwd(mb)
    elif kind == "gpipe":
        for mb in range(n_micro):
            fwd(mb)
        for mb in range(n_micro):
            bwd(mb)
    elif kind == "1f1b":
        warm = min(n_stages - stage - 1, n_micro)
        steady = n_micro - warm
        for mb in range(warm):
            fwd(mb)
        if steady > 0 and not first:
            ops.append(("recv_fwd", warm))
        for i in range(steady):
            f_mb, b_mb = warm + i, i
            ops.append(("F", f_mb))
            if not last:
                ops.append(("send_fwd", f_mb))      # grouped with ...
                ops.append(("recv_bwd", b_mb))      # ... this receive
            ops.append(("B", b_mb))
            if not first:
                ops.append(("send_bwd", b_mb))      # grouped with the next forward's receive
                if i + 1 < steady:
                    ops.append(("recv_fwd", f_mb + 1))
        for mb in range(steady, n_micro):
            bwd(mb)
    else:
        raise ValueError(f"unknown schedule {kind!r}")
    return ops


def group_comm(ops: List[Op]) -> List[List[Op]]:
    """Split a schedule into steps: each compute op alone, maximal runs of comm ops together."""
    steps: List[List[Op]] = []
    run: List[Op] = []
    for op in ops:
        if op[0] in COMM_OPS:
            run.append(op)
        else:
            if run:
                steps.append(run)
                run = []
            steps.append([op])
    if run:
        steps.append(run)
    return steps


def validate(kind: str, n_stages: int, n_micro: int) -> None:
    """Check cross-stage consistency: every send has a matching recv in the same pairwise
    order, every micro-batch is forwarded before it is backwarded, and a lock-step
    simulation of all stages terminates (no deadlock with blocking receives)."""
    scheds = [build_schedule(kind, n_stages, s, n_micro) for s in range(n_stages)]
    for s in range(n_stages - 1):
        assert [mb for op, mb in scheds[s] if op == "send_fwd"] == [mb for op, mb in scheds[s + 1] if op == "recv_fwd"]
        assert [mb for op, mb in scheds[s + 1] if op == "send_bwd"] == [mb for op, mb in scheds[s] if op == "recv_bwd"]
    for s, sch in enumerate(scheds):
        seen_f = set()
        for op, mb in sch:
            if op == "F":
                seen_f.add(mb)
            if op == "B":
                assert mb in seen_f, f"stage {s}: backward of {mb} before forward"
        assert sorted(mb for op, mb in sch if op == "F") == list(range(n_micro))
        assert sorted(mb for op, mb in sch if op == "B") == list(range(n_micro))
    # simulation: sends are asynchronous (buffered), receives block
    pc = [0] * n_stages
    sent = set()
    progressed = True
    while progressed:
        progressed = False
        for s in range(n_stages):
            while pc[s] < len(scheds[s]):
                op, mb = scheds[s][pc[s]]
                if op == "recv_fwd" and ("f", s - 1, mb) not in sent:
                    break
                if op == "recv_bwd" and ("b", s + 1, mb) not in sent:
                    break
                if op == "send_fwd":
                    sent.add(("f", s, mb))
                if op == "send_bwd":
                    sent.add(("b", s, mb))
                pc[s] += 1
                progressed = True
    assert all(pc[s] == len(scheds[s]) for s in range(n_stages)), f"deadlock: {pc}"


def bubble_fraction(n_stages: int, n_micro: int) -> float:
    """Idle fraction of a GPipe / 1F1B pipeline with equal-time stages."""
    return (n_stages - 1) / float(n_micro + n_stages - 1)


def peak_live_microbatches(kind: str, n_stages: int, stage: int, n_micro: int) -> int:
    live = peak = 0
    for op, _ in build_schedule(kind, n_stages, stage, n_micro):
        if op == "F":
            live += 1
            peak = max(peak, live)
        elif op == "B":
            live -= 1
    return peak
