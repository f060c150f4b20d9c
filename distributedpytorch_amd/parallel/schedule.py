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

import functools
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


def _peer_op(op: Op, stage: int) -> Tuple[int, Op]:
    """(peer stage, the operation that matches ``op`` there)."""
    kind, mb = op
    if kind == "send_fwd":
        return stage + 1, ("recv_fwd", mb)
    if kind == "recv_fwd":
        return stage - 1, ("send_fwd", mb)
    if kind == "send_bwd":
        return stage - 1, ("recv_bwd", mb)
    if kind == "recv_bwd":
        return stage + 1, ("send_bwd", mb)
    raise ValueError(kind)


@functools.lru_cache(maxsize=256)
def _all_steps(kind: str, n_stages: int, n_micro: int) -> Tuple[Tuple[Tuple[Op, ...], ...], ...]:
    """Steps of EVERY stage with the communication grouped so that it is deadlock-free under the strictest reading of
    RCCL's grouped point-to-point (``validate_rendezvous(strict=True)``): a group is one kernel on the rank's single
    in-order communication stream, ``ncclSend`` / ``ncclRecv`` are rendezvous operations, and nothing is assumed about
    operations of one group completing independently of each other.  Rules:

    * a group only holds operations with ONE peer (a run of adjacent operations is cut where the peer changes, order
      kept): no group waits for two neighbours at once;
    * operations with the same peer stay grouped only when the peer issues exactly the mirrored operations as ONE
      group too (1F1B's steady state: ``send_fwd f + recv_bwd b`` here is ``recv_fwd f + send_bwd b`` there - the
      exchange that MUST be grouped, since both sides send first); otherwise they go out one by one, in program order,
      which is the order their matches are issued in on the peer."""
    raw = [group_comm(build_schedule(kind, n_stages, s, n_micro)) for s in range(n_stages)]
    steps: List[List[List[Op]]] = []
    for s in range(n_stages):
        out: List[List[Op]] = []
        for g in raw[s]:
            if g[0][0] not in COMM_OPS:
                out.append(list(g))
                continue
            run: List[Op] = []
            for op in g:                                   # cut where the peer changes
                if run and _peer_op(op, s)[0] != _peer_op(run[0], s)[0]:
                    out.append(run)
                    run = []
                run.append(op)
            if run:
                out.append(run)
        steps.append(out)
    changed = True
    while changed:                                         # un-group what the peer does not mirror as one group
        changed = False
        groups = [{frozenset(g) for g in steps[s] if g[0][0] in COMM_OPS} for s in range(n_stages)]
        for s in range(n_stages):
            new: List[List[Op]] = []
            for g in steps[s]:
                if g[0][0] in COMM_OPS and len(g) > 1:
                    peer = _peer_op(g[0], s)[0]
                    mirror = frozenset(_peer_op(op, s)[1] for op in g)
                    if mirror not in groups[peer]:
                        new.extend([op] for op in g)
                        changed = True
                        continue
                new.append(g)
            steps[s] = new
    return tuple(tuple(tuple(g) for g in st) for st in steps)


def build_steps(kind: str, n_stages: int, stage: int, n_micro: int) -> List[List[Op]]:
    """What ``PipelineStage`` executes: compute operations alone, communication in rendezvous-safe groups."""
    return [list(g) for g in _all_steps(kind.lower(), n_stages, n_micro)[stage]]


def validate_rendezvous(kind: str, n_stages: int, n_micro: int, strict: bool = True, steps=None) -> int:
    """Deadlock check under the semantics of RCCL point-to-point, which ``validate`` (buffered sends) does not model:
    ``ncclSend`` / ``ncclRecv`` are RENDEZVOUS operations - neither side completes before the other has arrived - and
    every rank issues its groups on ONE in-order communication stream.  Each stage is a queue of groups; only the
    group at the head of a queue is running.

    ``strict``: a running group retires only when EVERY operation in it finds its match inside the group running at
    the head of the peer's queue (whole kernels pair up).  Not strict: operations of a running group complete one by
    one as soon as their match is running on the peer (how RCCL's kernels behave today: the send half and the
    receive half of a group progress independently), the group retires when all of them have.

    Compute steps need no modelling: a group waits for the compute issued before it, and that compute only ever waits
    for EARLIER groups of the same stage, which the in-order queue already implies.
    Raises AssertionError naming the stuck heads on a deadlock; returns the number of rounds the queues took."""
    if steps is None:
        steps = [build_steps(kind, n_stages, s, n_micro) for s in range(n_stages)]
    queues = [[list(g) for g in steps[s] if g[0][0] in COMM_OPS] for s in range(n_stages)]
    head = [0] * n_stages
    done = [set() for _ in range(n_stages)]              # completed operations of the running group (non-strict)
    rounds = 0
    while any(head[s] < len(queues[s]) for s in range(n_stages)):
        running = [set(queues[s][head[s]]) if head[s] < len(queues[s]) else set() for s in range(n_stages)]
        progressed = False
        if strict:
            retire = []
            for s in range(n_stages):
                if running[s] and all(0 <= _peer_op(op, s)[0] < n_stages and _peer_op(op, s)[1] in running[_peer_op(op, s)[0]]
                                      for op in running[s]):
                    retire.append(s)
            for s in retire:
                head[s] += 1
                progressed = True
        else:
            for s in range(n_stages):
                for op in running[s] - done[s]:
                    peer, want = _peer_op(op, s)
                    if 0 <= peer < n_stages and want in running[peer]:
                        done[s].add(op)
                        done[peer].add(want)
                        progressed = True
            for s in range(n_stages):
                if running[s] and done[s] >= running[s]:
                    head[s] += 1
                    done[s] = set()
                    progressed = True
        if not progressed:
            stuck = {s: queues[s][head[s]] for s in range(n_stages) if head[s] < len(queues[s])}
            raise AssertionError(f"rendezvous deadlock in {kind} (S={n_stages}, M={n_micro}, strict={strict}): heads {stuck}")
        rounds += 1
    return rounds


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
