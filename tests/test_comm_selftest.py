"""The native layer's first-contact self-test (comm.Communicator._selftest_native) at world size > 1, run on the CPU against a
stand-in for csrc/rccl_comm.cpp's RcclComm that produces, for ONE rank, what a correct W-rank job would deliver.  The real
thing needs several GPUs (tests/test_multi_gpu.py); this holds the Python side of it - phase order, expected values,
agreement after every phase, the all-ranks fallback - to its contract on every run."""
import pytest
import torch

from distributedpytorch_amd import comm as dcomm


class FakeRccl:
    """One rank's view of a W-rank communicator whose rank q contributes what the self-test makes rank q contribute."""

    def __init__(self, rank, world, broken=None, log=None):
        self.r, self.W, self.broken, self.log = rank, world, broken, log if log is not None else []
        self.aborted = False

    def _done(self, name):
        self.log.append(name)
        if self.broken == name:
            raise RuntimeError(f"injected failure in {name}")
        return len(self.log)

    def query(self, ticket):
        return True

    def wait(self, ticket):
        return None

    def synchronize(self):
        return None

    def abort(self):
        self.aborted = True

    def all_reduce(self, t, op):
        t.fill_(self.W * (self.W + 1) / 2.0)            # rank q contributed q + 1
        return self._done("all_reduce")

    def broadcast(self, t, root):
        t.fill_(float(root + 1))
        return self._done("broadcast")

    def reduce(self, t, root):
        if self.r == root:
            t.fill_(self.W * (self.W + 1) / 2.0)
        return self._done("reduce")

    def reduce_scatter(self, piece, src):
        W, r = self.W, self.r
        piece.copy_((torch.arange(8, dtype=torch.float32) + 8.0 * r) * W + 1000.0 * (W * (W - 1) / 2.0))
        return self._done("reduce_scatter")

    def all_gather(self, full, piece):
        W = self.W
        for q in range(W):
            full[8 * q: 8 * q + 8] = (torch.arange(8, dtype=torch.float32) + 8.0 * q) * W + 1000.0 * (W * (W - 1) / 2.0)
        return self._done("all_gather")

    def batch_p2p(self, ops):
        for is_send, t, peer in ops:
            if not is_send:
                t.fill_(float(peer))
        return self._done("batch_p2p")

    def split(self, color, key, sub_rank, sub_world):
        self._done("split")
        if color < 0:
            return None
        return FakeSub(sub_world, self.log, self.broken)


class FakeSub(FakeRccl):
    def __init__(self, world, log, broken=None):
        super().__init__(0, world, broken, log)

    def all_reduce(self, t, op):
        t.fill_(float(self.W))
        return self._done("sub_all_reduce")


def _comm(rank, world, native, votes=None):
    c = dcomm.Communicator(group=None, device=torch.device("cpu"), static=(rank, world), store=None)
    c.native = native
    seen = []
    def agree(ok):                      # the other ranks' votes: all fine unless the test says otherwise
        seen.append(ok)
        return ok and (votes is None or votes[len(seen) - 1])
    c._agree = agree
    return c, seen


@pytest.mark.parametrize("world,rank", [(2, 0), (2, 1), (4, 1), (4, 3), (8, 0), (8, 7)])
def test_native_selftest_passes_on_a_correct_communicator(world, rank):
    native = FakeRccl(rank, world)
    c, seen = _comm(rank, world, native)
    c._selftest_native(strict=True)
    assert seen == [True, True, True]                                   # one agreement per phase
    assert native.log[:2] == ["all_reduce", "all_reduce"] and "batch_p2p" in native.log
    assert native.log.count("split") == (2 if world > 2 else 1)          # halves, then the split that leaves the last rank out
    assert not native.aborted


@pytest.mark.parametrize("broken,phases_run", [("all_reduce", 1), ("all_gather", 2), ("batch_p2p", 2), ("sub_all_reduce", 3)])
def test_native_selftest_failure_stops_before_the_next_collective_phase(broken, phases_run):
    """A rank that fails a check votes no and NO rank enters the next phase (ncclCommSplit would block the others for good)."""
    native = FakeRccl(1, 4, broken=broken)
    c, seen = _comm(1, 4, native)
    with pytest.raises(RuntimeError, match="self-test"):
        c._selftest_native(strict=True)
    assert len(seen) == phases_run and seen[-1] is False
    if phases_run < 3:
        assert "split" not in native.log


def test_native_selftest_follows_another_ranks_failure():
    """This rank is fine, another one voted no after the second phase: nobody splits, the layer is dropped here too."""
    native = FakeRccl(0, 4)
    c, seen = _comm(0, 4, native, votes=[True, False, True])
    with pytest.raises(RuntimeError, match="another rank"):
        c._selftest_native(strict=True)
    assert seen == [True, True] and "split" not in native.log
