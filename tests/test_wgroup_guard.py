"""ops/wgroup.py: a backward pass that dies (exception inside backward) leaves held-back weight-gradient problems and a
queued-callback flag behind; the next pass must not launch them into its own gradient buffers nor rely on the dead callback."""
import pytest
import torch

from distributedpytorch_amd.ops import wgroup


class _Hold(torch.autograd.Function):
    @staticmethod
    def forward(ctx, a, batcher, boom):
        ctx.batcher, ctx.boom = batcher, boom
        return a * 2

    @staticmethod
    def backward(ctx, g):
        x = torch.zeros(4, 8)
        ctx.batcher.add(x, x, torch.zeros(8, 8), None, [], 1)
        if ctx.boom:
            raise RuntimeError("backward died")
        return g * 2, None, None


def test_problems_of_a_dead_backward_pass_are_dropped_not_launched(monkeypatch):
    b = wgroup.WgradBatcher(0)
    launched = []
    monkeypatch.setattr(b, "flush", lambda: (launched.append(len(b.items)), b.items.clear(), setattr(b, "tiles", 0)))
    a = torch.ones(3, requires_grad=True)
    with pytest.raises(RuntimeError, match="backward died"):
        _Hold.apply(a, b, True).sum().backward()
    assert len(b.items) == 1 and b.callback_queued and launched == []        # stranded: the callback never ran
    _Hold.apply(a, b, False).sum().backward()
    assert b.dropped == 1                                                     # the stale problem was discarded ...
    assert launched == [1] and not b.callback_queued                          # ... and only the new one went out, by the new callback
