"""utils/pacer.StepPacer: inert on the CPU; on a device it never lets more than ``depth`` recorded steps be outstanding."""
import pytest
import torch

from distributedpytorch_amd.utils.pacer import StepPacer


def test_pacer_is_inert_without_a_gpu():
    p = StepPacer(torch.device("cpu"), depth=2)
    for _ in range(10):
        p.step()
    p.drain()
    assert not p.enabled and p.waits == 0
    assert not StepPacer(None).enabled


@pytest.mark.gpu
def test_pacer_bounds_the_outstanding_steps_and_reuses_its_events():
    dev = torch.device("cuda", 0)
    p = StepPacer(dev, depth=2)
    x = torch.zeros(1 << 20, device=dev)
    for i in range(20):
        x.add_(1.0)
        p.step()
        assert len(p._events) <= 2
    assert p.waits == 18 and len(p._pool) + len(p._events) <= 3       # events are recycled, not created per step
    p.drain()
    assert len(p._events) == 0 and float(x[0]) == 20.0
    q = StepPacer(dev, depth=2, stride=8)                               # sub-millisecond steps: every 8th step is paced
    for i in range(64):
        q.step()
    assert q.waits == 64 // 8 - 2
    assert not StepPacer(dev, depth=0).enabled                          # --run-ahead 0: unbounded, as before
