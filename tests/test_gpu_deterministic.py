"""DPA_DETERMINISTIC / set_deterministic (csrc/runtime.hip): with it on, two runs of the same step from the same seed
give bit-identical gradients on both model families (the default kernels add with float atomics and differ in the
last bits from run to run - VERDICT r1 weak #7); and the deterministic routes compute the same gradients as the
default ones up to summation order."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _resnet_grads():
    from distributedpytorch_amd.models import resnet
    from distributedpytorch_amd.parallel.dp import DataParallel
    torch.manual_seed(3)
    m = resnet.tiny_resnet().to(DEV).train()
    dp = DataParallel(m)
    g = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn(32, 3, 64, 64, generator=g).to(DEV); y = torch.randint(0, 16, (32,), generator=g).to(DEV)
    loss = m.loss(x, y)
    loss.backward(); dp.finish()
    torch.cuda.synchronize()
    return float(loss.detach()), dp.flat.grad.clone(), {n: b.clone() for n, b in m.named_buffers() if "running" in n}


def _gpt2_grads():
    from distributedpytorch_amd.models import gpt2
    from distributedpytorch_amd.parallel.dp import DataParallel
    torch.manual_seed(4)
    m = gpt2.gpt2_tiny().to(DEV).train()
    dp = DataParallel(m)
    g = torch.Generator(device="cpu").manual_seed(6)
    tok = torch.randint(0, 500, (4, 128), generator=g).to(DEV)
    tgt = torch.randint(0, 500, (4, 128), generator=g).to(DEV)
    loss = m.loss(tok, tgt)
    loss.backward(); dp.finish()
    torch.cuda.synchronize()
    return float(loss.detach()), dp.flat.grad.clone(), {}


@pytest.mark.parametrize("family", ["resnet", "gpt2"])
def test_deterministic_mode_reproduces_gradients_bit_for_bit(family):
    import distributedpytorch_amd as dpa
    run = _resnet_grads if family == "resnet" else _gpt2_grads
    assert not dpa.is_deterministic()
    l0, g0, _ = run()                       # default kernels (atomics)
    dpa.set_deterministic(True)
    try:
        assert dpa.is_deterministic()
        l1, g1, b1 = run()
        l2, g2, b2 = run()
    finally:
        dpa.set_deterministic(False)
    assert l1 == l2, "loss must reproduce exactly"
    assert torch.equal(g1, g2), f"{int((g1 != g2).sum())} of {g1.numel()} gradient elements differ between two deterministic runs"
    for k in b1:
        assert torch.equal(b1[k], b2[k]), k
    # same mathematics as the default routes
    assert abs(l1 - l0) <= 1e-3 * abs(l0) + 1e-5
    assert ((g1 - g0).norm() / g0.norm()).item() < 2e-2
    assert torch.nn.functional.cosine_similarity(g1, g0, dim=0).item() > 0.9995
