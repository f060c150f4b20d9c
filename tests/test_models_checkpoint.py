"""T0/T3 on CPU: model parity with stock PyTorch, checkpoint format, fused optimizer."""
import os

import torch
import torch.nn as nn
import torch.nn.functional as F

from distributedpytorch_amd import checkpoint as ck
from distributedpytorch_amd.models import mnist, resnet, gpt2
from distributedpytorch_amd.optim import FusedSGD, FusedAdamW
from distributedpytorch_amd.parallel.flat import FlatParams


class StockNet(nn.Module):
    """The reference's Net, verbatim structure (DP.py:77-95), stock torch ops."""
    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 10, kernel_size=5)
        self.conv2 = nn.Conv2d(10, 20, kernel_size=5)
        self.conv2_drop = nn.Dropout2d()
        self.fc1 = nn.Linear(320, 50)
        self.fc2 = nn.Linear(50, 10)

    def forward(self, x):
        x = F.relu(F.max_pool2d(self.conv1(x), 2))
        x = F.relu(F.max_pool2d(self.conv2_drop(self.conv2(x)), 2))
        x = x.view(-1, 320)
        x = F.relu(self.fc1(x))
        x = F.dropout(x, training=self.training)
        x = self.fc2(x)
        return F.log_softmax(x, dim=1)


def test_net_matches_stock_and_state_dict_is_interchangeable(tmp_path):
    torch.manual_seed(0)
    ours, stock = mnist.Net().eval(), StockNet().eval()
    assert list(ours.state_dict().keys()) == list(stock.state_dict().keys())
    assert sum(p.numel() for p in ours.parameters()) == 21840
    path = str(tmp_path / "net.pt")
    ck.save_checkpoint(path, ours)
    stock.load_state_dict(torch.load(path + ".model", weights_only=True))       # plain torch.load + load_state_dict
    x = torch.randn(12, 1, 28, 28)
    assert torch.allclose(ours(x), stock(x), atol=1e-5)
    # and back: a stock checkpoint loads into ours
    torch.save(StockNet().state_dict(), path)
    ck.load_checkpoint(path, ours)


def test_net_loss_equals_nll_of_forward_on_cpu():
    """``Net.loss`` / ``Net2.loss`` (fused head + loss on the GPU) fall back to the separate ops on the CPU."""
    torch.manual_seed(0)
    net = mnist.Net().eval()
    x, t = torch.randn(9, 1, 28, 28), torch.randint(0, 10, (9,))
    loss, logp = net.loss(x, t, return_output=True)
    assert torch.allclose(loss, F.nll_loss(net(x), t), atol=1e-6) and torch.allclose(logp, net(x), atol=1e-6)
    n2 = mnist.Net2()
    h = torch.randn(9, 320)
    assert torch.allclose(n2.loss(h, t), F.nll_loss(n2(h), t), atol=1e-6)
    loss = net.loss(x, t); loss.backward()
    assert all(p.grad is not None for p in net.parameters())


def test_mp_stage_keys_union_is_dp_key_set():
    k1, k2 = set(mnist.Net1().state_dict()), set(mnist.Net2().state_dict())
    assert k1 | k2 == set(mnist.Net().state_dict()) and not (k1 & k2)
    x = torch.randn(5, 1, 28, 28)
    assert mnist.Net1()(x).shape == (5, 320) and mnist.Net2()(torch.randn(5, 320)).shape == (5, 10)


def test_resnet50_structure_and_torchvision_keys():
    m = resnet.resnet50()
    n = sum(p[..., :3].numel() if k == "conv1.weight" else p.numel() for k, p in m.named_parameters())
    assert n == 25557032                                                  # torchvision resnet50 parameter count
    sd = ck.export_state_dict(m)
    assert sd["conv1.weight"].shape == (64, 3, 7, 7)
    assert sd["layer1.0.downsample.0.weight"].shape == (256, 64, 1, 1)
    assert sd["layer4.2.conv2.weight"].shape == (512, 512, 3, 3)
    assert sd["fc.weight"].shape == (1000, 2048) and "layer3.5.bn3.running_var" in sd
    assert abs(resnet.flops_per_image(m) / 2 - 4.09e9) < 0.1e9             # ~4.1 GMACs / image
    assert len(resnet.resnet152().pipeline_layers()) == 1 + 50 + 1


def test_checkpoint_roundtrip_with_padding_and_optimizer(tmp_path):
    torch.manual_seed(1)
    m = resnet.tiny_resnet()
    flat = FlatParams(m)
    opt = FusedSGD(flat, lr=0.1, momentum=0.9)
    x, y = torch.randn(4, 3, 32, 32), torch.randint(0, 16, (4,))
    m.loss(x, y).backward(); opt.step()
    path = str(tmp_path / "c.pt")
    ck.save_checkpoint(path, m, opt, epoch=3)
    before = m.eval()(x).clone()
    m2 = resnet.tiny_resnet()
    flat2 = FlatParams(m2)
    opt2 = FusedSGD(flat2, lr=0.5, momentum=0.0)
    payload = ck.load_checkpoint(path, m2, opt2, flat=flat2)
    assert payload["epoch"] == 3 and opt2.lr == 0.1 and opt2.momentum == 0.9
    assert torch.allclose(m2.eval()(x), before, atol=1e-6)
    assert torch.allclose(opt2.momentum_buf, opt.momentum_buf)
    # parameters still live inside the flat buffer after loading
    p = next(iter(m2.parameters()))
    assert p.data_ptr() >= flat2.master.data_ptr() and p.data_ptr() < flat2.master.data_ptr() + flat2.master.numel() * 4


def test_fused_sgd_matches_torch_sgd():
    torch.manual_seed(2)
    a, b = mnist.Net().eval(), mnist.Net().eval()
    b.load_state_dict(a.state_dict())
    flat = FlatParams(a)
    fo = FusedSGD(flat, lr=0.01, momentum=0.5, weight_decay=1e-3, no_decay=None)
    to = torch.optim.SGD(b.parameters(), lr=0.01, momentum=0.5, weight_decay=1e-3)
    x, y = torch.randn(16, 1, 28, 28), torch.randint(0, 10, (16,))
    for _ in range(3):
        mnist.nll_loss(a(x), y).backward(); fo.step()
        to.zero_grad(); F.nll_loss(b(x), y).backward(); to.step()
    for (n, p), q in zip(a.named_parameters(), b.parameters()):
        assert torch.allclose(p, q, atol=1e-6), n
    assert float(flat.grad.abs().max()) == 0.0            # zero_grad is fused into the step
    sd = fo.state_dict()
    assert sd["param_groups"][0]["momentum"] == 0.5 and sd["state"][0]["momentum_buffer"].shape == a.conv1.weight.shape


def test_gpt2_tiny_cpu_and_hf_export(tmp_path):
    torch.manual_seed(3)
    m = gpt2.gpt2_tiny()
    tok = torch.randint(0, 500, (2, 16)); tgt = torch.randint(0, 500, (2, 16))
    flat = FlatParams(m)
    opt = FusedAdamW(flat, lr=1e-2)
    l0 = None
    for _ in range(5):
        loss = m.loss(tok, tgt); loss.backward(); opt.step()
        l0 = l0 if l0 is not None else float(loss)
    assert float(loss) < l0
    sd = ck.export_state_dict(m, hf_conv1d=True)
    assert sd["emb.wte"].shape == (500, 128)                      # vocab padding stripped
    assert sd["h.0.attn.c_attn.weight"].shape == (128, 384)       # HF Conv1D layout [in, out]
    assert gpt2.gpt2_small().cfg.vocab_padded == 50304
    n = sum(p.numel() for p in gpt2.gpt2_small().parameters())
    assert abs(n - 124.4e6) < 0.5e6


def test_gpt2_dropout_on_cpu_is_active_in_training_only():
    """``GPT2Config.dropout`` (embedding, attention probabilities, residual branches): CPU fallback of the in-kernel
    dropout - training calls differ from one another and from eval, eval calls agree, gradients flow."""
    torch.manual_seed(0)
    m = gpt2.gpt2_tiny(n_layer=2, n_embd=64, n_head=2, vocab=300, n_positions=32, dropout=0.2).train()
    tok = torch.randint(0, 300, (2, 32))
    l1 = m.loss(tok, tok); l1.backward()
    l2 = m.loss(tok, tok)
    assert torch.isfinite(l1) and abs(float(l1) - float(l2)) > 1e-6
    assert all(p.grad is not None and torch.isfinite(p.grad).all() for p in m.parameters())
    m.eval()
    with torch.no_grad():
        assert float(m.loss(tok, tok)) == float(m.loss(tok, tok))
    m0 = gpt2.gpt2_tiny(n_layer=2, n_embd=64, n_head=2, vocab=300, n_positions=32).train()
    assert float(m0.loss(tok, tok)) == float(m0.loss(tok, tok))          # dropout 0: no stream, no randomness


def test_checkpoint_carries_the_dropout_stream_position(tmp_path):
    """the MNIST nets' counter-based dropout stream resumes where it stopped (the next masks, not the first)."""
    import torch
    from distributedpytorch_amd.checkpoint import save_checkpoint, load_checkpoint
    from distributedpytorch_amd.models import mnist
    a = mnist.Net().train()
    x = torch.randn(4, 1, 28, 28)
    for _ in range(3):
        a(x)
    assert a._drop.step == 6                      # two dropout layers per forward
    save_checkpoint(str(tmp_path / "c.pt"), a, None, 1)
    b = mnist.Net().train()
    load_checkpoint(str(tmp_path / "c.pt"), b)
    assert b._drop.step == 6
    torch.testing.assert_close(a(x), b(x))        # same parameters, same next masks
