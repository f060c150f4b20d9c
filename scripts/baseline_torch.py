#!/usr/bin/env python
"""Comparison line: STOCK PyTorch-ROCm (torch.nn on MIOpen / hipBLASLt, channels_last,
bf16 autocast, torch.optim.SGD) training ResNet-50 on the same synthetic shapes on the
same MI355X.  The reference publishes no number, so this measured
number is what ``bench.py`` reports ``vs_baseline`` against.  torchvision is not
installed, so the torchvision ResNet-50 v1.5 architecture is re-stated here with
stock ``torch.nn`` modules.

    python scripts/baseline_torch.py --steps 20 --warmup 5 --batch 256
"""
import argparse
import json
import time

import torch
import torch.nn as nn
import torch.nn.functional as F


class Bottleneck(nn.Module):
    def __init__(self, inp, planes, stride=1, down=False):
        super().__init__()
        out = planes * 4
        self.conv1 = nn.Conv2d(inp, planes, 1, bias=False); self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = nn.Conv2d(planes, planes, 3, stride, 1, bias=False); self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = nn.Conv2d(planes, out, 1, bias=False); self.bn3 = nn.BatchNorm2d(out)
        self.downsample = nn.Sequential(nn.Conv2d(inp, out, 1, stride, bias=False), nn.BatchNorm2d(out)) if down else None

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        y = F.relu(self.bn1(self.conv1(x)))
        y = F.relu(self.bn2(self.conv2(y)))
        return F.relu(self.bn3(self.conv3(y)) + idt)


class ResNet50(nn.Module):
    def __init__(self, layers=(3, 4, 6, 3), classes=1000):
        super().__init__()
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False); self.bn1 = nn.BatchNorm2d(64)
        inp = 64
        stages = []
        for i, (pl, n) in enumerate(zip((64, 128, 256, 512), layers)):
            blocks = [Bottleneck(inp, pl, 1 if i == 0 else 2, True)]
            inp = pl * 4
            blocks += [Bottleneck(inp, pl) for _ in range(n - 1)]
            stages.append(nn.Sequential(*blocks))
        self.layers = nn.Sequential(*stages)
        self.fc = nn.Linear(inp, classes)

    def forward(self, x):
        x = F.max_pool2d(F.relu(self.bn1(self.conv1(x))), 3, 2, 1)
        x = self.layers(x)
        return self.fc(torch.flatten(F.adaptive_avg_pool2d(x, 1), 1))


class GPT2Block(nn.Module):
    def __init__(self, C, H):
        super().__init__()
        self.ln_1 = nn.LayerNorm(C); self.ln_2 = nn.LayerNorm(C)
        self.c_attn = nn.Linear(C, 3 * C); self.c_proj = nn.Linear(C, C)
        self.c_fc = nn.Linear(C, 4 * C); self.c_proj2 = nn.Linear(4 * C, C)
        self.H = H

    def forward(self, x):
        B, T, C = x.shape
        q, k, v = self.c_attn(self.ln_1(x)).split(C, dim=2)
        q, k, v = (t.view(B, T, self.H, C // self.H).transpose(1, 2) for t in (q, k, v))
        a = F.scaled_dot_product_attention(q, k, v, is_causal=True).transpose(1, 2).reshape(B, T, C)
        x = x + self.c_proj(a)
        return x + self.c_proj2(F.gelu(self.c_fc(self.ln_2(x)), approximate="tanh"))


class GPT2Small(nn.Module):
    """GPT-2 small (124M): 12 layers, 768 wide, 12 heads, vocab 50257 (padded 50304), tied LM head."""

    def __init__(self, V=50304, T=1024, C=768, L=12, H=12):
        super().__init__()
        self.wte = nn.Embedding(V, C); self.wpe = nn.Embedding(T, C)
        self.h = nn.ModuleList([GPT2Block(C, H) for _ in range(L)])
        self.ln_f = nn.LayerNorm(C)

    def forward(self, tok):
        x = self.wte(tok) + self.wpe(torch.arange(tok.shape[1], device=tok.device))
        for b in self.h:
            x = b(x)
        return F.linear(self.ln_f(x), self.wte.weight)


class MnistNet(nn.Module):
    """The reference's DP model (multinodeDP.py:77-95)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 10, 5); self.conv2 = nn.Conv2d(10, 20, 5); self.conv2_drop = nn.Dropout2d()
        self.fc1 = nn.Linear(320, 50); self.fc2 = nn.Linear(50, 10)

    def forward(self, x):
        x = F.relu(F.max_pool2d(self.conv1(x), 2))
        x = F.relu(F.max_pool2d(self.conv2_drop(self.conv2(x)), 2)).view(-1, 320)
        x = F.dropout(F.relu(self.fc1(x)), training=self.training)
        return F.log_softmax(self.fc2(x), dim=1)


def _heartbeat():
    """MIOpen's first-call kernel search can run for minutes without output."""
    import threading
    t0 = time.time()

    def beat():
        while True:
            time.sleep(30)
            print(f"[baseline] alive, {time.time() - t0:.0f}s (MIOpen search / warmup)", flush=True)
    threading.Thread(target=beat, daemon=True).start()


def main():
    _heartbeat()
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--out", default=None)
    ap.add_argument("--model", default="resnet50", choices=["resnet50", "resnet152", "gpt2", "mnist"])
    ap.add_argument("--seq-len", type=int, default=1024)
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(1234)
    torch.backends.cudnn.benchmark = True
    items = a.batch
    if a.model in ("resnet50", "resnet152"):
        layers = (3, 4, 6, 3) if a.model == "resnet50" else (3, 8, 36, 3)
        model = ResNet50(layers).to(dev).to(memory_format=torch.channels_last).train()
        opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
        x = torch.randn(a.batch, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
        y = torch.randint(0, 1000, (a.batch,), device=dev)
        lossf = lambda: F.cross_entropy(model(x), y)  # noqa: E731
        amp = True
    elif a.model == "gpt2":
        model = GPT2Small().to(dev).train()
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9)
        x = torch.randint(0, 50257, (a.batch, a.seq_len), device=dev)
        y = torch.randint(0, 50257, (a.batch, a.seq_len), device=dev)
        lossf = lambda: F.cross_entropy(model(x).view(-1, 50304).float(), y.view(-1))  # noqa: E731
        items = a.batch * a.seq_len
        amp = True
    else:
        model = MnistNet().to(dev).train()
        opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.5)
        x = torch.randn(a.batch, 1, 28, 28, device=dev)
        y = torch.randint(0, 10, (a.batch,), device=dev)
        lossf = lambda: F.nll_loss(model(x), y)  # noqa: E731
        amp = False

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16, enabled=amp):
            loss = lossf()
        loss.backward()
        opt.step()
        return loss

    for i in range(a.warmup):
        step()
        torch.cuda.synchronize()
        print(f"[baseline] warmup step {i} done", flush=True)   # MIOpen's first-call search is slow; keep the log alive
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    res = {"what": f"stock PyTorch-ROCm {a.model} ({'bf16 autocast' if amp else 'fp32'}), torch.optim.SGD momentum", "model": a.model,
           "batch": a.batch, "steps": a.steps, "items_per_sec": round(items * a.steps / dt, 2),
           "images_per_sec": round(a.batch * a.steps / dt, 2), "ms_per_step": round(1000 * dt / a.steps, 3),
           "torch": torch.__version__, "device": torch.cuda.get_device_name(0)}
    print(json.dumps(res), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f)


if __name__ == "__main__":
    main()
