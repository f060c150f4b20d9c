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
    a = ap.parse_args()
    dev = torch.device("cuda")
    torch.manual_seed(1234)
    torch.backends.cudnn.benchmark = True
    model = ResNet50().to(dev).to(memory_format=torch.channels_last).train()
    opt = torch.optim.SGD(model.parameters(), lr=0.1, momentum=0.9, weight_decay=1e-4)
    x = torch.randn(a.batch, 3, 224, 224, device=dev).contiguous(memory_format=torch.channels_last)
    y = torch.randint(0, 1000, (a.batch,), device=dev)

    def step():
        opt.zero_grad(set_to_none=True)
        with torch.autocast("cuda", dtype=torch.bfloat16):
            loss = F.cross_entropy(model(x), y)
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
    res = {"what": "stock PyTorch-ROCm ResNet-50 bf16 autocast channels_last, SGD momentum", "batch": a.batch,
           "steps": a.steps, "images_per_sec": round(a.batch * a.steps / dt, 2), "ms_per_step": round(1000 * dt / a.steps, 3),
           "torch": torch.__version__, "device": torch.cuda.get_device_name(0)}
    print(json.dumps(res), flush=True)
    if a.out:
        with open(a.out, "w") as f:
            json.dump(res, f)


if __name__ == "__main__":
    main()
