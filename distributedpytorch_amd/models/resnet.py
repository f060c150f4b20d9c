"""ResNet-50 / -101 / -152 (v1.5 bottleneck, torchvision naming) on the NHWC bf16 kernels.

Not in the reference (it only has the MNIST CNN); required by BASELINE.json configs 3
and 4.  Module / parameter names follow torchvision (``conv1 bn1 layer1.0.conv1 ...
layer1.0.downsample.0 fc``) so exported checkpoints interoperate.  Every conv is the
implicit-GEMM MFMA kernel, its BatchNorm statistics come out of the conv epilogue,
and BN + residual add + ReLU are one streaming kernel.
"""
from __future__ import annotations

from typing import List, Sequence

import torch
import torch.nn as nn

from .. import ops
from ..nn import Conv2d, BatchNorm2d, Linear, conv_bn
from ..ops import block as _block


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes: int, planes: int, stride: int = 1, downsample: bool = False):
        super().__init__()
        out = planes * self.expansion
        self.conv1 = Conv2d(inplanes, planes, 1)
        self.bn1 = BatchNorm2d(planes)
        self.conv2 = Conv2d(planes, planes, 3, stride=stride, padding=1)   # v1.5: stride on the 3x3
        self.bn2 = BatchNorm2d(planes)
        self.conv3 = Conv2d(planes, out, 1)
        self.bn3 = BatchNorm2d(out, zero_init=True)
        if downsample:
            self.downsample = nn.Sequential(Conv2d(inplanes, out, 1, stride=stride), BatchNorm2d(out))
        else:
            self.downsample = None

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if _block.can_fuse(self, x):
            return _block.bottleneck(self, x)      # one autograd node, hand-scheduled backward
        identity = x
        if self.downsample is not None:
            identity = conv_bn(self.downsample[0], self.downsample[1], x, relu=False)
        y = conv_bn(self.conv1, self.bn1, x, relu=True)
        y = conv_bn(self.conv2, self.bn2, y, relu=True)
        return conv_bn(self.conv3, self.bn3, y, relu=True, residual=identity)   # relu(bn3(conv3) + identity)


class Stem(nn.Module):
    """NCHW input -> NHWC(8ch) -> conv7x7/2 -> BN -> ReLU -> maxpool3x3/2."""

    def __init__(self):
        super().__init__()
        self.conv1 = Conv2d(3, 64, 7, stride=2, padding=3)
        self.bn1 = BatchNorm2d(64)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() == 4 and x.shape[1] == 3:          # NCHW image batch -> kernel layout
            if _block.can_fuse_stem_packed(self, x):
                return _block.stem_packed(self, x)     # pixel-pair layout: the 7x7 conv as 7 x 4 taps (ops/block.py)
            x = ops.to_nhwc(x, self.conv1.in_padded)
        if _block.can_fuse_stem(self, x):
            return _block.stem(self, x)            # one autograd node; the 112x112 activation is never written
        y = conv_bn(self.conv1, self.bn1, x, relu=True)
        return ops.max_pool_3x3_s2_nhwc(y)


class Head(nn.Module):
    def __init__(self, in_features: int, num_classes: int):
        super().__init__()
        self.fc = Linear(in_features, num_classes)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.fc(ops.global_avg_pool_nhwc(x))


class ResNet(nn.Module):
    def __init__(self, layers: Sequence[int] = (3, 4, 6, 3), num_classes: int = 1000):
        super().__init__()
        stem = Stem()
        self.conv1, self.bn1 = stem.conv1, stem.bn1     # torchvision key names
        self._stem = [stem]                              # not registered twice
        inplanes = 64
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            stride = 1 if i == 0 else 2
            blocks: List[nn.Module] = [Bottleneck(inplanes, planes, stride, downsample=True)]
            inplanes = planes * Bottleneck.expansion
            blocks += [Bottleneck(inplanes, planes) for _ in range(1, n)]
            setattr(self, f"layer{i + 1}", nn.Sequential(*blocks))
        head = Head(inplanes, num_classes)
        self.fc = head.fc
        self._head = [head]
        self.num_classes = num_classes

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = self._stem[0](x)
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        return self._head[0](x)

    def loss(self, x: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
        return ops.softmax_cross_entropy(self.forward(x), target, self.num_classes)

    def pipeline_layers(self) -> List[nn.Module]:
        """Flat list of sequential units for the pipeline partitioner."""
        units: List[nn.Module] = [self._stem[0]]
        for i in range(1, 5):
            units += list(getattr(self, f"layer{i}"))
        units.append(self._head[0])
        return units


def resnet50(num_classes: int = 1000) -> ResNet:
    return ResNet((3, 4, 6, 3), num_classes)


def resnet101(num_classes: int = 1000) -> ResNet:
    return ResNet((3, 4, 23, 3), num_classes)


def resnet152(num_classes: int = 1000) -> ResNet:
    return ResNet((3, 8, 36, 3), num_classes)


def tiny_resnet(num_classes: int = 16, width: int = 16) -> nn.Module:
    """A 2-block bottleneck net for tests / smoke (same code paths, tiny shapes)."""
    class Tiny(nn.Module):
        def __init__(self):
            super().__init__()
            self.stem = Stem()
            self.block1 = Bottleneck(64, width, 1, downsample=True)
            self.block2 = Bottleneck(width * 4, width, 2, downsample=True)
            self.head = Head(width * 4, num_classes)
            self.num_classes = num_classes

        def forward(self, x):
            return self.head(self.block2(self.block1(self.stem(x))))

        def loss(self, x, target):
            return ops.softmax_cross_entropy(self.forward(x), target, self.num_classes)

        def pipeline_layers(self):
            return [self.stem, self.block1, self.block2, self.head]

    return Tiny()


def flops_per_image(model: ResNet, hw: int = 224) -> float:
    """Forward multiply-accumulates * 2 of the conv / fc layers (for MFU reporting)."""
    total = 0.0

    def conv_flops(c: Conv2d, h: int) -> float:
        ho = (h + 2 * c.padding - c.kernel_size) // c.stride + 1
        return 2.0 * ho * ho * c.out_channels * c.in_channels * c.kernel_size ** 2, ho

    h = hw
    f, h = conv_flops(model.conv1, h)
    total += f
    h = (h + 2 - 3) // 2 + 1
    for i in range(1, 5):
        for blk in getattr(model, f"layer{i}"):
            h_in = h
            f, h1 = conv_flops(blk.conv1, h_in); total += f
            f, h2 = conv_flops(blk.conv2, h1); total += f
            f, h3 = conv_flops(blk.conv3, h2); total += f
            if blk.downsample is not None:
                f, _ = conv_flops(blk.downsample[0], h_in); total += f
            h = h3
    total += 2.0 * model.fc.in_features * model.fc.out_features
    return total
