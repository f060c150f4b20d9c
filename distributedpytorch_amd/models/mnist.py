"""The reference's MNIST networks on the fused fp32 kernels.

``Net`` (multinode-DP/multinodeDP.py:77-95), ``Net1`` / ``Net2`` (the two model-parallel
stages, multinode-MP/multinodeMP.py:76-100).  Attribute names - and therefore
``state_dict`` keys, shapes (OIHW) and dtypes - are exactly the reference's
(``conv1 conv2 conv2_drop fc1 fc2``), so a checkpoint moves freely between the stock
PyTorch model and this one, and ``Net1`` + ``Net2`` together load into ``Net``.
The ``nn.Conv2d`` / ``nn.Linear`` members are used as parameter containers only;
the math runs in ``ops.conv5_pool_relu`` / ``ops.linear_f32`` (one fused kernel per
layer) instead of cuDNN/cuBLAS + separate pool / relu kernels.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import ops


class _DropState:
    """Counter-based dropout stream: masks are a pure function of (seed, step, layer).

    Eager mode counts steps on the host.  ``use_device_counter`` moves the step counter into
    device memory (bumped by a one-thread kernel, ``advance()``), so a training step captured in
    a hipGraph still draws fresh masks at every replay."""

    def __init__(self, seed: int = 1234):
        self.seed, self.step = seed, 0
        self.dev_step = None

    def use_device_counter(self, device) -> None:
        self.dev_step = torch.zeros(1, dtype=torch.int64, device=device)

    def next(self, layer: int) -> int:
        if self.dev_step is not None:
            return layer * 1_000_003          # the device counter supplies the step term
        self.step += 1
        return self.step * 1_000_003 + layer

    def advance(self) -> None:
        """End of a step whose loss did NOT go through ``Net.loss`` (that kernel bumps the device counter itself)."""
        if self.dev_step is not None:
            from ..ops import _ext
            _ext.load().counter_add(self.dev_step, 1)


class Net(nn.Module):
    """2 conv + 2 fc MNIST CNN with dropout (reference DP.py:77-95). 21,840 parameters."""

    def __init__(self, seed: int = 1234):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 10, kernel_size=5)
        self.conv2 = nn.Conv2d(10, 20, kernel_size=5)
        self.conv2_drop = nn.Dropout2d()
        self.fc1 = nn.Linear(320, 50)
        self.fc2 = nn.Linear(50, 10)
        self._drop = _DropState(seed)

    def _drop_arg(self, p: float, layer: int):
        if not self.training or p <= 0:
            return None
        return (p, self._drop.seed, self._drop.next(layer), self._drop.dev_step)

    def features(self, x: torch.Tensor) -> torch.Tensor:
        x = ops.conv5_pool_relu(x, self.conv1.weight, self.conv1.bias)
        x = ops.conv5_pool_relu(x, self.conv2.weight, self.conv2.bias, self._drop_arg(self.conv2_drop.p, 0))
        return x.reshape(-1, 320)

    def hidden(self, x: torch.Tensor) -> torch.Tensor:
        return ops.linear_f32(self.features(x), self.fc1.weight, self.fc1.bias, relu=True, drop=self._drop_arg(0.5, 7))

    def logits(self, x: torch.Tensor) -> torch.Tensor:
        return ops.linear_f32(self.hidden(x), self.fc2.weight, self.fc2.bias)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.linear_log_softmax(self.hidden(x), self.fc2.weight, self.fc2.bias)

    def loss(self, x: torch.Tensor, target: torch.Tensor, return_output: bool = False):
        """``F.nll_loss(self(x), target)`` (reference DP.py:151-152) with the head, the log-softmax and the loss in one
        kernel each way.  In training with a device-resident dropout counter that kernel also closes the step's dropout
        stream, so callers of ``loss`` do NOT call ``_drop.advance()``."""
        bump = self._drop.dev_step if self.training else None
        loss, logp = ops.linear_log_softmax_nll(self.hidden(x), self.fc2.weight, self.fc2.bias, target, bump)
        return (loss, logp) if return_output else loss


class Net1(nn.Module):
    """Model-parallel stage 0: the two conv blocks, no dropout (reference MP.py:76-88)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 10, kernel_size=5)
        self.conv2 = nn.Conv2d(10, 20, kernel_size=5)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = ops.conv5_pool_relu(x, self.conv1.weight, self.conv1.bias)
        x = ops.conv5_pool_relu(x, self.conv2.weight, self.conv2.bias)
        return x.reshape(-1, 320)


class Net2(nn.Module):
    """Model-parallel stage 1: the two FC layers + log-softmax (reference MP.py:90-100)."""

    def __init__(self):
        super().__init__()
        self.fc1 = nn.Linear(320, 50)
        self.fc2 = nn.Linear(50, 10)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        x = ops.linear_f32(x, self.fc1.weight, self.fc1.bias, relu=True)
        return ops.linear_log_softmax(x, self.fc2.weight, self.fc2.bias)

    def loss(self, x: torch.Tensor, target: torch.Tensor, return_output: bool = False):
        """Last pipeline stage: activations in, loss out (reference MP.py:158-160), head and loss fused as in ``Net.loss``."""
        h = ops.linear_f32(x, self.fc1.weight, self.fc1.bias, relu=True)
        loss, logp = ops.linear_log_softmax_nll(h, self.fc2.weight, self.fc2.bias, target)
        return (loss, logp) if return_output else loss


def nll_loss(log_probs: torch.Tensor, target: torch.Tensor) -> torch.Tensor:
    """``F.nll_loss`` equivalent used by both reference loops (DP.py:152, MP.py:160)."""
    return ops.nll_loss(log_probs, target)


def mnist_stages():
    """Layer list for the generic pipeline partitioner (2 stages = the reference split)."""
    return [Net1(), Net2()]
