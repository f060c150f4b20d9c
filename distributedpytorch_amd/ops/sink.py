"""Gradient sink: lets the wgrad kernels accumulate straight into the flat fp32
gradient buffer owned by the parallel engine.

Without a sink an op returns its weight gradient as a fresh tensor and autograd adds
it into ``param.grad`` (two extra kernels and one extra read+write per parameter).
With a sink installed (``parallel.flat.FlatParams.install_sink``) the kernel
atomically adds into the parameter's slice of the flat buffer and then reports the
parameter as ready, which is what drives the bucketed all-reduce overlap - the
role DDP's autograd hooks play, without the copy into bucket memory.
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import torch

_targets: Dict[int, torch.Tensor] = {}
_ready_cb: Optional[Callable[[torch.Tensor], None]] = None
_enabled = False


def install(targets: Dict[int, torch.Tensor], ready_cb: Optional[Callable[[torch.Tensor], None]] = None) -> None:
    global _targets, _ready_cb, _enabled
    _targets = dict(targets)
    _ready_cb = ready_cb
    _enabled = True


def set_ready_callback(cb: Optional[Callable[[torch.Tensor], None]]) -> None:
    global _ready_cb
    _ready_cb = cb


def uninstall() -> None:
    global _targets, _ready_cb, _enabled
    _targets, _ready_cb, _enabled = {}, None, False


def target(param: torch.Tensor) -> Optional[torch.Tensor]:
    """fp32 buffer slice to accumulate ``param``'s gradient into, or None."""
    if not _enabled:
        return None
    return _targets.get(id(param))


def ready(param: torch.Tensor) -> None:
    if _ready_cb is not None:
        _ready_cb(param)
