"""Gradient sink: lets the wgrad kernels accumulate straight into the flat fp32
gradient buffer owned by the parallel engine.

Without a sink an op returns its weight gradient as a fresh tensor and autograd adds
it into ``param.grad`` (two extra kernels and one extra read+write per parameter).
With a sink installed (``parallel.flat.FlatParams.install_sink``) the kernel
atomically adds into the parameter's slice of the flat buffer and then reports the
parameter as ready, which is what drives the bucketed all-reduce overlap - the
role DDP's autograd hooks play, without the copy into bucket memory.

A sink is an OBJECT owned by one ``FlatParams``; every parameter it covers carries a
reference to it (``param._dpa_sink``), and the ops find it through the parameter they
are differentiating.  There is no process-wide state: several engines (two
``DataParallel`` wrappers, a pipeline stage beside a stand-alone model) each feed
their own flat buffer and their own ready callback.
"""
from __future__ import annotations

from typing import Callable, Dict, Iterable, Optional

import torch

_ATTR = "_dpa_sink"


class GradSink:
    """``targets``: id(param) -> fp32 view to accumulate into.  ``ready_cb(param)``: called after a kernel that
    accumulated ``param``'s gradient has been enqueued."""

    def __init__(self, params: Iterable[torch.Tensor], targets: Dict[int, torch.Tensor],
                 ready_cb: Optional[Callable[[torch.Tensor], None]] = None):
        self.params = list(params)
        self.targets = dict(targets)
        self.ready_cb = ready_cb
        self.enabled = True
        for p in self.params:
            setattr(p, _ATTR, self)

    def uninstall(self) -> None:
        self.enabled = False
        for p in self.params:
            if getattr(p, _ATTR, None) is self:
                delattr(p, _ATTR)
        self.params, self.targets, self.ready_cb = [], {}, None


def of(param: torch.Tensor) -> Optional[GradSink]:
    return getattr(param, _ATTR, None)


def target(param: torch.Tensor) -> Optional[torch.Tensor]:
    """fp32 buffer slice to accumulate ``param``'s gradient into, or None."""
    s = getattr(param, _ATTR, None)
    if s is None or not s.enabled:
        return None
    return s.targets.get(id(param))


def ready(param: torch.Tensor) -> None:
    s = getattr(param, _ATTR, None)
    if s is not None and s.ready_cb is not None:
        s.ready_cb(param)
