"""Loader for the in-tree gfx950 extension ``distributedpytorch_amd._C``.

The HIP kernels are the ONLY GPU implementation of the ops in this package: when a
GPU is visible and the extension cannot be imported we fail loudly instead of
silently falling back to stock PyTorch operators.  On a machine without a GPU the
plain-PyTorch CPU path (used as the numerics oracle and for the gloo plumbing
tests) does not need the extension, so a missing build is tolerated there.
"""
from __future__ import annotations

import importlib
import torch

_C = None
_ERR = None


def load():
    """Return the extension module (importing it on first use) or raise."""
    global _C, _ERR
    if _C is not None:
        return _C
    try:
        _C = importlib.import_module("distributedpytorch_amd._C")
    except Exception as e:  # noqa: BLE001
        _ERR = e
        raise RuntimeError(
            "distributedpytorch_amd._C (gfx950 HIP kernels) is not built or failed to load: "
            f"{e!r}.  Build it in-tree with `python -m distributedpytorch_amd.build_ext` "
            "(or `python -c 'import __graft_entry__ as g; g.build()'`)."
        ) from e
    return _C


def available() -> bool:
    try:
        load()
        return True
    except RuntimeError:
        return False


def use_hip(t: torch.Tensor) -> bool:
    """True when ``t`` lives on a GPU: the op must run on our kernels."""
    if t.is_cuda:
        load()  # raises when missing: no silent fallback on a GPU box
        return True
    return False
