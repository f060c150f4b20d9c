"""Checkpoint / resume in the stock PyTorch ``state_dict`` format.

The reference has no checkpointing at all; the north star asks for ``torch.save``
state_dicts "in the same format", which can only mean the stock ``state_dict`` of the
reference's modules: an ``OrderedDict[str, Tensor]`` of fp32 CPU tensors keyed by the
attribute names of multinode-DP/multinodeDP.py:82-86 / multinode-MP/multinodeMP.py:81-95
(``conv1.weight [10,1,5,5]`` ... ``fc2.bias [10]``), loadable with plain ``torch.load`` +
``nn.Module.load_state_dict``.  Whatever layout the kernels use internally (KRSC filters,
channel padding, row padding of Linear weights, bf16 shadows, flat buffers) is
converted here, at the save/load edge:

* ``nn.Conv2d``-layout export of our NHWC ``Conv2d``: ``[K,R,S,Cpad] -> [K,C,R,S]``;
* padded ``Linear`` rows / padded vocab rows are sliced off;
* GPT-2 linears are exported as HuggingFace ``Conv1D`` weights (``[in, out]``);
* data-parallel: replicas are identical, rank 0 writes; pipeline: every stage's shard is
  gathered to rank 0 and merged into ONE state_dict (``Net1`` + ``Net2`` keys = ``Net`` keys).
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import nn as dnn


def export_state_dict(module: nn.Module, prefix: str = "", hf_conv1d: bool = False) -> "OrderedDict[str, torch.Tensor]":
    """Stock-layout fp32 CPU state_dict of ``module``."""
    out: "OrderedDict[str, torch.Tensor]" = OrderedDict()
    owners: Dict[int, nn.Module] = {}
    for mod in module.modules():
        for _, p in mod._parameters.items():
            if p is not None:
                owners.setdefault(id(p), mod)
    for name, t in module.state_dict().items():
        key = prefix + name
        v = t.detach().float().cpu().clone() if t.is_floating_point() else t.detach().cpu().clone()
        # find the owning module of this entry
        mod = module
        parts = name.split(".")
        try:
            for part in parts[:-1]:
                mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
        except (AttributeError, IndexError, KeyError, TypeError):
            mod = None
        leaf = parts[-1]
        if isinstance(mod, dnn.Conv2d) and leaf == "weight":
            v = v[..., : mod.in_channels].permute(0, 3, 1, 2).contiguous()          # KRSC(+pad) -> OIHW
        elif isinstance(mod, dnn.Linear):
            if leaf == "weight":
                v = v[: mod.out_features]
                if hf_conv1d:
                    v = v.t().contiguous()
            elif leaf == "bias":
                v = v[: mod.out_features]
        elif leaf == "wte" and hasattr(mod, "wte") and hasattr(module, "cfg"):
            v = v[: module.cfg.vocab_size]
        out[key] = v
    key_map = getattr(module, "_dpa_key_map", None)      # pipeline stage -> model-level key names
    if key_map:
        out = OrderedDict((prefix + key_map.get(k[len(prefix):], k[len(prefix):]), v) for k, v in out.items())
    return out


def import_state_dict(module: nn.Module, sd: Dict[str, torch.Tensor], strict: bool = True, hf_conv1d: bool = False):
    """Load a stock-layout state_dict into ``module`` (inverse of ``export_state_dict``).
    Copies IN PLACE, so parameters that were re-pointed into flat buffers stay there."""
    own = module.state_dict()
    key_map = getattr(module, "_dpa_key_map", None)
    if key_map:      # accept a merged, model-level state_dict on a pipeline stage
        inv = {g: l for l, g in key_map.items()}
        sd = {inv.get(k, k): v for k, v in sd.items()}
    missing = [k for k in own if k not in sd]
    unexpected = [k for k in sd if k not in own]
    if strict and (missing or unexpected):
        raise KeyError(f"state_dict mismatch: missing={missing[:5]} unexpected={unexpected[:5]}")
    with torch.no_grad():
        for name, dst in own.items():
            if name not in sd:
                continue
            src = sd[name]
            mod = module
            parts = name.split(".")
            try:
                for part in parts[:-1]:
                    mod = getattr(mod, part) if not part.isdigit() else mod[int(part)]
            except (AttributeError, IndexError, KeyError, TypeError):
                mod = None
            leaf = parts[-1]
            if isinstance(mod, dnn.Conv2d) and leaf == "weight":
                w = torch.zeros(dst.shape, dtype=torch.float32)
                w[..., : mod.in_channels] = src.float().permute(0, 2, 3, 1)
                src = w
            elif isinstance(mod, dnn.Linear) and leaf in ("weight", "bias"):
                s = src.float()
                if leaf == "weight" and hf_conv1d:
                    s = s.t()
                w = torch.zeros(dst.shape, dtype=torch.float32)
                w[: mod.out_features] = s
                src = w
            elif leaf == "wte" and src.shape != dst.shape:
                w = torch.zeros(dst.shape, dtype=torch.float32)
                w[: src.shape[0]] = src.float()
                src = w
            dst.copy_(src.to(dst.dtype).to(dst.device))
    return missing, unexpected


def _dropout_state(module: nn.Module) -> dict:
    """Positions of the counter-based dropout streams under ``module``: the MNIST nets' ``_DropState`` (one step
    counter, host- or device-resident) and GPT-2's ``_DropSites`` (a call counter per module - a pipeline stage that is
    one bare layer, e.g. the ``Embedding``, exposes one of these as ``module._drop`` too)."""
    out: dict = {}
    calls = {}
    for name, mod in module.named_modules():
        drop = getattr(mod, "_drop", None)
        if drop is None:
            continue
        if hasattr(drop, "dev_step") and hasattr(drop, "step"):
            if mod is module:
                out["dropout_step"] = int(drop.dev_step.item()) if drop.dev_step is not None else int(drop.step)
        elif hasattr(drop, "calls"):
            calls[name] = int(drop.calls)
    if calls:
        key_map = getattr(module, "_dpa_key_map", None)
        out["dropout_calls"] = calls if not key_map else {_model_level_module(name, key_map): v for name, v in calls.items()}
    return out


def _model_level_module(name: str, key_map: Dict[str, str]) -> str:
    """Model-level name of a stage-local sub-module, through any state_dict entry that lives under it."""
    prefix = name + "." if name else ""
    for local, glob in key_map.items():
        if local.startswith(prefix):
            tail = local[len(prefix):]
            if glob.endswith(tail):
                return glob[: len(glob) - len(tail)].rstrip(".")
    return name


def _restore_dropout_state(module: nn.Module, payload: dict) -> None:
    drop = getattr(module, "_drop", None)
    if drop is not None and hasattr(drop, "dev_step") and "dropout_step" in payload:
        drop.step = int(payload["dropout_step"])
        if drop.dev_step is not None:
            drop.dev_step.fill_(int(payload["dropout_step"]))
    calls = payload.get("dropout_calls")
    if calls:
        key_map = getattr(module, "_dpa_key_map", None)
        for name, mod in module.named_modules():
            d = getattr(mod, "_drop", None)
            if d is None or not hasattr(d, "calls") or hasattr(d, "dev_step"):
                continue
            key = _model_level_module(name, key_map) if key_map else name
            if key in calls:
                d.calls = int(calls[key])


def save_checkpoint(path: str, module: nn.Module, optimizer=None, epoch: int = 0, extra: Optional[dict] = None,
                    comm=None, pipeline_comm=None, flat=None) -> Optional[str]:
    """Write ``{"model": state_dict, "optimizer": ..., "epoch": ...}`` with ``torch.save``.

    ``comm``: data-parallel group (only its rank 0 writes).  ``pipeline_comm``: pipeline
    group - the stage shards are gathered to its rank 0 and merged.  Returns the path on
    the writing rank, None elsewhere.  ``path + ".model"`` additionally holds the bare
    model state_dict, which is the file a stock-PyTorch user ``torch.load``s."""
    sd = export_state_dict(module)
    opt_sd = optimizer.state_dict() if optimizer is not None and hasattr(optimizer, "state_dict") else None
    writer = True
    if pipeline_comm is not None and pipeline_comm.world_size > 1:
        shards = pipeline_comm.gather_object((sd, opt_sd, _dropout_state(module)), dst=0)
        if pipeline_comm.rank == 0:
            merged: "OrderedDict[str, torch.Tensor]" = OrderedDict()
            stage_drop: dict = {}
            for shard, _, dstate in shards:
                merged.update(shard)      # export_state_dict already names every entry by its model-level key
                stage_drop.setdefault("dropout_calls", {}).update(dstate.get("dropout_calls", {}))
                if "dropout_step" in dstate:
                    stage_drop["dropout_step"] = dstate["dropout_step"]
            sd = merged
            opt_sd = [o for _, o, _ in shards]
        else:
            writer = False
    if comm is not None and comm.world_size > 1 and comm.rank != 0:
        writer = False
    if not writer:
        return None
    os.makedirs(os.path.dirname(os.path.abspath(path)) or ".", exist_ok=True)
    payload = {"model": sd, "optimizer": opt_sd, "epoch": epoch, "format": "torch-state_dict-v1"}
    if extra:
        payload["extra"] = extra
    # counter-based dropout streams: resume draws the NEXT masks (pipeline: every stage's, gathered above)
    dstate = stage_drop if (pipeline_comm is not None and pipeline_comm.world_size > 1) else _dropout_state(module)
    payload.update({k: v for k, v in dstate.items() if v not in ({}, None)})
    tmp = path + ".tmp"
    torch.save(payload, tmp)
    os.replace(tmp, path)
    torch.save(sd, path + ".model")
    return path


def peek_extra(path: str) -> dict:
    """The ``extra`` record of a checkpoint (e.g. the pipeline's stage bounds) without touching any module."""
    payload = torch.load(path, map_location="cpu", weights_only=True)
    ex = payload.get("extra") if isinstance(payload, dict) else None
    return dict(ex) if isinstance(ex, dict) else {}


def load_checkpoint(path: str, module: nn.Module, optimizer=None, strict: bool = True, flat=None,
                    stage: Optional[int] = None) -> dict:
    """Resume: loads model (and optimizer) state; refreshes the bf16 shadows.  ``stage``: this rank's pipeline
    stage - a pipeline checkpoint stores one optimizer state per stage (a list), each stage takes its own back."""
    payload = torch.load(path, map_location="cpu", weights_only=True)
    sd = payload["model"] if isinstance(payload, dict) and "model" in payload else payload
    key_map = getattr(module, "_dpa_key_map", None)
    own_keys = set(key_map.values()) if key_map else set(module.state_dict().keys())
    sd_use = {k: v for k, v in sd.items() if k in own_keys} if (not strict or key_map) else sd
    import_state_dict(module, sd_use, strict=strict)
    if optimizer is not None and isinstance(payload, dict) and payload.get("optimizer") is not None:
        o = payload["optimizer"]
        if isinstance(o, (list, tuple)):
            o = o[stage] if (stage is not None and 0 <= stage < len(o)) else None
        if isinstance(o, dict):
            optimizer.load_state_dict(o)
    if flat is not None:
        flat.refresh_lp()
    if isinstance(payload, dict):
        _restore_dropout_state(module, payload)
    return payload if isinstance(payload, dict) else {"model": sd}
