"""distributedpytorch_amd - MI355X-native distributed training framework.

Covers what EunjuYang/DistributedPyTorch's ``multinodeDP.py`` / ``multinodeMP.py``
demonstrate (multi-node data-parallel and model-parallel training launched one process
per GPU from a ``--GPU-list`` / ``--node-index`` description), re-designed for
MI355X: hand-written gfx950 HIP kernels for the hot ops, RCCL over xGMI with bucketed
all-reduce overlapped with backward, micro-batched pipeline stages, fused optimizers
over flat buffers, stock-format checkpoints.
"""
from .comm import (Communicator, init_process_group, destroy_process_group, get_rank, get_world_size,  # noqa: F401
                   build_grid)
from .launch import spawn, topology, Topology, ChildFailed  # noqa: F401
from .data import Partition, DataPartitioner, partition_dataset, BatchLoader  # noqa: F401
from .config import TrainConfig, parse_dp, parse_mp  # noqa: F401

__version__ = "0.1.0"


def set_deterministic(on: bool = True) -> None:
    """Bit-reproducible gradients for debugging (also ``DPA_DETERMINISTIC=1``): every float-atomic accumulation in the
    kernels (pixel-split wgrad, BatchNorm statistics in conv epilogues, the fused bottleneck tail, embedding
    gradients, the loss sum) takes a single-writer / fixed-order route instead.  Slower; see csrc/runtime.hip."""
    from .ops import _ext
    _ext.load().set_deterministic(bool(on))


def is_deterministic() -> bool:
    from .ops import _ext
    return bool(_ext.load().deterministic())
