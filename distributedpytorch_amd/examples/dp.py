"""Data-parallel example: drop-in for the reference's ``multinodeDP.py`` launch line.

    python -m distributedpytorch_amd.examples.dp --dist-url=tcp://MASTER_IP:PORT \\
        --backend=rccl --node-index=0 --GPU-list=4,4

(reference multinode-DP/README.md:15).  One invocation per node; one process per GPU of
that node; global rank = offset of the node + local rank (DP.py:170-178).
"""
from __future__ import annotations

import sys

from ..config import parse_dp
from ..launch import spawn
from ..train import run_dp


def main(argv=None):
    cfg = parse_dp(argv)
    spawn(run_dp, cfg.gpu_list, cfg.node_index, args=(cfg,))


if __name__ == "__main__":
    main(sys.argv[1:])
