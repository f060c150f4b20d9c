"""Model-parallel example: drop-in for the reference's ``multinodeMP.py`` launch lines.

    python -m distributedpytorch_amd.examples.mp --batch-size=256 --backend=rccl \\
        --node-index=0 --GPU-list=2 --IP=127.0.0.1 --portNum=39990

(reference multinode-MP/README.md:20,25).  Stage k runs on global rank k.  Extra flags:
``--micro-batches`` / ``--schedule`` (1 micro-batch + ``serial`` reproduces the
reference's lock-step schedule), ``--stages``, ``--dp-replicas``, ``--model``.
"""
from __future__ import annotations

import sys

from ..config import parse_mp
from ..launch import spawn
from ..train import run_mp


def main(argv=None):
    cfg = parse_mp(argv)
    spawn(run_mp, cfg.gpu_list, cfg.node_index, args=(cfg,))


if __name__ == "__main__":
    main(sys.argv[1:])
