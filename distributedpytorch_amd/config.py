"""Flag / config system.

Keeps the flag *names* of the reference scripts so a user can switch launch lines
unchanged (reference: multinode-DP/multinodeDP.py:24-40 and
multinode-MP/multinodeMP.py:20-40), but every flag is real here: ``--epochs``,
``--lr``, ``--momentum`` and ``--seed`` are honoured instead of being hard-coded
(the reference ignores ``--epochs`` and hard-codes 10 / 100 epochs, DP.py:146,
MP.py:126).  The parsed result is a picklable dataclass that is handed to the
children explicitly, so the *spawn* start method works (the reference relies on
``fork`` + a module global, DP.py:166).
"""
from __future__ import annotations

import argparse
import os
from dataclasses import dataclass, field, asdict
from typing import List, Optional, Sequence


def parse_gpu_list(text) -> List[int]:
    """``"4,4"`` -> ``[4, 4]``.  Accepts an int / list as well (the reference's
    default is the int 0, which crashes on ``.split`` - DP.py:37,171)."""
    if isinstance(text, (list, tuple)):
        out = [int(v) for v in text]
    elif isinstance(text, int):
        out = [text]
    else:
        out = [int(item) for item in str(text).split(",") if item.strip() != ""]
    if not out or any(v < 0 for v in out) or sum(out) == 0:
        raise ValueError(f"--GPU-list must name at least one process, got {text!r}")
    return out


def normalize_dist_url(url: str) -> str:
    """The reference's default ``127.0.0.1:8885`` has no scheme and is rejected by
    c10d (DP.py:27); accept it and add ``tcp://``."""
    if "://" not in url:
        return "tcp://" + url
    return url


@dataclass
class TrainConfig:
    # ---- topology (reference flags) ----
    gpu_list: List[int] = field(default_factory=lambda: [1])
    node_index: int = 0
    backend: str = "auto"          # auto -> rccl when GPUs are visible else gloo
    dist_url: str = "tcp://127.0.0.1:8885"
    # ---- training ----
    epochs: int = 10
    batch_size: int = 128          # DP: global batch (DP.py:109); MP: per-step batch (MP.py:35)
    lr: float = 0.01
    momentum: float = 0.5
    weight_decay: float = 0.0
    seed: int = 1234
    workers: int = 0
    device_data: bool = False      # synthetic image batches generated on the GPU instead of on the host
    # ---- model / data ----
    model: str = "mnist_cnn"
    dataset_size: int = 60000
    data_dir: Optional[str] = None  # MNIST IDX files (train-images-idx3-ubyte ...); absent -> synthetic data of the same shape
    dropout: float = 0.0           # GPT-2: embd / attn / resid dropout probability (the MNIST nets keep the reference's 0.5)
    # ---- engine ----
    bucket_mb: float = 32.0
    first_bucket_mb: float = 4.0
    shard_optimizer: bool = False  # ZeRO stage 1 (optim.ShardedSGD)
    graph: bool = True             # DP, small CNN on a GPU: replay forward+backward from a hipGraph
    optimizer: str = "sgd"         # sgd | adamw
    opt_overlap: str = "auto"      # optimizer per bucket behind its all-reduce during backward: auto (model families where it was measured to pay) | on | off
    accum_steps: int = 1           # micro-steps accumulated locally per optimizer step
    clip_grad_norm: float = 0.0    # global-norm clipping (0 = off)
    warmup_steps: int = 0          # linear LR warm-up; with cosine_steps > 0 (or warm-up alone) a cosine decay follows
    cosine_steps: int = 0
    min_lr: float = 0.0
    micro_batches: int = 1
    schedule: str = "1f1b"
    stages: int = 2
    dp_replicas: int = 1
    steps_per_epoch: Optional[int] = None
    checkpoint: Optional[str] = None
    resume: Optional[str] = None
    log_json: Optional[str] = None
    timeout_s: float = 300.0
    sync_debug: bool = False
    breakdown: bool = False        # per-phase HIP-event timers (forward / backward / exposed comm / optimizer) in the JSONL
    device: str = "auto"           # auto | cuda | cpu

    # -------- derived topology (reference: DP.py:170-174) --------
    @property
    def local_size(self) -> int:
        return self.gpu_list[self.node_index]

    @property
    def world_size(self) -> int:
        return sum(self.gpu_list)

    @property
    def rank_offset(self) -> int:
        return sum(self.gpu_list[: self.node_index])

    def to_dict(self):
        return asdict(self)


def _common_flags(p: argparse.ArgumentParser) -> None:
    # Reference flags (kept verbatim so launch lines carry over).
    p.add_argument("--workers", default=0, type=int, metavar="N",
                   help="host threads assembling batches ahead of the training loop (honoured, unlike the reference)")
    p.add_argument("--gpu", default=None, type=int, help="accepted for compatibility; device = local rank")
    p.add_argument("--rank", default=0, type=int, help="accepted for compatibility; rank comes from --GPU-list/--node-index")
    p.add_argument("--epochs", default=None, type=int, help="number of epochs (honoured, unlike the reference)")
    p.add_argument("--backend", default="auto", type=str, help="auto | rccl | nccl | gloo")
    p.add_argument("--GPU-list", default="1", type=str,
                   help="processes per node, e.g. --GPU-list=4,4 (heterogeneous lists such as 4,2,1 work)")
    p.add_argument("--node-index", default=0, type=int, help="index of this node inside --GPU-list")
    # New, real hyper-parameters.
    p.add_argument("--lr", default=None, type=float)
    p.add_argument("--momentum", default=None, type=float)
    p.add_argument("--weight-decay", default=0.0, type=float)
    p.add_argument("--seed", default=1234, type=int)
    p.add_argument("--dropout", default=0.0, type=float, help="GPT-2 models: embedding / attention / residual dropout probability")
    p.add_argument("--model", default=None, type=str)
    p.add_argument("--dataset-size", default=60000, type=int)
    p.add_argument("--data-dir", default=None, type=str,
                   help="directory holding the MNIST IDX files (the reference's datasets.MNIST('./data'), DP.py:100; "
                        "never downloaded); without them the MNIST models train on synthetic 1x28x28 data")
    p.add_argument("--breakdown", action="store_true",
                   help="time forward / backward / exposed communication / optimizer per step with HIP events and write "
                        "them to the epoch line and --log-json")
    p.add_argument("--steps-per-epoch", default=None, type=int)
    p.add_argument("--device", default="auto", type=str)
    p.add_argument("--checkpoint", default=None, type=str, help="path to write a state_dict checkpoint")
    p.add_argument("--resume", default=None, type=str, help="path of a checkpoint to resume from")
    p.add_argument("--log-json", default=None, type=str, help="append JSON-lines metrics to this file")
    p.add_argument("--timeout", default=300.0, type=float, help="rendezvous / collective timeout in seconds")
    p.add_argument("--sync-debug", action="store_true", help="synchronise after every comm op and check grads")
    p.add_argument("--device-data", action="store_true",
                   help="generate the synthetic image batches on the GPU (no host work, no H2D copy)")
    p.add_argument("--bucket-mb", default=32.0, type=float, help="gradient bucket size of the data-parallel reduction (MiB)")
    p.add_argument("--first-bucket-mb", default=4.0, type=float, help="size of the first bucket launched during backward (MiB)")
    p.add_argument("--ifname", default=None, type=str,
                   help="network interface for inter-node traffic: exported as NCCL_SOCKET_IFNAME and GLOO_SOCKET_IFNAME "
                        "(the reference's README asks for `export GLOO_SOCKET_IFNAME=eth0` by hand, multinode-MP/README.md:12-14)")


def dp_parser() -> argparse.ArgumentParser:
    """Parser for the data-parallel example (reference: DP.py:24-40)."""
    p = argparse.ArgumentParser(description="MI355X data-parallel training")
    _common_flags(p)
    p.add_argument("--dist-url", default="tcp://127.0.0.1:8885", type=str,
                   help="rendezvous url; a bare host:port is accepted")
    p.add_argument("--batch-size", default=128, type=int, help="GLOBAL batch (reference fixes 128, DP.py:109)")
    p.add_argument("--shard-optimizer", action="store_true", help="optimizer state sharded over the ranks (ZeRO stage 1)")
    p.add_argument("--no-graph", action="store_true", help="small CNN on a GPU: do not replay the step from a hipGraph")
    p.add_argument("--optimizer", default="sgd", choices=["sgd", "adamw"])
    p.add_argument("--opt-overlap", default="auto", choices=["auto", "on", "off"],
                   help="update each gradient bucket behind its all-reduce while backward is still running "
                        "(constant-lr SGD without clipping only); auto = the model families where it was measured to pay")
    p.add_argument("--accum-steps", default=1, type=int, help="gradient accumulation: micro-steps per optimizer step")
    p.add_argument("--clip-grad-norm", default=0.0, type=float, help="global gradient-norm clipping (0 = off)")
    p.add_argument("--warmup-steps", default=0, type=int, help="linear LR warm-up steps (then cosine decay)")
    p.add_argument("--cosine-steps", default=0, type=int, help="total steps of the cosine schedule (default: the whole run)")
    p.add_argument("--min-lr", default=0.0, type=float)
    return p


def mp_parser() -> argparse.ArgumentParser:
    """Parser for the model-parallel example (reference: MP.py:20-40)."""
    p = argparse.ArgumentParser(description="MI355X model-parallel (pipeline) training")
    _common_flags(p)
    p.add_argument("--batch-size", default=1024, type=int, help="batch size per step (MP.py:35)")
    p.add_argument("--IP", default="127.0.0.1", type=str, help="master address")
    p.add_argument("--portNum", default="39990", type=str, help="master port")
    p.add_argument("--micro-batches", default=1, type=int, help="1 reproduces the reference's serial schedule")
    p.add_argument("--schedule", default="1f1b", choices=["serial", "gpipe", "1f1b"])
    p.add_argument("--stages", default=2, type=int)
    p.add_argument("--dp-replicas", default=1, type=int, help="pipeline replicas (DP x PP hybrid)")
    return p


def config_from_args(args: argparse.Namespace, mode: str) -> TrainConfig:
    gpu_list = parse_gpu_list(args.GPU_list)
    if not 0 <= args.node_index < len(gpu_list):
        raise ValueError(f"--node-index {args.node_index} outside --GPU-list {gpu_list}")
    if mode == "dp":
        url = normalize_dist_url(args.dist_url)
        defaults = dict(epochs=10, lr=0.01, momentum=0.5, model="mnist_cnn")
    else:
        url = f"tcp://{args.IP}:{args.portNum}"
        defaults = dict(epochs=100, lr=0.01, momentum=0.0, model="mnist_cnn")
    cfg = TrainConfig(
        gpu_list=gpu_list, node_index=args.node_index, backend=args.backend, dist_url=url,
        epochs=args.epochs if args.epochs is not None else defaults["epochs"],
        batch_size=args.batch_size,
        lr=args.lr if args.lr is not None else defaults["lr"],
        momentum=args.momentum if args.momentum is not None else defaults["momentum"],
        weight_decay=args.weight_decay, seed=args.seed, workers=args.workers, dropout=args.dropout,
        model=args.model or defaults["model"], dataset_size=args.dataset_size,
        steps_per_epoch=args.steps_per_epoch, device=args.device,
        checkpoint=args.checkpoint, resume=args.resume, log_json=args.log_json,
        timeout_s=args.timeout, sync_debug=args.sync_debug, device_data=args.device_data,
        data_dir=args.data_dir, breakdown=args.breakdown,
    )
    cfg.bucket_mb = args.bucket_mb
    cfg.first_bucket_mb = args.first_bucket_mb
    if mode == "dp":
        cfg.shard_optimizer = args.shard_optimizer
        cfg.graph = not args.no_graph
        cfg.optimizer, cfg.accum_steps, cfg.clip_grad_norm = args.optimizer, args.accum_steps, args.clip_grad_norm
        cfg.opt_overlap = args.opt_overlap
        cfg.warmup_steps, cfg.cosine_steps, cfg.min_lr = args.warmup_steps, args.cosine_steps, args.min_lr
    else:
        cfg.micro_batches = args.micro_batches
        cfg.schedule = args.schedule
        cfg.stages = args.stages
        cfg.dp_replicas = args.dp_replicas
    if getattr(args, "ifname", None):        # before any process is spawned: the children inherit it
        os.environ["NCCL_SOCKET_IFNAME"] = args.ifname
        os.environ["GLOO_SOCKET_IFNAME"] = args.ifname
    return cfg


def parse_dp(argv: Optional[Sequence[str]] = None) -> TrainConfig:
    return config_from_args(dp_parser().parse_args(argv), "dp")


def parse_mp(argv: Optional[Sequence[str]] = None) -> TrainConfig:
    return config_from_args(mp_parser().parse_args(argv), "mp")
