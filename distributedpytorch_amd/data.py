"""Dataset partitioning and synthetic datasets.

``Partition`` / ``DataPartitioner`` keep the reference's semantics
(multinode-DP/multinodeDP.py:42-75): a ``random.Random`` seeded with 1234 shuffles
``range(len(data))`` and consecutive chunks of ``int(frac * len)`` indices are cut
per entry of ``sizes``.  Because every rank uses the same seed they all compute the
same permutation, so the shards are disjoint and consistent with no communication.

torchvision is not available on the target image and there is no network, so the
datasets are synthetic tensors of the named shapes (MNIST 1x28x28, ImageNet
3x224x224, GPT-2 token streams), generated deterministically from a seed - on the
device when one is given, so the input pipeline can never stall the GPU.
"""
from __future__ import annotations

from math import ceil
from random import Random
from typing import List, Optional, Sequence, Tuple

import torch


class Partition:
    """Dataset-like view of a subset of ``data`` (reference DP.py:42-55)."""

    def __init__(self, data, index: Sequence[int]):
        self.data = data
        self.index = index

    def __len__(self) -> int:
        return len(self.index)

    def __getitem__(self, i):
        return self.data[self.index[i]]


class DataPartitioner:
    """Seeded, communication-free partitioning (reference DP.py:57-75)."""

    def __init__(self, data, sizes: Sequence[float] = (0.7, 0.2, 0.1), seed: int = 1234):
        self.data = data
        self.partitions: List[List[int]] = []
        rng = Random()
        rng.seed(seed)
        n = len(data)
        order = list(range(n))
        rng.shuffle(order)
        start = 0
        for frac in sizes:
            part_len = int(frac * n)
            self.partitions.append(order[start:start + part_len])
            start += part_len

    def use(self, partition: int) -> Partition:
        return Partition(self.data, self.partitions[partition])


class SyntheticImageDataset:
    """Deterministic fake images + labels, materialised lazily per index block.

    Samples are a pure function of ``(seed, index)`` so any rank can produce any
    sample; class-dependent means make the task learnable (loss must go down in the
    end-to-end tests).
    """

    def __init__(self, n: int = 60000, shape: Tuple[int, int, int] = (1, 28, 28), classes: int = 10,
                 seed: int = 1234, device: Optional[torch.device] = None, on_device: bool = False):
        self.n, self.shape, self.classes, self.seed = n, tuple(shape), classes, seed
        self.device = device or torch.device("cpu")
        g = torch.Generator().manual_seed(seed)
        self.labels = torch.randint(0, classes, (n,), generator=g)
        # one low-rank "prototype" per class
        self.proto = torch.randn(classes, *self.shape, generator=g) * 0.5
        # on_device: batches are GENERATED on the GPU (prototype gather + device noise, a pure function of (seed, first
        # index) there too) - a synthetic source has no bytes to read, so the host should not be its bottleneck: at
        # 3x224x224 one host thread draws ~1,000 images/s of Gaussian noise.  Values differ from the host generator's.
        self.on_device = bool(on_device) and self.device.type == "cuda"
        if self.on_device:
            self._labels_dev = self.labels.to(self.device)
            self._proto_dev = self.proto.to(self.device)

    def __len__(self) -> int:
        return self.n

    def __getitem__(self, i: int):
        y = int(self.labels[i])
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(i))
        x = self.proto[y] + 0.3 * torch.randn(self.shape, generator=g)
        return x, y

    def batch(self, indices: Sequence[int]):
        idx = torch.as_tensor(list(indices), dtype=torch.long)
        if self.on_device and len(idx):
            y = self._labels_dev[idx.to(self.device, non_blocking=True)]
            g = torch.Generator(device=self.device).manual_seed(self.seed * 1000003 + int(idx[0]))
            noise = torch.randn((len(idx),) + self.shape, generator=g, device=self.device)
            return self._proto_dev[y].add_(noise, alpha=0.3), y
        y = self.labels[idx]
        g = torch.Generator().manual_seed(self.seed * 1000003 + int(idx[0]) if len(idx) else self.seed)
        x = self.proto[y] + 0.3 * torch.randn((len(idx),) + self.shape, generator=g)
        return x.to(self.device, non_blocking=True), y.to(self.device, non_blocking=True)


MNIST_MEAN, MNIST_STD = 0.1307, 0.3081      # transforms.Normalize((0.1307,), (0.3081,)), DP.py:105 / MP.py:107


def _read_idx(path: str) -> torch.Tensor:
    """One IDX file (the MNIST distribution format; ``.gz`` accepted): magic = 0, 0, type code, rank; big-endian
    uint32 dims; row-major payload.  Only the unsigned-byte type (0x08) MNIST uses is read."""
    import gzip
    import struct
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        head = f.read(4)
        if len(head) != 4 or head[0] != 0 or head[1] != 0:
            raise ValueError(f"{path}: not an IDX file (magic {head!r})")
        if head[2] != 0x08:
            raise ValueError(f"{path}: IDX element type 0x{head[2]:02x} is not unsigned byte")
        rank = head[3]
        dims = struct.unpack(">" + "I" * rank, f.read(4 * rank))
        count = 1
        for d in dims:
            count *= d
        payload = f.read(count)
        if len(payload) != count:
            raise ValueError(f"{path}: truncated IDX payload ({len(payload)} of {count} bytes)")
    return torch.frombuffer(bytearray(payload), dtype=torch.uint8).reshape(dims)


def find_mnist(root: Optional[str], train: bool = True) -> Optional[Tuple[str, str]]:
    """(images, labels) IDX paths of the MNIST split under ``root`` - in ``root`` itself, ``root/MNIST/raw``
    (torchvision's layout: ``datasets.MNIST('./data', download=True)``, DP.py:100) or ``root/raw`` - or None.
    Never downloads."""
    import os
    if not root:
        return None
    stem = "train" if train else "t10k"
    for sub in ("", os.path.join("MNIST", "raw"), "raw"):
        for sep in ("-", "."):
            for ext in ("", ".gz"):
                img = os.path.join(root, sub, f"{stem}-images{sep}idx3-ubyte{ext}")
                lab = os.path.join(root, sub, f"{stem}-labels{sep}idx1-ubyte{ext}")
                if os.path.isfile(img) and os.path.isfile(lab):
                    return img, lab
    return None


class MNISTIdx:
    """The reference's dataset (multinode-DP/multinodeDP.py:100-107, multinode-MP/multinodeMP.py:104-110): MNIST from
    its IDX files with ``ToTensor()`` + ``Normalize((0.1307,), (0.3081,))``.  There is no torchvision and no network
    here, so the files are the user's (``--data-dir``); nothing is ever downloaded.  The whole split (47 MB of bytes)
    is held as one uint8 tensor and normalised per batch - on the GPU when ``device`` is one, so the host only
    gathers bytes."""

    shape, classes = (1, 28, 28), 10

    def __init__(self, root: str, train: bool = True, device: Optional[torch.device] = None):
        paths = find_mnist(root, train)
        if paths is None:
            raise FileNotFoundError(f"no MNIST IDX files ({'train' if train else 't10k'}-images-idx3-ubyte[.gz] + "
                                    f"labels) under {root!r}")
        self.images = _read_idx(paths[0])                   # [n, 28, 28] uint8
        self.labels = _read_idx(paths[1]).to(torch.long)    # [n]
        if self.images.dim() != 3 or self.images.shape[0] != self.labels.shape[0]:
            raise ValueError(f"{paths[0]}: {tuple(self.images.shape)} images vs {tuple(self.labels.shape)} labels")
        self.shape = (1,) + tuple(self.images.shape[1:])
        self.n = int(self.images.shape[0])
        self.device = device or torch.device("cpu")
        self.paths = paths

    def __len__(self) -> int:
        return self.n

    @staticmethod
    def _normalise(raw: torch.Tensor) -> torch.Tensor:
        # ToTensor: uint8 -> [0, 1] float, channel axis first; Normalize: (x - mean) / std
        return raw.to(torch.float32).div_(255.0).sub_(MNIST_MEAN).div_(MNIST_STD).unsqueeze(-3)

    def __getitem__(self, i: int):
        return self._normalise(self.images[i]), int(self.labels[i])

    def batch(self, indices: Sequence[int]):
        idx = torch.as_tensor(list(indices), dtype=torch.long)
        raw, y = self.images[idx], self.labels[idx]
        if self.device.type == "cuda":
            raw, y = raw.to(self.device, non_blocking=True), y.to(self.device, non_blocking=True)
        return self._normalise(raw), y


def mnist_or_synthetic(data_dir: Optional[str], n: int = 60000, seed: int = 1234, device: Optional[torch.device] = None,
                       on_device: bool = False, log=None):
    """The training set of the two reference scripts: real MNIST when ``data_dir`` holds the IDX files, else the
    synthetic MNIST-shaped set (and say which)."""
    if data_dir:
        if find_mnist(data_dir, True) is not None:
            ds = MNISTIdx(data_dir, True, device=device if on_device else None)
            if log is not None:
                log(f"MNIST: {len(ds)} images from {ds.paths[0]}")
            return ds
        if log is not None:
            log(f"no MNIST IDX files under {data_dir!r}: synthetic 1x28x28 data instead")
    return SyntheticImageDataset(n=n, shape=(1, 28, 28), classes=10, seed=seed, device=device, on_device=on_device)


class SyntheticTokenDataset:
    """Token sequences for GPT-2 shaped workloads: ``(tokens[S], targets[S])``."""

    def __init__(self, n: int = 4096, seq_len: int = 1024, vocab: int = 50257, seed: int = 1234):
        self.n, self.seq_len, self.vocab, self.seed = n, seq_len, vocab, seed

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        g = torch.Generator().manual_seed(self.seed * 7919 + int(i))
        t = torch.randint(0, self.vocab, (self.seq_len + 1,), generator=g)
        return t[:-1], t[1:]

    def batch(self, indices, device=None):
        xs, ys = zip(*(self[i] for i in indices))
        x, y = torch.stack(xs), torch.stack(ys)
        if device is not None:
            x, y = x.to(device), y.to(device)
        return x, y


class BatchLoader:
    """Minimal loader over a ``Partition``: optional per-epoch seeded shuffle,
    ragged last batch kept (the reference's DataLoader has no ``drop_last``,
    DP.py:113-114, so the last batch is short: 12 at world 8)."""

    def __init__(self, partition, batch_size: int, shuffle: bool = True, seed: int = 0, drop_last: bool = False,
                 device: Optional[torch.device] = None):
        if batch_size < 1:
            raise ValueError("batch_size must be >= 1 (global batch smaller than world size?)")
        self.partition, self.batch_size, self.shuffle = partition, int(batch_size), shuffle
        self.seed, self.drop_last, self.device = seed, drop_last, device
        self.epoch = 0
        self.dataset = partition  # attribute name used by the reference (DP.py:145)

    def __len__(self) -> int:
        n = len(self.partition)
        return n // self.batch_size if self.drop_last else ceil(n / self.batch_size)

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def plan(self) -> List[List[int]]:
        """Dataset indices of every batch of the current epoch, in order."""
        n = len(self.partition)
        order = list(range(n))
        if self.shuffle:
            Random(self.seed * 9973 + self.epoch).shuffle(order)
        index = self.partition.index if isinstance(self.partition, Partition) else None
        out = []
        for lo in range(0, n, self.batch_size):
            chunk = order[lo:lo + self.batch_size]
            if self.drop_last and len(chunk) < self.batch_size:
                break
            out.append([index[i] for i in chunk] if index is not None else chunk)
        return out

    def fetch(self, ids: Sequence[int]):
        """Assemble one batch (thread-safe: reads the dataset only)."""
        base = self.partition.data if isinstance(self.partition, Partition) else self.partition
        if hasattr(base, "batch"):
            x, y = base.batch(ids)
        else:
            xs, ys = zip(*(base[i] for i in ids))
            x = torch.stack([torch.as_tensor(v) for v in xs])
            y = torch.as_tensor(ys)
        if self.device is not None:
            x, y = x.to(self.device, non_blocking=True), y.to(self.device, non_blocking=True)
        return x, y

    def __iter__(self):
        for ids in self.plan():
            yield self.fetch(ids)
        self.epoch += 1


class PrefetchLoader:
    """Runs a batch loader ``depth`` batches ahead of the training loop.

    A producer thread assembles batch N+1.. on the host while the GPU works on batch N, stages it in pinned memory and
    issues the host-to-device copy on a dedicated copy stream; the training stream only waits on that copy's event.
    With the loader in line (the reference: ``for data, target in train_set`` + ``.to(device)``, DP.py:147-149) a
    ResNet-50 step at batch 64 spent 60 of its 67 ms assembling and copying the batch with the GPU idle.

    Order, content and ragged last batch are exactly the wrapped loader's.  An exception in the producer is re-raised
    in the consumer; leaving the loop early (``break``) stops the producer.
    """

    _END = object()

    def __init__(self, loader, device: Optional[torch.device] = None, depth: int = 2, workers: int = 1):
        self.loader, self.depth, self.workers = loader, max(1, int(depth)), max(1, int(workers))
        self.device = torch.device(device) if device is not None else None
        self._copy_stream = None

    def __len__(self) -> int:
        return len(self.loader)

    def __getattr__(self, name):              # dataset / partition / batch_size / epoch of the wrapped loader
        return getattr(self.__dict__["loader"], name)

    def set_epoch(self, epoch: int) -> None:
        self.loader.set_epoch(epoch)

    def _stage(self, batch):
        """Pinned staging + host-to-device copy on the copy stream (any thread)."""
        if self.device is None or self.device.type != "cuda":
            return batch
        torch.cuda.set_device(self.device)
        host = tuple(t if t.is_cuda else t.pin_memory() for t in batch)
        with torch.cuda.stream(self._copy_stream):
            dev = tuple(t.to(self.device, non_blocking=True) for t in host)
            done = torch.cuda.Event()
            done.record(self._copy_stream)
        return dev, done, host                               # host buffers stay referenced until the consumer took the batch

    def _batches(self, stop):
        """Staged batches in loader order.  With ``workers > 1`` and a loader that can assemble batches out of order
        (``plan`` / ``fetch``), up to ``workers`` batches are assembled at once - ``--workers`` of the reference's CLI,
        which it parses and never uses (DP.py:24-26)."""
        if self.workers > 1 and hasattr(self.loader, "plan") and hasattr(self.loader, "fetch"):
            from collections import deque
            from concurrent.futures import ThreadPoolExecutor
            plan = self.loader.plan()
            with ThreadPoolExecutor(self.workers, thread_name_prefix="dpa-fetch") as pool:
                ahead: deque = deque()
                nxt = 0
                while nxt < len(plan) or ahead:
                    while nxt < len(plan) and len(ahead) < self.workers + self.depth and not stop.is_set():
                        ahead.append(pool.submit(lambda ids=plan[nxt]: self._stage(self.loader.fetch(ids))))
                        nxt += 1
                    if stop.is_set():
                        for f in ahead:
                            f.cancel()
                        return
                    yield ahead.popleft().result()
            self.loader.epoch += 1
            return
        for batch in self.loader:
            yield self._stage(batch)

    def _produce(self, q, stop) -> None:
        import queue
        cuda = self.device is not None and self.device.type == "cuda"
        try:
            if cuda:
                torch.cuda.set_device(self.device)
                if self._copy_stream is None:
                    self._copy_stream = torch.cuda.Stream(device=self.device)
            for item in self._batches(stop):
                while not stop.is_set():
                    try:
                        q.put(item, timeout=0.1)
                        break
                    except queue.Full:
                        continue
                if stop.is_set():
                    return
            item = self._END
        except BaseException as exc:                        # noqa: BLE001 - handed to the consumer
            item = exc
        while not stop.is_set():
            try:
                q.put(item, timeout=0.1)
                return
            except queue.Full:
                continue

    def __iter__(self):
        import queue
        import threading
        q: "queue.Queue" = queue.Queue(maxsize=self.depth)
        stop = threading.Event()
        worker = threading.Thread(target=self._produce, args=(q, stop), name="dpa-prefetch", daemon=True)
        worker.start()
        try:
            while True:
                item = q.get()
                if item is self._END:
                    return
                if isinstance(item, BaseException):
                    raise item
                if self.device is not None and self.device.type == "cuda":
                    dev, done, _host = item
                    cur = torch.cuda.current_stream(self.device)
                    cur.wait_event(done)
                    for t in dev:
                        t.record_stream(cur)                 # allocated on the copy stream, consumed on this one
                    yield dev
                else:
                    yield item
        finally:
            stop.set()
            worker.join(timeout=10.0)


def partition_dataset(dataset, rank: int, world_size: int, global_batch: int = 128, seed: int = 1234,
                      device: Optional[torch.device] = None, shuffle: bool = True, workers: int = 1):
    """Reference ``partition_dataset`` (DP.py:98-116): equal shards, global batch
    fixed, per-rank batch = ``int(global_batch / world_size)``.  Returns
    ``(loader, bsz)`` with ``bsz`` the *float* per-rank batch, as the reference does
    (it is used for ``num_batches = ceil(len(partition) / bsz)``, DP.py:145)."""
    bsz = global_batch / float(world_size)
    sizes = [1.0 / world_size for _ in range(world_size)]
    part = DataPartitioner(dataset, sizes, seed=seed).use(rank)
    if device is not None and torch.device(device).type == "cuda":
        # host assembly + pinned staging + copy-stream H2D run ahead of the training loop
        loader = PrefetchLoader(BatchLoader(part, int(bsz), shuffle=shuffle, seed=seed + rank, device=None), device,
                                workers=workers)
    else:
        loader = BatchLoader(part, int(bsz), shuffle=shuffle, seed=seed + rank, device=device)
    return loader, bsz


def synthetic_batch(kind: str, batch: int, device, dtype=torch.float32, seed: int = 0, seq_len: int = 1024,
                    vocab: int = 50257, classes: int = 1000):
    """One on-device synthetic batch of a named shape, for benchmarks."""
    g = torch.Generator(device=device).manual_seed(seed)
    if kind == "mnist":
        x = torch.randn(batch, 1, 28, 28, generator=g, device=device, dtype=torch.float32).to(dtype)
        y = torch.randint(0, 10, (batch,), generator=g, device=device)
    elif kind == "imagenet":
        x = torch.randn(batch, 3, 224, 224, generator=g, device=device, dtype=torch.float32).to(dtype)
        y = torch.randint(0, classes, (batch,), generator=g, device=device)
    elif kind == "tokens":
        t = torch.randint(0, vocab, (batch, seq_len + 1), generator=g, device=device)
        x, y = t[:, :-1].contiguous(), t[:, 1:].contiguous()
    else:
        raise ValueError(kind)
    return x, y
