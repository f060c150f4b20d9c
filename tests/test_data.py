"""T0: Partition / DataPartitioner semantics (reference DP.py:42-75)."""
from random import Random

import pytest
import torch

from distributedpytorch_amd import data


def _reference_partitions(n, sizes, seed=1234):
    """The reference algorithm, restated independently (DP.py:60-72)."""
    rng = Random(); rng.seed(seed)
    idx = list(range(n)); rng.shuffle(idx)
    out = []
    for frac in sizes:
        k = int(frac * n)
        out.append(idx[:k]); idx = idx[k:]
    return out


def test_partitioner_matches_reference_and_is_disjoint():
    n = 60000
    for world in (1, 2, 8):
        sizes = [1.0 / world] * world
        p = data.DataPartitioner(list(range(n)), sizes)
        assert p.partitions == _reference_partitions(n, sizes)
        seen = set()
        for i in range(world):
            part = p.use(i)
            assert len(part) == int(n / world)              # 7,500 per rank at world 8
            s = set(part.index)
            assert not (s & seen)
            seen |= s
        assert len(seen) == int(n / world) * world
    p = data.DataPartitioner(list(range(100)))                # default sizes .7/.2/.1
    assert [len(x) for x in p.partitions] == [70, 20, 10]


def test_partition_view():
    base = [10, 11, 12, 13, 14]
    part = data.Partition(base, [4, 0, 2])
    assert len(part) == 3 and [part[i] for i in range(3)] == [14, 10, 12]


def test_partition_dataset_reference_batching():
    ds = data.SyntheticImageDataset(60000, (1, 28, 28), 10)
    loader, bsz = data.partition_dataset(ds, rank=3, world_size=8, global_batch=128)
    assert bsz == 16.0 and loader.batch_size == 16
    assert len(loader.dataset) == 7500 and len(loader) == 469        # last batch of 12
    sizes = [x.shape[0] for x, _ in loader]
    assert sizes[-1] == 12 and sum(sizes) == 7500
    x, y = next(iter(loader))
    assert x.shape == (16, 1, 28, 28) and y.dtype == torch.int64


def test_synthetic_dataset_deterministic():
    a = data.SyntheticImageDataset(100, (1, 28, 28), 10, seed=5)
    b = data.SyntheticImageDataset(100, (1, 28, 28), 10, seed=5)
    xa, ya = a[17]; xb, yb = b[17]
    assert torch.equal(xa, xb) and ya == yb
    t = data.SyntheticTokenDataset(10, 32, 100)
    x, y = t[3]
    assert x.shape == (32,) and torch.equal(x[1:], y[:-1])


def test_prefetch_loader_same_batches_early_break_and_errors():
    import threading
    from distributedpytorch_amd.data import BatchLoader, PrefetchLoader, SyntheticImageDataset
    ds = SyntheticImageDataset(50, (1, 8, 8), 10, seed=5)
    plain = BatchLoader(ds, 16, shuffle=True, seed=3)
    pre = PrefetchLoader(BatchLoader(ds, 16, shuffle=True, seed=3), None, depth=2)
    assert len(pre) == len(plain) == 4 and pre.batch_size == 16 and pre.dataset is ds
    for epoch in range(2):
        plain.set_epoch(epoch); pre.set_epoch(epoch)
        a, b = list(plain), list(pre)
        assert len(a) == len(b) == 4 and b[-1][0].shape[0] == 2          # ragged last batch kept
        for (xa, ya), (xb, yb) in zip(a, b):
            assert torch.equal(xa, xb) and torch.equal(ya, yb)
    multi = PrefetchLoader(BatchLoader(ds, 16, shuffle=True, seed=3), None, depth=2, workers=3)
    for epoch in range(2):
        plain.set_epoch(epoch); multi.set_epoch(epoch)
        for (xa, ya), (xb, yb) in zip(list(plain), list(multi)):         # several batches in flight, order kept
            assert torch.equal(xa, xb) and torch.equal(ya, yb)
    before = threading.active_count()
    for i, _ in enumerate(multi):
        if i == 0:
            break
    for i, _ in enumerate(pre):                                          # leaving early stops the producer
        if i == 1:
            break
    assert threading.active_count() <= before

    class Broken:
        def __len__(self):
            return 3

        def __iter__(self):
            yield torch.zeros(1), torch.zeros(1)
            raise RuntimeError("disk on fire")

    it = iter(PrefetchLoader(Broken()))
    next(it)
    with pytest.raises(RuntimeError, match="disk on fire"):
        next(it)


def _write_idx(path, arr, gz=False):
    """A tiny IDX file as the MNIST distribution lays it out: 0, 0, 0x08 (unsigned byte), rank; big-endian dims; bytes."""
    import gzip
    import struct
    head = bytes([0, 0, 0x08, arr.dim()]) + struct.pack(">" + "I" * arr.dim(), *arr.shape)
    opener = gzip.open if gz else open
    with opener(path, "wb") as f:
        f.write(head + arr.contiguous().numpy().tobytes())


@pytest.mark.parametrize("layout,gz", [("", False), ("MNIST/raw", True)])
def test_mnist_idx_reader_normalises_like_the_reference(tmp_path, layout, gz):
    """data.MNISTIdx on a generated fixture (the reference reads torchvision's MNIST with ToTensor + Normalize((0.1307,),
    (0.3081,)), multinode-DP/multinodeDP.py:100-107; no torchvision and no network here, so the files are the user's)."""
    import os
    from distributedpytorch_amd import data as D
    g = torch.Generator().manual_seed(5)
    imgs = torch.randint(0, 256, (37, 28, 28), generator=g, dtype=torch.uint8)
    labs = torch.randint(0, 10, (37,), generator=g, dtype=torch.uint8)
    root = tmp_path / layout
    os.makedirs(root, exist_ok=True)
    ext = ".gz" if gz else ""
    _write_idx(str(root / f"train-images-idx3-ubyte{ext}"), imgs, gz)
    _write_idx(str(root / f"train-labels-idx1-ubyte{ext}"), labs, gz)
    ds = D.MNISTIdx(str(tmp_path))
    assert len(ds) == 37 and ds.shape == (1, 28, 28)
    x, y = ds[5]
    want = (imgs[5].float() / 255.0 - 0.1307) / 0.3081
    assert x.shape == (1, 28, 28) and y == int(labs[5]) and torch.allclose(x[0], want, atol=1e-6)
    xb, yb = ds.batch([3, 36, 0])
    assert xb.shape == (3, 1, 28, 28) and torch.equal(yb, labs[[3, 36, 0]].long())
    assert torch.allclose(xb[1, 0], (imgs[36].float() / 255.0 - 0.1307) / 0.3081, atol=1e-6)
    # the reference's partitioning + loader on top of it (DP.py:108-114): ragged last batch kept
    loader, bsz = D.partition_dataset(ds, 0, 2, global_batch=8)
    got = list(loader)
    assert bsz == 4.0 and sum(b[0].shape[0] for b in got) == 18 and got[0][0].shape[1:] == (1, 28, 28)
    # with the files absent the same entry point falls back to synthetic data of the same shape, and says so
    said = []
    assert isinstance(D.mnist_or_synthetic(str(tmp_path / "nowhere"), n=32, log=said.append), D.SyntheticImageDataset)
    assert said and "synthetic" in said[0]
    assert isinstance(D.mnist_or_synthetic(str(tmp_path), log=said.append), D.MNISTIdx)


def test_mnist_idx_reader_refuses_what_is_not_idx(tmp_path):
    from distributedpytorch_amd import data as D
    (tmp_path / "train-images-idx3-ubyte").write_bytes(b"\x00\x00\x0d\x03" + b"\x00" * 64)     # float element type
    (tmp_path / "train-labels-idx1-ubyte").write_bytes(b"\x00\x00\x08\x01\x00\x00\x00\x02\x01\x02")
    with pytest.raises(ValueError, match="unsigned byte"):
        D.MNISTIdx(str(tmp_path))
    (tmp_path / "train-images-idx3-ubyte").write_bytes(b"\x00\x00\x08\x03" + b"\x00\x00\x00\x02" * 3 + b"\x01" * 5)
    with pytest.raises(ValueError, match="truncated"):
        D.MNISTIdx(str(tmp_path))
