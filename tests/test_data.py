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
