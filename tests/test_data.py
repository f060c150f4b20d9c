"""T0: Partition / DataPartitioner semantics (reference DP.py:42-75)."""
from random import Random

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
