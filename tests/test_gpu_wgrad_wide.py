"""The wide-tile, several-layers-per-launch weight gradient (csrc/wgrad_wide.hip) against an fp32 reference.

dw += dy^T x with both operands token-major: the fragments come from transposing LDS reads of an XOR-swizzled image
filled by LDS-DMA, so the tests that matter are PLACEMENT tests (a selector operand makes every output element a copy of
one input element: a wrong swizzle, a transposed tile or a shifted token row cannot pass), ragged edges in all three
dimensions, the token split with its ordered fold (bit-reproducible), accumulation into a non-zero dw, the bias column
sums, and a launch with several layers whose items interleave on the workgroups.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _C():
    from distributedpytorch_amd.ops import _ext
    return _ext.load()


def _rand(shape, scale, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


class _Split:
    def __init__(self, n):
        self.n = n

    def __enter__(self):
        _C().set_wgrad_wide_splitk(self.n)

    def __exit__(self, *exc):
        _C().set_wgrad_wide_splitk(0)
        return False


def _run(layers, split, with_bias=True, seed=0):
    """layers: [(M, N, K)] with one M.  Returns per layer (dw_got, dw_ref, db_got, db_ref)."""
    C = _C()
    xs, dys, dws, dbs, refs = [], [], [], [], []
    for q, (M, N, K) in enumerate(layers):
        x = _rand((M, K), 1.0, seed + 10 * q + 1)
        dy = _rand((M, N), 0.5, seed + 10 * q + 2)
        dw0 = torch.randn(N, K, generator=torch.Generator().manual_seed(seed + 10 * q + 3)).to(DEV)
        db0 = torch.randn(N, generator=torch.Generator().manual_seed(seed + 10 * q + 4)).to(DEV)
        xs.append(x); dys.append(dy); dws.append(dw0.clone()); dbs.append(db0.clone() if with_bias else None)
        refs.append((dw0.double() + dy.double().t() @ x.double(), db0.double() + dy.double().sum(0)))
    with _Split(split):
        C.wgrad_wide_group(xs, dys, dws, dbs)
    torch.cuda.synchronize()
    return [(dws[q], refs[q][0], dbs[q], refs[q][1]) for q in range(len(layers))]


def _check(got, ref, what, rel=2e-5):
    err = (got.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    assert err <= rel * scale + 1e-4, f"{what}: max err {err:.4g} (scale {scale:.4g})"


@pytest.mark.parametrize("split", [1, 2, 3])
@pytest.mark.parametrize("M,N,K", [(64, 256, 256), (1024, 256, 256), (1000, 264, 520), (4096, 768, 512), (2050, 200, 1032)])
def test_one_layer_against_fp64(M, N, K, split):
    if split > 1 and (M + 63) // 64 < 8 * split:
        pytest.skip("too few K steps for this split")
    (dw, rw, db, rb), = _run([(M, N, K)], split)
    _check(dw, rw, f"dw {M}x{N}x{K} split {split}")
    _check(db, rb, f"dbias {M}x{N}x{K} split {split}", rel=1e-4)


@pytest.mark.parametrize("split", [1, 2])
def test_selector_operand_places_every_element(split):
    """dy = one-hot rows: dw[n][:] must be EXACTLY the x row of the one token that selects n (plus what dw held)."""
    C = _C()
    M, N, K = 1024, 512, 520
    x = _rand((M, K), 1.0, 5)
    perm = torch.randperm(M, generator=torch.Generator().manual_seed(6))[:N]     # token perm[n] selects output row n
    dy = torch.zeros(M, N)
    dy[perm, torch.arange(N)] = 1.0
    dy = dy.to(torch.bfloat16).to(DEV)
    dw = torch.zeros(N, K, device=DEV)
    with _Split(split):
        C.wgrad_wide_group([x], [dy], [dw], [None])
    want = x.float()[perm.to(DEV)]
    assert torch.equal(dw, want)


def test_transformer_block_group_with_biases_and_the_fold_is_deterministic():
    layers = [(2048, 768, 3072), (2048, 3072, 768), (2048, 768, 768), (2048, 2304, 768)]      # fc2, fc, proj, qkv of GPT-2
    a = _run(layers, 0)
    b = _run(layers, 0)
    for q, ((dw, rw, db, rb), (dw2, _, db2, _)) in enumerate(zip(a, b)):
        _check(dw, rw, f"layer {q} dw")
        _check(db, rb, f"layer {q} dbias", rel=1e-4)
        assert torch.equal(dw, dw2), f"layer {q}: the ordered fold must reproduce bit for bit"


def test_more_items_than_workgroups_walks_item_seams():
    """600 tiles on 256 workgroups: persistent workgroups flush a tile and carry the DMA cursors into the next item."""
    (dw, rw, db, rb), = _run([(512, 6144, 6400)], 1, seed=40)
    _check(dw, rw, "dw 512x6144x6400")
    _check(db, rb, "dbias", rel=1e-4)
    (dw, rw, db, rb), = _run([(1536, 5120, 4104)], 2, seed=50)
    _check(dw, rw, "dw 1536x5120x4104 split 2")


def test_ragged_last_round_runs_whole_rounds_unsplit_and_splits_the_rest():
    """600 tiles with the split left to the launcher: 512 tiles in one unsplit launch (each tile's one writer adds it into
    dw), the other 88 in a second launch split over the tokens (slab + ordered fold); every tile - and every bias sum -
    exactly once.  Two layers, so that the cut between the launches falls INSIDE a layer."""
    layers = [(2048, 2560, 2560), (2048, 5120, 5120)]        # 100 + 400 tiles ... + a third below
    res = _run(layers + [(2048, 2560, 2560)], 0, seed=70)     # 600 tiles: the cut is inside the third layer
    for q, (dw, rw, db, rb) in enumerate(res):
        _check(dw, rw, f"hybrid layer {q} dw")
        _check(db, rb, f"hybrid layer {q} dbias", rel=1e-4)


def test_without_bias_and_eligibility_query():
    C = _C()
    (dw, rw, db, rb), = _run([(2048, 512, 256)], 0, with_bias=False)
    _check(dw, rw, "dw no bias")
    ok, tiles = C.wgrad_wide_query(16384, 2304, 768)
    assert ok and tiles == 27
    ok, _ = C.wgrad_wide_query(16384, 64, 256)           # a ResNet-sized filter stays on conv_wgrad.hip
    assert not ok


def test_grouped_weight_gradients_through_the_dp_engine_match_per_layer_launches():
    """ops/wgroup.py in its real place: a GPT-2 with 256-wide blocks over 2,048 tokens through DataParallel with small
    buckets (bucket launches must flush held-back groups and see every gradient), with a second backward accumulated
    under no_sync.  Against the same model with the batching off (each layer through conv_wgrad): the gradients may
    differ by summation order only.  Also: the batcher really launched groups, and left nothing behind."""
    from distributedpytorch_amd.models import gpt2
    from distributedpytorch_amd.ops import wgroup
    from distributedpytorch_amd.parallel.dp import DataParallel

    def grads(batching):
        keep = wgroup.ENABLED
        wgroup.ENABLED = batching
        try:
            torch.manual_seed(11)
            m = gpt2.gpt2_tiny(n_layer=3, n_embd=256, n_head=4, vocab=504, n_positions=512).to(DEV).train()
            g = torch.Generator(device=DEV).manual_seed(5)
            x = torch.randint(0, 504, (4, 512), generator=g, device=DEV)
            y = torch.randint(0, 504, (4, 512), generator=g, device=DEV)
            dp = DataParallel(m, bucket_mb=0.5, first_bucket_mb=0.1)
            assert len(dp.buckets) >= 3
            b = wgroup.for_device(torch.cuda.current_device())
            before = (b.launches, b.problems)
            with dp.no_sync():
                m.loss(x, y).backward()
            m.loss(x, y).backward()
            dp.finish()
            torch.cuda.synchronize()
            assert not b.items, "a held-back weight gradient survived finish()"
            return dp.flat.grad.clone(), (b.launches - before[0], b.problems - before[1])
        finally:
            wgroup.ENABLED = keep

    g1, (launches, problems) = grads(True)
    g0, (l0, p0) = grads(False)
    assert l0 == 0 and p0 == 0
    # two backwards x (three blocks x four Linear layers + the LM head), several layers per launch
    assert problems == 2 * (3 * 4 + 1) and 2 <= launches < problems // 2, (launches, problems)
    scale = float(g0.abs().max())
    assert float((g1 - g0).abs().max()) <= 2e-3 * scale, (float((g1 - g0).abs().max()), scale)


def test_resnet_late_stage_1x1_weight_gradients_go_through_the_batcher():
    """ops/block.py hands the 1 x 1 layers of the late stages (filters >= 192 x 192, >= 2,048 pixels) to the batcher: with
    ResNet-50 at batch 16 that is stage 3 (3,136 pixels; stage 4's 7 x 7 maps have 784 and stay).  Same gradients as the per-layer
    launches of conv_wgrad up to summation order, in aggregate (BatchNorm amplifies the atomics' last-bit noise)."""
    from distributedpytorch_amd.models import resnet
    from distributedpytorch_amd.ops import block, wgroup
    from distributedpytorch_amd.parallel.dp import DataParallel

    def grads(wide):
        keep = block._WIDE_1X1
        block._WIDE_1X1 = wide
        try:
            torch.manual_seed(21)
            m = resnet.resnet50().to(DEV).train()
            g = torch.Generator(device=DEV).manual_seed(9)
            x = torch.randn(16, 3, 224, 224, generator=g, device=DEV)
            y = torch.randint(0, 1000, (16,), generator=g, device=DEV)
            dp = DataParallel(m)
            b = wgroup.for_device(torch.cuda.current_device())
            before = b.problems
            m.loss(x, y).backward()
            dp.finish()
            torch.cuda.synchronize()
            assert not b.items
            return dp.flat.grad.clone(), b.problems - before
        finally:
            block._WIDE_1X1 = keep

    g1, n1 = grads(True)
    g0, n0 = grads(False)
    # stage 3: six blocks x (conv1, conv3), and conv1 of stage 4's first block (it still sees the 14 x 14 map)
    assert n0 == 0 and n1 == 13, (n0, n1)
    rel = float((g1 - g0).norm() / g0.norm())
    assert rel < 2e-2, rel
