"""The 256 x 256-tile GEMM main loop (csrc/gemm_wide.hip) against an fp32 reference AND against the 128 x 128
kernel it replaces on large grids (same operands, same epilogue: the two may differ by summation order only).

Shapes are chosen to hit: one K tile (K = 64: prologue and tail only), two, an odd count; ragged pixel and channel
edges (rows / columns past M / N must read zeros and never be written); more tiles than one group of 8 pixel
tiles; every fused epilogue the kernel implements.  Operands are random and asymmetric.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
BN = 256          # output channels per tile forced by _Wide: every test below runs once per tile width (fixture)


@pytest.fixture(autouse=True, params=[256, 192], ids=["tile256x256", "tile256x192"])
def _tile_width(request):
    """The 256 x 192 variant (template NT = 6) shares the LDS image, DMA plan and barrier structure of the 256 x 256 one;
    only the dead filter rows, the MFMA count of phases 3-4 and the store counts behind vmcnt differ - so it gets every
    test the wide tile gets."""
    global BN
    BN = request.param
    yield
    BN = 256


def _C():
    from distributedpytorch_amd.ops import _ext
    return _ext.load()


def _rand(shape, scale, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


class _Wide:
    """Route every eligible launch to the wide kernel (min items 1) or to the 128 x 128 kernel (0); ``splitk`` > 0
    forces that many K ranges per tile (0: the cost model decides)."""

    def __init__(self, on, splitk=1, bn=None):
        self.on, self.splitk, self.bn = on, splitk, (BN if bn is None else bn)

    def __enter__(self):
        C = _C()
        self.keep = C.wide_gemm_min_tiles()
        C.set_wide_gemm_min_tiles(1 if self.on else 0)
        C.set_wide_gemm_splitk(self.splitk)
        C.set_wide_gemm_bn(self.bn)

    def __exit__(self, *exc):
        C = _C()
        C.set_wide_gemm_min_tiles(self.keep)
        C.set_wide_gemm_splitk(0)
        C.set_wide_gemm_bn(0)
        return False


def _check(got, ref, narrow, what, mid=None):
    got, ref, narrow = got.float().cpu(), ref.float().cpu(), narrow.float().cpu()
    scale = float(ref.abs().max())
    err = (got - ref).abs()
    assert float(err.max()) <= 2e-2 * scale + 1e-3, f"{what}: max err {float(err.max()):.4g} vs fp32 reference (scale {scale:.4g})"
    # against the other kernel: one bf16 rounding of an fp32 sum taken in a different order
    d = (got - narrow).abs()
    tol = 8e-3 * narrow.abs() + 1e-4 * scale + 1e-6
    if mid is not None:     # an addend joins the bf16-ROUNDED product: one more rounding, at the product's magnitude
        tol = tol + 8e-3 * mid.float().cpu().abs()
    bad = int((d > tol).sum())
    assert bad == 0, f"{what}: {bad}/{d.numel()} elements differ from the 128x128 kernel beyond one rounding, max {float(d.max()):.4g}"


# the last two: more items than workgroups (persistent workgroups walk several tiles; the DMA runs through item seams)
SHAPES = [(256, 64, 256), (300, 128, 264), (1000, 768, 776), (513, 192, 520), (2304, 64, 512), (4096, 320, 256),
          (4100, 64, 5000), (6000, 192, 4104)]


@pytest.mark.parametrize("M,K,N", SHAPES)
def test_wide_gemm_forward_plain(M, K, N):
    C = _C()
    x = _rand((M, 1, 1, K), 1.0, 1)
    w = _rand((N, 1, 1, K), 0.1, 2)
    with _Wide(True):
        y = C.conv_fprop(x, w, None, 1, 0, 0, False, None)[0]
    with _Wide(False):
        y0 = C.conv_fprop(x, w, None, 1, 0, 0, False, None)[0]
    ref = x.float().view(M, K) @ w.float().view(N, K).t()
    _check(y.view(M, N), ref, y0.view(M, N), f"fprop {M}x{N}x{K}")


def test_wide_gemm_identity_weights_place_every_column_where_it_belongs():
    """A = I-style check with an asymmetric operand: w = identity-like selector, so y[m][n] = x[m][perm[n]] exactly -
    a transposed, shifted or swizzle-mismatched tile cannot pass."""
    C = _C()
    M, K, N = 520, 256, 512
    x = _rand((M, 1, 1, K), 1.0, 3)
    perm = torch.randperm(K, generator=torch.Generator().manual_seed(4))
    sel = torch.zeros(N, K)
    cols = torch.arange(N)
    sel[cols, perm[cols % K]] = 1.0
    w = sel.to(torch.bfloat16).to(DEV).view(N, 1, 1, K)
    with _Wide(True):
        y = C.conv_fprop(x, w, None, 1, 0, 0, False, None)[0].view(M, N)
    want = x.view(M, K)[:, perm[cols % K].to(DEV)]
    assert torch.equal(y, want)


@pytest.mark.parametrize("act", [0, 1, 2])
def test_wide_gemm_bias_activation_and_residual(act):
    C = _C()
    M, K, N = 700, 128, 520
    x = _rand((M, 1, 1, K), 1.0, 5)
    w = _rand((N, 1, 1, K), 0.1, 6)
    bias = torch.randn(N, device=DEV) * 0.5
    res = _rand((M, 1, 1, N), 1.0, 7)
    outs = []
    for on in (True, False):
        with _Wide(on):
            ya = C.conv_fprop(x, w, bias, 1, 0, act, False, None)[0]
            if act == 0:
                yr = C.conv_fprop(x, w, bias, 1, 0, 0, False, None, res)[0]
            else:
                yr = ya
        outs.append((ya, yr))
    lin = x.float().view(M, K) @ w.float().view(N, K).t() + bias
    ref = lin if act == 0 else torch.relu(lin) if act == 1 else torch.nn.functional.gelu(lin, approximate="tanh")
    (ya, yr), (ya0, yr0) = outs
    _check(ya.view(M, N), ref, ya0.view(M, N), f"bias act {act}")
    if act == 0:
        _check(yr.view(M, N), lin + res.float().view(M, N), yr0.view(M, N), "residual epilogue", mid=lin)


@pytest.mark.parametrize("M,K,N", [(600, 256, 264), (1024, 64, 768), (333, 1024, 256)])
def test_wide_gemm_input_gradient_with_addend_mask_and_gelu_backward(M, K, N):
    """mode 1 (dgrad of a Linear / 1x1 conv): dy [M, K] x wt [N, K]; plain, + masked addend, x gelu'(h)."""
    C = _C()
    dy = _rand((M, 1, 1, K), 1.0, 8)
    wt = _rand((N, 1, 1, K), 0.1, 9)
    add = _rand((M, 1, 1, N), 1.0, 10)
    mask = torch.randint(0, 256, (M * N // 8,), dtype=torch.uint8, generator=torch.Generator().manual_seed(11)).to(DEV)
    h = _rand((M, 1, 1, N), 1.5, 12)
    got = []
    for on in (True, False):
        with _Wide(on):
            p = C.conv_dgrad(dy, wt, 1, 1, 1, 0, None, None)
            a = C.conv_dgrad(dy, wt, 1, 1, 1, 0, add, None)
            am = C.conv_dgrad(dy, wt, 1, 1, 1, 0, add, mask)
            g = C.conv_dgrad(dy, wt, 1, 1, 1, 0, None, None, None, None, None, h)
        got.append((p, a, am, g))
    ref = dy.float().view(M, K) @ wt.float().view(N, K).t()
    bits = ((mask.view(-1, 1).cpu() >> torch.arange(8).view(1, 8)) & 1).view(M, N).float().to(DEV)
    hf = h.float().view(M, N).requires_grad_()
    torch.nn.functional.gelu(hf, approximate="tanh").sum().backward()
    wide, narrow = got
    _check(wide[0].view(M, N), ref, narrow[0].view(M, N), "dgrad")
    _check(wide[1].view(M, N), ref + add.float().view(M, N), narrow[1].view(M, N), "dgrad + addend", mid=ref)
    _check(wide[2].view(M, N), ref + add.float().view(M, N) * bits, narrow[2].view(M, N), "dgrad + masked addend", mid=ref)
    _check(wide[3].view(M, N), ref * hf.grad, narrow[3].view(M, N), "dgrad x gelu'", mid=ref)


@pytest.mark.parametrize("M,K,N,splitk", [(520, 1024, 264, 2), (300, 2048, 776, 5), (1024, 3072, 256, 8), (8192, 768, 520, 3)])
def test_wide_gemm_split_k_slabs_fold_to_the_same_result(M, K, N, splitk):
    """K ranges of a tile as separate work items, fp32 partial slabs, ordered fold + epilogue: against the reference,
    against the 128 x 128 kernel, and bit-for-bit reproducible from launch to launch (no atomics)."""
    C = _C()
    dy = _rand((M, 1, 1, K), 1.0, 21)
    wt = _rand((N, 1, 1, K), 0.05, 22)
    add = _rand((M, 1, 1, N), 1.0, 23)
    bias = torch.randn(N, device=DEV) * 0.5
    with _Wide(True, splitk):
        p = C.conv_dgrad(dy, wt, 1, 1, 1, 0, None, None)
        p2 = C.conv_dgrad(dy, wt, 1, 1, 1, 0, None, None)
        a = C.conv_dgrad(dy, wt, 1, 1, 1, 0, add, None)
        f = C.conv_fprop(dy, wt, bias, 1, 0, 1, False, None)[0]
    with _Wide(False):
        p0 = C.conv_dgrad(dy, wt, 1, 1, 1, 0, None, None)
        a0 = C.conv_dgrad(dy, wt, 1, 1, 1, 0, add, None)
        f0 = C.conv_fprop(dy, wt, bias, 1, 0, 1, False, None)[0]
    ref = dy.float().view(M, K) @ wt.float().view(N, K).t()
    assert torch.equal(p, p2), "split-K must be deterministic"
    _check(p.view(M, N), ref, p0.view(M, N), f"split-K {splitk} dgrad")
    _check(a.view(M, N), ref + add.float().view(M, N), a0.view(M, N), f"split-K {splitk} dgrad + addend", mid=ref)
    _check(f.view(M, N), torch.relu(ref + bias), f0.view(M, N), f"split-K {splitk} fprop + bias + relu")


def test_wide_gemm_is_what_large_launches_use_by_default_and_small_ones_do_not():
    C = _C()
    assert C.wide_gemm_min_tiles() > 0, "the wide kernel must be on by default (DPA_WIDE_MIN_TILES=0 disables it)"


def test_split_k_workspace_comes_from_torchs_caching_allocator():
    """The split-K slabs (fp32, split x M x N) are taken from torch's caching allocator for the launch's stream and handed
    back behind the fold kernel: visible to torch.cuda.memory_*, not a private hipMalloc that grows and is never freed."""
    C = _C()
    M, K, N, S = 4096, 1024, 512, 4
    x = _rand((M, 1, 1, K), 1.0, 21)
    w = _rand((N, 1, 1, K), 0.1, 22)
    with _Wide(True, splitk=S):
        C.conv_fprop(x, w, None, 1, 0, 0, False, None)                     # warm: code object, allocator pools
        torch.cuda.synchronize()
        torch.cuda.reset_peak_memory_stats()
        base = torch.cuda.memory_allocated()
        y = C.conv_fprop(x, w, None, 1, 0, 0, False, None)[0]
        torch.cuda.synchronize()
        peak = torch.cuda.max_memory_allocated() - base
    slabs = S * M * N * 4
    assert peak >= slabs, f"peak {peak} bytes does not include the {slabs}-byte split-K workspace"
    assert torch.cuda.memory_allocated() - base <= y.numel() * 2 + (1 << 20)    # ... and it was handed back
    ref = x.float().view(M, K) @ w.float().view(N, K).t()
    assert float((y.view(M, N).float() - ref).abs().max()) <= 2e-2 * float(ref.abs().max()) + 1e-3


@pytest.mark.parametrize("M,K,N", [(300, 128, 264), (4100, 64, 5000), (16384, 768, 3072)])
def test_wide_gemm_forward_gelu_twin(M, K, N):
    """fprop with the second output gelu_tanh(y) written by the same epilogue (GPT-2's c_fc: y stays for the GELU's backward,
    the twin feeds c_proj): both outputs against the 128 x 128 kernel's and against fp32; ragged edges never written."""
    C = _C()
    x = _rand((M, 1, 1, K), 1.0, 11)
    w = _rand((N, 1, 1, K), 0.1, 12)
    b = (torch.randn(N, generator=torch.Generator().manual_seed(13)) * 0.5).to(DEV)
    with _Wide(True):
        y, t = C.conv_fprop(x, w, b, 1, 0, 0, False, None, None, True)
    with _Wide(False):
        y0, t0 = C.conv_fprop(x, w, b, 1, 0, 0, False, None, None, True)
    ref = x.float().view(M, K) @ w.float().view(N, K).t() + b
    _check(y.view(M, N), ref, y0.view(M, N), f"twin fprop y {M}x{N}x{K}")
    # the twin is a function of the ROUNDED y: compare against gelu of this kernel's own y
    want = torch.nn.functional.gelu(y.float(), approximate="tanh")
    err = (t.float() - want).abs().max().item()
    assert err <= 8e-3 * float(want.abs().max()) + 1e-3, err
    _check(t.view(M, N), torch.nn.functional.gelu(ref, approximate="tanh"), t0.view(M, N), f"twin fprop gelu(y) {M}x{N}x{K}")
