"""csrc/conv3x3_halo.hip (halo-tiled 3 x 3 / stride 1 / pad 1, 64 -> 64 channels, 56-wide maps: ResNet stage 1) against an
fp32 ``F.conv2d`` reference AND against the implicit-GEMM kernel it replaces there (same operands, same epilogues: the two
may differ by summation order only).

What a halo kernel can get wrong and a random interior cannot show: image borders (zero rows above / below the image come
from the buffer descriptor, zero columns left / right from LDS slots that are never written), the seam between two
images inside one workgroup's run of tiles, the junk columns of the slot space leaking into stored pixels or into the
statistics, mirrored taps of the input gradient, a tile count that does not divide by the number of workgroups.  The
operands are random and asymmetric; one test plants single impulses at the image corners and reads the nine taps back."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _C():
    from distributedpytorch_amd.ops import _ext
    return _ext.load()


class _Halo:
    def __init__(self, on):
        self.on = on

    def __enter__(self):
        C = _C()
        self.keep = C.conv3x3_halo_enabled()
        C.set_conv3x3_halo(self.on)

    def __exit__(self, *exc):
        _C().set_conv3x3_halo(self.keep)
        return False


def _rand(shape, scale, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


def _ref_fprop(x, w):
    """fp32 reference: x NHWC bf16, w KRSC bf16 -> NHWC fp32."""
    y = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), padding=1)
    return y.permute(0, 2, 3, 1).contiguous()


def _close(got, ref, what, tol=2e-2):
    got, ref = got.float(), ref.float()
    scale = float(ref.abs().max())
    err = float((got - ref).abs().max())
    assert err <= tol * scale + 1e-3, f"{what}: max err {err:.4g} (scale {scale:.4g})"


@pytest.mark.parametrize("N", [1, 3, 37, 300])      # 14 / 42 / 518 / 4,200 tiles on 256 workgroups: ragged runs, image seams inside a run
def test_halo_forward_matches_fp32_reference_and_the_implicit_gemm(N):
    C = _C()
    x = _rand((N, 56, 56, 64), 1.0, 1)
    w = _rand((64, 3, 3, 64), 0.05, 2)
    with _Halo(True):
        sb = torch.zeros(32, 2, 64, device=DEV)
        y, st = C.conv_fprop(x, w, None, 1, 1, 0, True, sb)
    with _Halo(False):
        sb0 = torch.zeros(32, 2, 64, device=DEV)
        y0, st0 = C.conv_fprop(x, w, None, 1, 1, 0, True, sb0)
    ref = _ref_fprop(x, w)
    _close(y, ref, "halo fprop vs fp32")
    d = (y.float() - y0.float()).abs()
    assert float(d.max()) <= 8e-3 * float(y0.float().abs().max()) + 1e-3, "halo fprop vs implicit GEMM beyond one rounding"
    # BatchNorm statistics of the bf16 output: per-channel sum and sum of squares, folded over the replica rows
    s = st.sum(0).cpu()
    yf = y.float().view(-1, 64)
    want = torch.stack([yf.sum(0), (yf * yf).sum(0)]).cpu()
    assert torch.allclose(s, want, rtol=2e-3, atol=1e-2 * float(want.abs().max())), (s[:, :4], want[:, :4])
    s0 = st0.sum(0).cpu()
    assert torch.allclose(s, s0, rtol=5e-3, atol=2e-2 * float(s0.abs().max()))


def test_halo_forward_impulses_at_the_borders_read_back_the_taps():
    """x = one unit impulse per probed pixel (channel c), w[k][r][s][c] = a distinct small integer per (k, r, s): the output
    around the impulse IS the mirrored filter, exactly (bf16-exact integers) - a transposed / shifted / wrongly keyed tile,
    or a pad slot that is not zero, cannot pass.  Probes: the four corners, an edge, the tile seam rows 3 | 4, the last row."""
    C = _C()
    N, c = 2, 5
    x = torch.zeros(N, 56, 56, 64, device=DEV, dtype=torch.bfloat16)
    probes = [(0, 0, 0), (0, 0, 55), (0, 55, 0), (1, 55, 55), (0, 3, 17), (0, 4, 40), (1, 30, 0), (1, 20, 55)]
    for n, h, wq in probes:
        x[n, h, wq, c] = 1.0
    w = torch.zeros(64, 3, 3, 64, device=DEV)
    k = torch.arange(64, device=DEV).view(64, 1, 1)
    w[:, :, :, c] = (k % 7 + 1).float() * 8 + torch.arange(9, device=DEV).view(1, 3, 3).float() - 4.0      # |value| <= 60: exact in bf16
    w = w.to(torch.bfloat16)
    with _Halo(True):
        y = C.conv_fprop(x, w, None, 1, 1, 0, False, None)[0]
    ref = _ref_fprop(x, w)
    assert torch.equal(y.float(), ref), f"{int((y.float() != ref).sum())} outputs differ from the exact reference"


@pytest.mark.parametrize("N,with_stats", [(2, False), (37, True), (300, True)])
def test_halo_input_gradient_matches_reference_and_the_implicit_gemm(N, with_stats):
    """conv_dgrad through the halo kernel (mirrored taps over the K/C-swapped twin), alone and with the fused reduction pass
    of the BatchNorm in front of the gradient's consumer (epi 4: sum g', sum g' * c under the ReLU bits)."""
    C = _C()
    dy = _rand((N, 56, 56, 64), 1.0, 11)
    w = _rand((64, 3, 3, 64), 0.05, 12)                      # forward filter KRSC
    wt = C.transpose_filter(w)                               # [C][R][S][K] twin
    bn_x = _rand((N, 56, 56, 64), 1.0, 13) if with_stats else None
    mask = None
    if with_stats:
        g = torch.Generator(device="cpu").manual_seed(14)
        mask = torch.randint(0, 256, (N * 56 * 56 * 64 // 8,), generator=g, dtype=torch.uint8).to(DEV)
    outs = []
    for on in (True, False):
        with _Halo(on):
            sb = torch.zeros(32, 2, 64, device=DEV) if with_stats else None
            dx = C.conv_dgrad(dy, wt, 56, 56, 1, 1, None, None, bn_x, mask, sb)
            outs.append((dx, None if sb is None else sb.sum(0).cpu()))
    # fp32 reference of the input gradient: correlation of dy with the flipped, channel-swapped filter
    ref = torch.nn.grad.conv2d_input((N, 64, 56, 56), w.float().permute(0, 3, 1, 2), dy.float().permute(0, 3, 1, 2), padding=1)
    ref = ref.permute(0, 2, 3, 1).contiguous()
    _close(outs[0][0], ref, "halo dgrad vs fp32")
    d = (outs[0][0].float() - outs[1][0].float()).abs()
    assert float(d.max()) <= 8e-3 * float(outs[1][0].float().abs().max()) + 1e-3
    if with_stats:
        gf = outs[0][0].float().view(-1, 64)
        bits = ((mask.view(-1, 1).int() >> torch.arange(8, device=DEV).view(1, 8)) & 1).view(-1, 64).float()
        gm = gf * bits
        want = torch.stack([gm.sum(0), (gm * bn_x.float().view(-1, 64)).sum(0)]).cpu()
        assert torch.allclose(outs[0][1], want, rtol=2e-3, atol=1e-2 * float(want.abs().max()))
        assert torch.allclose(outs[0][1], outs[1][1], rtol=5e-3, atol=2e-2 * float(outs[1][1].abs().max()))


def test_halo_kernel_is_the_one_that_runs_and_falls_back_where_it_does_not_apply():
    """Dispatch: 64 -> 64 / 56-wide / stride 1 goes to the halo kernel; other widths, channel counts and strides stay on the
    implicit GEMM (same numbers either way); the deterministic mode stays on the implicit GEMM."""
    C = _C()
    for shape, wshape, stride in [((4, 28, 28, 64), (64, 3, 3, 64), 1), ((4, 56, 56, 128), (128, 3, 3, 128), 1),
                                  ((4, 56, 56, 64), (64, 3, 3, 64), 2), ((4, 54, 56, 64), (64, 3, 3, 64), 1)]:
        x, w = _rand(shape, 1.0, 3), _rand(wshape, 0.05, 4)
        with _Halo(True):
            y = C.conv_fprop(x, w, None, stride, 1, 0, False, None)[0]
        ref = F.conv2d(x.float().permute(0, 3, 1, 2), w.float().permute(0, 3, 1, 2), stride=stride, padding=1).permute(0, 2, 3, 1)
        _close(y, ref, f"fallback {shape} stride {stride}")
