"""csrc/tail_bwd.hip - BatchNorm-backward apply + 1x1 conv wgrad + dgrad (+ the reduction pass of the BatchNorm in
front) in one kernel - against (a) a plain fp32 PyTorch evaluation of the same formulas and (b) the three-kernel
sequence it replaces (bn_bwd_apply -> conv_wgrad + conv_dgrad with the statistics epilogue).

Shapes: fewer rows than one tile, a ragged last tile, several tiles per persistent workgroup (P > 512 x 64 rows).
Operands are random and asymmetric; the filter is NOT symmetric, so a transposed image cannot pass.
"""
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
SHAPES = [(64, 256), (128, 512)]        # stage 1 and stage 2 of a ResNet: two kernels behind one entry point


def _C():
    from distributedpytorch_amd.ops import _ext
    return _ext.load()


def _bf(shape, scale, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).to(DEV)


def _bits(mask, shape):
    b = ((mask.view(-1, 1).cpu() >> torch.arange(8).view(1, 8)) & 1).reshape(shape).float()
    return b.to(DEV)


def _case(N, H, W, seed, CM, CO, with_stats=True, with_mask=True):
    C = _C()
    P = N * H * W
    dout = _bf((N, H, W, CO), 1.0, seed)
    craw = _bf((N, H, W, CO), 1.5, seed + 1)
    xin = _bf((N, H, W, CM), 1.0, seed + 2)
    w = _bf((CO, 1, 1, CM), 0.1, seed + 3)              # [K=CO][1][1][C=CM]
    wt = C.transpose_filter(w)                          # [CM][1][1][CO]
    g = torch.Generator(device="cpu").manual_seed(seed + 4)
    kout = torch.randint(0, 256, (P * CO // 8,), dtype=torch.uint8, generator=g).to(DEV) if with_mask else None
    co = torch.zeros(5, CO, device=DEV)
    co[2] = torch.rand(CO, generator=g).to(DEV) + 0.5
    co[3] = (torch.rand(CO, generator=g).to(DEV) - 0.5) * 0.2
    co[4] = (torch.rand(CO, generator=g).to(DEV) - 0.5) * 0.1
    sx = _bf((N, H, W, CM), 1.0, seed + 5)
    smask = torch.randint(0, 256, (P * CM // 8,), dtype=torch.uint8, generator=g).to(DEV)
    return dict(P=P, CM=CM, CO=CO, dout=dout, craw=craw, xin=xin, w=w, wt=wt, kout=kout, co=co, sx=sx, smask=smask)


def _reference(c):
    P, CM, CO = c["P"], c["CM"], c["CO"]
    d = c["dout"].float().view(P, CO)
    if c["kout"] is not None:
        d = d * _bits(c["kout"], (P, CO))
    gfull = c["co"][2] * d + c["co"][3] * c["craw"].float().view(P, CO) + c["co"][4]
    gb = gfull.to(torch.bfloat16).float()               # what the MFMAs see
    dw = gb.t() @ c["xin"].float().view(P, CM)          # [CO][CM]
    dx = gb @ c["w"].float().view(CO, CM)               # [P][CM]
    dxb = dx.to(torch.bfloat16).float() * _bits(c["smask"], (P, CM))
    s = dxb.sum(0)
    q = (dxb * c["sx"].float().view(P, CM)).sum(0)
    return gb, dw, dx, s, q


@pytest.mark.parametrize("CM,CO", SHAPES)
@pytest.mark.parametrize("N,H,W", [(1, 5, 7), (2, 14, 14), (3, 25, 27), (8, 112, 112)])
def test_tail_bwd_matches_fp32_reference_and_the_three_kernel_sequence(N, H, W, CM, CO):
    C = _C()
    if CM == 128 and N == 8:
        H = W = 56                                   # 25,088 pixels = 1,568 tiles of 16: several tiles per workgroup
    c = _case(N, H, W, 100 + N, CM, CO)
    P = c["P"]
    assert C.tail_bwd_ok(P, CM, CO)
    rows = 32
    dw = torch.zeros(CO, 1, 1, CM, device=DEV)
    stats = torch.zeros(rows, 2, CM, device=DEV)
    dx = C.tail_bwd(c["dout"], c["kout"], c["craw"], c["co"], c["xin"], c["wt"], dw, c["sx"], c["smask"], stats)
    torch.cuda.synchronize()
    gb, dw_ref, dx_ref, s_ref, q_ref = _reference(c)

    def close(got, ref, what, rel=2e-2):
        scale = float(ref.abs().max())
        err = float((got.float() - ref).abs().max())
        assert err <= rel * scale + 1e-3, f"{what}: max err {err:.4g} at scale {scale:.4g}"

    close(dx.view(P, CM), dx_ref, "dx")
    close(dw.view(CO, CM), dw_ref, "dw", rel=5e-3)
    st = stats.sum(0)
    close(st[0], s_ref, "sum dx'", rel=5e-3)
    close(st[1], q_ref, "sum dx' * sx", rel=5e-3)

    # the sequence it replaces
    g3 = C.bn_bwd_apply(c["dout"], c["kout"], c["craw"], c["co"], True)
    dw0 = torch.zeros(CO, 1, 1, CM, device=DEV)
    C.conv_wgrad(c["xin"], g3, dw0, 1, 0)
    stats0 = torch.zeros(rows, 2, CM, device=DEV)
    dx0 = C.conv_dgrad(g3, c["wt"], H, W, 1, 0, None, None, c["sx"], c["smask"], stats0)
    torch.cuda.synchronize()
    dg = (g3.float().view(P, CO) - gb).abs()             # fused multiply-adds vs torch's separate roundings: <= one bf16 ulp
    assert int((dg > 8e-3 * gb.abs() + 1e-6).sum()) == 0, "the kernel-level oracle and bn_bwd_apply disagree on g"
    d = (dx.float() - dx0.float()).abs().view(P, CM)
    tol = 8e-3 * dx0.float().abs().view(P, CM) + 1e-4 * float(dx_ref.abs().max()) + 1e-6
    assert int((d > tol).sum()) == 0, "dx differs from conv_dgrad's beyond one bf16 rounding"
    close(dw.view(CO, CM), dw0.view(CO, CM).float(), "dw vs conv_wgrad", rel=2e-3)
    st0 = stats0.sum(0)
    close(st[0], st0[0], "stats[0] vs dgrad epilogue", rel=5e-3)
    close(st[1], st0[1], "stats[1] vs dgrad epilogue", rel=5e-3)


@pytest.mark.parametrize("CM,CO", SHAPES)
def test_tail_bwd_without_statistics_and_without_mask_accumulates_into_dw(CM, CO):
    C = _C()
    c = _case(2, 9, 11, 7, CM, CO, with_mask=False)
    P = c["P"]
    dw = torch.full((CO, 1, 1, CM), 3.0, device=DEV)                    # accumulated into, not overwritten
    dx = C.tail_bwd(c["dout"], None, c["craw"], c["co"], c["xin"], c["wt"], dw)
    torch.cuda.synchronize()
    _, dw_ref, dx_ref, _, _ = _reference(c)
    assert float((dw.view(CO, CM) - 3.0 - dw_ref).abs().max()) <= 5e-3 * float(dw_ref.abs().max()) + 1e-3
    assert float((dx.float().view(P, CM) - dx_ref).abs().max()) <= 2e-2 * float(dx_ref.abs().max()) + 1e-3


@pytest.mark.parametrize("CM,CO", SHAPES)
def test_tail_bwd_identity_filter_returns_g_itself(CM, CO):
    """w = selector (dx[:, cm] = g[:, perm[cm]]): exact, so any row / column / swizzle mismatch shows."""
    C = _C()
    c = _case(2, 16, 16, 11, CM, CO)
    P = c["P"]
    perm = torch.randperm(CO, generator=torch.Generator().manual_seed(5))[:CM]
    w = torch.zeros(CO, CM)
    w[perm, torch.arange(CM)] = 1.0
    w = w.to(torch.bfloat16).to(DEV).view(CO, 1, 1, CM)
    wt = C.transpose_filter(w)
    dw = torch.zeros(CO, 1, 1, CM, device=DEV)
    dx = C.tail_bwd(c["dout"], c["kout"], c["craw"], c["co"], c["xin"], wt, dw)
    g3 = C.bn_bwd_apply(c["dout"], c["kout"], c["craw"], c["co"], True)
    torch.cuda.synchronize()
    assert torch.equal(dx.view(P, CM), g3.view(P, CO)[:, perm.to(DEV)])
