"""GPU: the weights-stationary direct 64->64 3x3 convolution (csrc/xr_conv64.hip) against the fp32 CPU convolution and
against the implicit-GEMM kernel it replaces, including its fusions (normalise + PReLU on load, per-image output
statistics, residual-gradient sum, bias) and ragged images (edge tiles)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detgen as G

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(key, *shape, scale=1.0):
    n = int(np.prod(shape))
    return torch.from_numpy((G.normal(key, n) * scale).reshape(shape).astype(np.float32))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


CASES = [  # N, H, W
    (2, 32, 32),     # 2 x 2 full tiles per image
    (3, 20, 28),     # ragged: edge tiles in both directions
    (1, 112, 112),   # the FSRNet layer
    (5, 7, 9),       # image smaller than one tile
    (300, 16, 16),   # more tiles than workgroups hold in one pass is not needed: 300 tiles over <= 256 workgroups
]


def _nhwc(x):
    return x.permute(0, 2, 3, 1).contiguous()


@pytest.mark.parametrize("wide", [0])
@pytest.mark.parametrize("case", CASES)
def test_direct_conv64_forward_and_input_gradient(case, wide):
    from xrface import ops
    from xrface._lib import lib, ptr, stream
    N, H, W = case
    lib.xr_tune(13, wide)
    try:
        x = rnd(f"d64x{case}", N, 64, H, W).bfloat16().float()
        w = rnd(f"d64w{case}", 64, 64, 3, 3, scale=(64 * 9) ** -0.5)
        b = rnd(f"d64b{case}", 64, scale=0.1)
        gy = rnd(f"d64g{case}", N, 64, H, W).bfloat16().float()
        y_ref = F.conv2d(x, w, b, 1, 1)
        dx_ref = F.conv_transpose2d(gy, w, None, 1, 1)          # input gradient of the (bias-free) convolution
        xb = _nhwc(x).to(DEV).bfloat16()
        gb = _nhwc(gy).to(DEV).bfloat16()
        wd = w.to(DEV)
        pk, _ = ops._packed(wd, "fwd", torch.bfloat16, 64, 1, 9, 64, 64, 64 * 9, 0, 1, 9)
        pkd, _ = ops._packed(wd, "dgrad", torch.bfloat16, 64, 1, 9, 64, 64, 9, 0, 1, 64 * 9)
        y = torch.empty_like(xb)
        dx = torch.empty_like(xb)
        bias = b.to(DEV)
        lib.xr_conv64_direct(ptr(xb), ptr(pk), ptr(bias), ptr(y), N, H, W, 0, None, None, None, None, None, stream())
        lib.xr_conv64_direct(ptr(gb), ptr(pkd), None, ptr(dx), N, H, W, 1, None, None, None, None, None, stream())
        torch.cuda.synchronize()
        assert rel(y.float().permute(0, 3, 1, 2), y_ref) < 1e-2
        assert rel(dx.float().permute(0, 3, 1, 2), dx_ref) < 1e-2
        # against the implicit-GEMM kernel on the same packs: same products, fp32 accumulation in another order, one bf16 rounding
        y2 = torch.empty_like(xb)
        dx2 = torch.empty_like(xb)
        lib.xr_conv_igemm(0, ptr(xb), ptr(pk), ptr(bias), ptr(y2), N, H, W, 64, H, W, 64, 3, 3, 1, 1, 0, 576, 64, None, 0, None, None, None,
                          1, None, None, None, stream())
        lib.xr_conv_igemm(0, ptr(gb), ptr(pkd), None, ptr(dx2), N, H, W, 64, H, W, 64, 3, 3, 1, 1, 1, 576, 64, None, 0, None, None, None,
                          1, None, None, None, stream())
        torch.cuda.synchronize()
        assert rel(y.float(), y2.float()) < 8e-3 and rel(dx.float(), dx2.float()) < 8e-3
        assert float((y.float() - y2.float()).abs().mean()) < 2e-3 * float(y2.float().abs().mean())
    finally:
        lib.xr_tune(13, 0)


@pytest.mark.parametrize("transposed", [0, 1])
@pytest.mark.parametrize("case", [(2, 32, 32), (3, 20, 28), (2, 112, 112)])
def test_direct_conv64_fusions(case, transposed):
    """Normalise + PReLU on load (zero padding applies AFTER the transform), per-image output statistics, ep_add."""
    from xrface import ops
    from xrface._lib import lib, ptr, stream
    N, H, W = case
    x = (rnd(f"f64x{case}", N, 64, H, W) * 1.5 + 0.4).bfloat16().float()
    w = rnd(f"f64w{case}", 64, 64, 3, 3, scale=(64 * 9) ** -0.5)
    sc = rnd(f"f64s{case}", N, 64) * 0.3 + 1.0
    sh = rnd(f"f64h{case}", N, 64) * 0.5
    al = rnd(f"f64a{case}", 64).abs() * 0.3 + 0.05
    add = rnd(f"f64e{case}", N, 64, H, W).bfloat16().float()
    z = x * sc[:, :, None, None] + sh[:, :, None, None]
    xn = torch.where(z > 0, z, z * al[None, :, None, None]).bfloat16().float()     # what the kernel feeds the MFMAs
    if transposed:
        y_ref = F.conv_transpose2d(xn, w, None, 1, 1)
    else:
        y_ref = F.conv2d(xn, w, None, 1, 1)
    xb = _nhwc(x).to(DEV).bfloat16()
    wd = w.to(DEV)
    if transposed:
        pk, _ = ops._packed(wd, "dgrad", torch.bfloat16, 64, 1, 9, 64, 64, 9, 0, 1, 64 * 9)
    else:
        pk, _ = ops._packed(wd, "fwd", torch.bfloat16, 64, 1, 9, 64, 64, 64 * 9, 0, 1, 9)
    scd, shd, ald = sc.to(DEV), sh.to(DEV), al.to(DEV)
    # (a) transform on load + output statistics
    y = torch.empty_like(xb)
    stats = torch.zeros(2, N, 64, device=DEV)
    lib.xr_conv64_direct(ptr(xb), ptr(pk), None, ptr(y), N, H, W, transposed, ptr(scd), ptr(shd), ptr(ald), ptr(stats), None, stream())
    torch.cuda.synchronize()
    yf = y.float()
    assert rel(yf.permute(0, 3, 1, 2), y_ref) < 1e-2
    s_ref = torch.stack((yf.sum((1, 2)), (yf * yf).sum((1, 2))))      # statistics are taken of the ROUNDED output
    assert rel(stats, s_ref) < 2e-5
    # (b) affine only (no activation), and ep_add
    z2 = z.bfloat16().float()
    y_ref2 = (F.conv_transpose2d(z2, w, None, 1, 1) if transposed else F.conv2d(z2, w, None, 1, 1)) + add
    y2 = torch.empty_like(xb)
    addb = _nhwc(add).to(DEV).bfloat16()
    lib.xr_conv64_direct(ptr(xb), ptr(pk), None, ptr(y2), N, H, W, transposed, ptr(scd), ptr(shd), None, None, ptr(addb), stream())
    torch.cuda.synchronize()
    assert rel(y2.float().permute(0, 3, 1, 2), y_ref2) < 1e-2


def test_direct_conv64_argument_checks():
    from xrface._lib import lib
    with pytest.raises(RuntimeError, match="xr_conv64_direct"):
        lib.xr_conv64_direct(None, None, None, None, 1, 16, 16, 0, None, None, None, None, None, None)


def _block_run(direct, x, gout, seed=0):
    import xrface
    from xrface import ops
    from xrface.model.FSRnet import _Residual_Block
    old = ops._cfg["direct64"]
    ops._cfg["direct64"] = direct
    xrface.set_compute_dtype(torch.bfloat16)
    try:
        blk = _Residual_Block(64)
        blk.load_state_dict(G.det_state_dict(blk.state_dict(), seed))
        blk.to(DEV)
        xg = x.clone().to(DEV).requires_grad_(True)
        y = blk(xg)
        y.backward(gout.to(DEV))
        torch.cuda.synchronize()
        return y.detach().float().cpu(), xg.grad.float().cpu(), {k: p_.grad.float().cpu() for k, p_ in blk.named_parameters()}
    finally:
        ops._cfg["direct64"] = old
        xrface.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("shape", [(2, 32, 32), (3, 28, 28), (2, 112, 112)])
def test_fused_residual_block_matches_the_unfused_path_and_fp32(shape):
    """_Residual_Block in bf16: the fused op on the direct kernel (statistics in conv epilogues, IN1 + PReLU on conv2's load)
    against the unfused bf16 path (same arithmetic, the normalised activation rounded to bf16 in both) and against the
    fp32 CPU block (model/FSRnet.py:90-98 restated by the oracle)."""
    from oracle import cpu_ref as R
    N, H, W = shape
    x = rnd(f"rbx{shape}", N, 64, H, W).bfloat16().float()
    gout = rnd(f"rbg{shape}", N, 64, H, W).bfloat16().float()
    y_f, dx_f, g_f = _block_run(1, x, gout)
    y_u, dx_u, g_u = _block_run(0, x, gout)
    cos = lambda a, b: float((a.double().flatten() @ b.double().flatten()) / (a.double().norm() * b.double().norm() + 1e-30))
    assert rel(y_f, y_u) < 3e-2 and cos(y_f, y_u) > 0.9999
    assert cos(dx_f, dx_u) > 0.999
    for k in g_u:
        assert cos(g_f[k], g_u[k]) > 0.995, (k, cos(g_f[k], g_u[k]))
    # fp32 reference
    from xrface.model.FSRnet import _Residual_Block
    blk = _Residual_Block(64)
    sd = G.det_state_dict(blk.state_dict(), 0)
    sdg = R.with_grad({"b." + k: v for k, v in sd.items()})
    xr = x.clone().requires_grad_(True)
    y_ref = R.residual_block(sdg, "b", xr)
    y_ref.backward(gout)
    assert rel(y_f, y_ref) < 3e-2 and cos(y_f, y_ref) > 0.9995
    assert cos(dx_f, xr.grad) > 0.998
    for k in g_f:
        assert cos(g_f[k], sdg["b." + k].grad) > 0.99, (k, cos(g_f[k], sdg["b." + k].grad))


# ------------------------------------------------------------------------------------------------ direct weight gradient
WG_CASES = [  # N, H, W
    (2, 16, 16),      # one 16-pixel chunk per row, more workgroups than rows would allow -> one row each
    (3, 24, 40),      # non-square, W not a multiple of 16 (40 -> three chunks, the last half empty)
    (1, 112, 112),    # the FSRNet layer, one image over many workgroups
    (5, 56, 56),      # IR stage 1
    (600, 8, 8),      # 4800 rows over 256 workgroups: runs that start and end mid-image
    (7, 112, 112),    # 784 rows: runs of four rows crossing image boundaries
]


def _unpack(slabs, ns):
    from xrface._lib import lib, ptr, stream
    dw = torch.empty(64, 64, 3, 3, device=DEV)
    lib.xr_unpack_wgrad(ptr(slabs), ptr(dw), 64, 1, 9, 64, 64, 576, 576, 0, 1, 9, 0, ns, stream())
    return dw


@pytest.mark.parametrize("case", WG_CASES)
def test_direct_wgrad64_matches_fp32_and_the_sliced_kernel(case):
    """xr_conv64_wgrad vs (a) the fp32 CPU weight gradient of the same bf16-rounded operands, (b) xr_conv_wgrad (the sliced
    implicit GEMM it replaces: same products, fp32 sums in another order)."""
    from xrface._lib import lib, ptr, stream
    N, H, W = case
    x = rnd(f"wg64x{case}", N, 64, H, W).bfloat16().float()
    gy = rnd(f"wg64g{case}", N, 64, H, W).bfloat16().float()
    ref = torch.nn.grad.conv2d_weight(x.double(), (64, 64, 3, 3), gy.double(), stride=1, padding=1).float()
    xb, gb = _nhwc(x).to(DEV).bfloat16(), _nhwc(gy).to(DEV).bfloat16()
    slabs = torch.full((256, 64, 576), float("nan"), device=DEV)
    ns = lib.xr_conv64_wgrad(ptr(xb), ptr(gb), ptr(slabs), N, H, W, 256, None, None, None, stream())
    assert 1 <= ns <= 256
    dw = _unpack(slabs, ns)
    split = min(64, (N * H * W + 63) // 64)
    slabs2 = torch.empty(split, 64, 576, device=DEV)
    ns2 = lib.xr_conv_wgrad(0, ptr(xb), ptr(gb), ptr(slabs2), N, H, W, 64, H, W, 64, 3, 3, 1, 1, 0, 64, 576, split, stream())
    dw2 = _unpack(slabs2, ns2)
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all()
    assert rel(dw, ref) < 2e-5, rel(dw, ref)          # exact bf16 products, fp32 accumulation
    assert rel(dw, dw2) < 2e-5, rel(dw, dw2)


@pytest.mark.parametrize("case", [(2, 16, 16), (3, 24, 40), (2, 112, 112), (300, 8, 8)])
@pytest.mark.parametrize("with_alpha", [True, False])
def test_direct_wgrad64_transform_on_load(case, with_alpha):
    """in_scale / in_shift / in_alpha: the weight gradient against x' = prelu(x * scale[n] + shift[n], alpha) rounded to bf16,
    the activation the forward's on-load transform fed to the MFMAs (never written to HBM)."""
    from xrface._lib import lib, ptr, stream
    N, H, W = case
    x = rnd(f"wgx64x{case}", N, 64, H, W).bfloat16().float()
    gy = rnd(f"wgx64g{case}", N, 64, H, W).bfloat16().float()
    sc = (rnd(f"wgx64s{case}", N, 64).abs() + 0.5)
    sf = rnd(f"wgx64h{case}", N, 64)
    al = rnd(f"wgx64a{case}", 64).abs() * 0.3
    xt = x * sc[:, :, None, None] + sf[:, :, None, None]
    if with_alpha:
        xt = torch.where(xt > 0, xt, xt * al[None, :, None, None])
    xt = xt.bfloat16().float()
    ref = torch.nn.grad.conv2d_weight(xt.double(), (64, 64, 3, 3), gy.double(), stride=1, padding=1).float()
    xb, gb = _nhwc(x).to(DEV).bfloat16(), _nhwc(gy).to(DEV).bfloat16()
    scd, sfd, ald = sc.to(DEV), sf.to(DEV), al.to(DEV)
    slabs = torch.empty(256, 64, 576, device=DEV)
    ns = lib.xr_conv64_wgrad(ptr(xb), ptr(gb), ptr(slabs), N, H, W, 256, ptr(scd), ptr(sfd), ptr(ald) if with_alpha else None, stream())
    dw = _unpack(slabs, ns)
    torch.cuda.synchronize()
    # the GPU evaluates x * scale + shift as one fused multiply-add: a last-bit difference before the bf16 rounding flips a few
    # of the 2.4e5 .. 2.5e7 rounded activations by one bf16 ulp
    assert rel(dw, ref) < 2e-3, rel(dw, ref)


def test_direct_wgrad64_argument_checks():
    from xrface._lib import lib, ptr, stream
    x = torch.zeros(1, 12, 12, 64, device=DEV, dtype=torch.bfloat16)
    slabs = torch.empty(4, 64, 576, device=DEV)
    with pytest.raises(RuntimeError):
        lib.xr_conv64_wgrad(ptr(x), ptr(x), ptr(slabs), 1, 12, 12, 4, None, None, None, stream())      # W % 8 != 0
    xs = torch.zeros(1, 8, 8, 64, device=DEV, dtype=torch.bfloat16)
    sc = torch.zeros(1, 64, device=DEV)
    with pytest.raises(RuntimeError):
        lib.xr_conv64_wgrad(ptr(xs), ptr(xs), ptr(slabs), 1, 8, 8, 4, ptr(sc), None, None, stream())   # scale without shift
    with pytest.raises(RuntimeError):
        lib.xr_conv64_wgrad(ptr(xs), ptr(xs), ptr(slabs), 1, 8, 8, 0, None, None, None, stream())      # no slab capacity
    ns = lib.xr_conv64_wgrad(ptr(xs), ptr(xs), ptr(slabs), 1, 8, 8, 4, None, None, None, stream())     # capacity caps the grid
    assert ns == 4


# ------------------------------------------------------------------------------------------------ row-walking weight gradient
ROWS_CASES = [  # N, C, K, H, W, stride
    (3, 64, 64, 16, 16, 1),      # one chunk per row, RS = 14 rows per step over images of 16 rows
    (5, 128, 64, 14, 14, 1),     # the 14 x 14 layers: one image per step; two input-channel tiles
    (2, 64, 192, 28, 28, 1),     # RS = 7, three output-channel tiles
    (2, 128, 128, 56, 40, 1),    # RS = 3, non-square, 40 columns in a 64-column plan
    (1, 64, 128, 112, 112, 1),   # RS = 1 (the 112-wide plan), K != C
    (2, 64, 64, 112, 112, 2),    # stride 2 at the input resolution (model_irse.py:60, first block)
    (3, 128, 64, 56, 56, 2),     # stride 2, RS = 3
    (5, 64, 128, 28, 28, 2),     # stride 2, RS = 6 over images of 14 output rows
    (70, 64, 64, 16, 20, 1),     # more steps than one pass: ragged last step, runs that start mid-image
    (9, 64, 64, 32, 24, 2),      # stride 2, ragged widths (12 output columns in a 16-column plan)
    (4, 64, 64, 7, 7, 1),        # the 7 x 7 maps (not routed here by default: 7 of 16 MFMA pixels used), two images per step
    (4, 128, 64, 14, 14, 2),     # stride 2 down to 7 x 7
    (2, 64, 128, 9, 13, 1),      # odd sizes: 13-pixel rows, staging groups that are not a multiple of 64 chunks
    (1, 64, 64, 3, 8, 1),        # fewer rows than one step
]


@pytest.mark.parametrize("case", ROWS_CASES)
def test_rows_wgrad_matches_fp64_and_the_sliced_kernel(case):
    """xr_conv_wgrad_rows vs the fp64 CPU weight gradient of the same bf16-rounded operands and vs xr_conv_wgrad (same products)."""
    from xrface._lib import lib, ptr, stream
    N, C, K, H, W, s = case
    Ho, Wo = H // s, W // s
    x = rnd(f"wgrx{case}", N, C, H, W).bfloat16().float()
    gy = rnd(f"wgrg{case}", N, K, Ho, Wo).bfloat16().float()
    ref = torch.nn.grad.conv2d_weight(x.double(), (K, C, 3, 3), gy.double(), stride=s, padding=1).float()
    xb, gb = _nhwc(x).to(DEV).bfloat16(), _nhwc(gy).to(DEV).bfloat16()
    kg = 9 * C
    cap = max(1, 256 // ((K // 64) * (C // 64)))
    slabs = torch.full((cap, K, kg), float("nan"), device=DEV)
    ns = lib.xr_conv_wgrad_rows(ptr(xb), ptr(gb), ptr(slabs), N, H, W, C, K, s, cap, stream())
    assert 1 <= ns <= cap
    dw = torch.empty(K, C, 3, 3, device=DEV)
    lib.xr_unpack_wgrad(ptr(slabs), ptr(dw), K, 1, 9, C, C, kg, 9 * C, 0, 1, 9, 0, ns, stream())
    split = min(32, (N * Ho * Wo + 63) // 64)
    slabs2 = torch.empty(split, K, kg, device=DEV)
    ns2 = lib.xr_conv_wgrad(0, ptr(xb), ptr(gb), ptr(slabs2), N, H, W, C, Ho, Wo, K, 3, 3, s, 1, 0, K, kg, split, stream())
    dw2 = torch.empty(K, C, 3, 3, device=DEV)
    lib.xr_unpack_wgrad(ptr(slabs2), ptr(dw2), K, 1, 9, C, C, kg, 9 * C, 0, 1, 9, 0, ns2, stream())
    torch.cuda.synchronize()
    assert torch.isfinite(dw).all()
    assert rel(dw, ref) < 2e-5, rel(dw, ref)
    assert rel(dw, dw2) < 2e-5, rel(dw, dw2)


def test_rows_wgrad_argument_checks():
    from xrface._lib import lib, ptr, stream
    x = torch.zeros(1, 16, 16, 64, device=DEV, dtype=torch.bfloat16)
    slabs = torch.empty(2, 64, 576, device=DEV)
    with pytest.raises(RuntimeError):
        lib.xr_conv_wgrad_rows(ptr(x), ptr(x), ptr(slabs), 1, 16, 16, 48, 64, 1, 2, stream())     # C % 64
    with pytest.raises(RuntimeError):
        lib.xr_conv_wgrad_rows(ptr(x), ptr(x), ptr(slabs), 1, 16, 16, 64, 64, 3, 2, stream())     # stride
    with pytest.raises(RuntimeError):
        lib.xr_conv_wgrad_rows(ptr(x), ptr(x), ptr(slabs), 1, 15, 16, 64, 64, 2, 2, stream())     # H % stride
    assert lib.xr_conv_wgrad_rows(ptr(x), ptr(x), ptr(slabs), 1, 16, 16, 64, 64, 1, 2, stream()) in (1, 2)


@pytest.mark.parametrize("case", [(2, 32, 32), (3, 20, 28), (2, 112, 112), (300, 16, 16)])
@pytest.mark.parametrize("with_alpha", [True, False])
def test_direct_conv64_backward_reduction_epilogue(case, with_alpha):
    """xr_conv64_direct_bwdred: same output as the plain input-gradient launch, and its three per-image sums equal those of
    xr_affine_act_bwd_reduce run over (c1, the bf16 output) -- the pass it replaces."""
    from xrface import ops
    from xrface._lib import lib, ptr, stream, ACT_PRELU
    N, H, W = case
    HW = H * W
    g = _nhwc(rnd(f"brg{case}", N, 64, H, W)).to(DEV).bfloat16()
    c1 = _nhwc(rnd(f"brc{case}", N, 64, H, W)).to(DEV).bfloat16()
    w = rnd(f"brw{case}", 64, 64, 3, 3, scale=(64 * 9) ** -0.5).to(DEV)
    sc = (rnd(f"brs{case}", N, 64).abs() + 0.5).to(DEV)
    sf = rnd(f"brh{case}", N, 64).to(DEV)
    al = (rnd(f"bra{case}", 64).abs() * 0.3).to(DEV) if with_alpha else torch.ones(64, device=DEV)
    pkd, _ = ops._packed(w, "dgrad", torch.bfloat16, 64, 1, 9, 64, 64, 9, 0, 1, 576)
    out0 = torch.empty_like(g)
    lib.xr_conv64_direct(ptr(g), ptr(pkd), None, ptr(out0), N, H, W, 1, None, None, None, None, None, stream())
    out = torch.empty_like(g)
    red = torch.zeros(3, N, 64, device=DEV)
    lib.xr_conv64_direct_bwdred(ptr(g), ptr(pkd), ptr(out), N, H, W, 1, ptr(c1), ptr(sc), ptr(sf), ptr(al) if with_alpha else None,
                                ptr(red), stream())
    ref = torch.zeros(3, N, 64, device=DEV)
    lib.xr_affine_act_bwd_reduce(0, ptr(c1), ptr(sc), ptr(sf), None, ptr(al), ACT_PRELU, ptr(out0), ptr(ref), N, HW, 64, 1, stream())
    torch.cuda.synchronize()
    assert torch.equal(out, out0)
    for k in range(2 if not with_alpha else 3):        # slope 1: the PReLU slope term is not meaningful
        assert rel(red[k], ref[k]) < 2e-5, (k, rel(red[k], ref[k]))


@pytest.mark.parametrize("shape", [(2, 32, 32), (3, 24, 40), (2, 112, 112)])
@pytest.mark.parametrize("flat", [False, True])
def test_chained_residual_trunk_matches_block_by_block(shape, flat):
    """ops._ResTrunk64 (three blocks x three passes as one op, tails chained through xr_conv64_direct_tailred) against the same
    blocks applied one _ResBlock64 at a time, both in bf16, and both against the fp32 parity mode of the same nine applications:
    the chained path must be as close to fp32 as the unchained one (it rounds the tail gradient to bf16 once more, nothing else
    differs).  With FlatParams the gradients are accumulated in place at nine use sites per parameter."""
    import xrface
    from xrface import ops, parallel
    from xrface.model.FSRnet import _Residual_Block
    from xrface.nn import res_trunk
    import torch.nn as nn
    N, H, W = shape
    x = rnd(f"trx{shape}", N, 64, H, W).bfloat16().float().to(DEV)
    gout = rnd(f"trg{shape}", N, 64, H, W).bfloat16().float().to(DEV)
    res = {}
    try:
        for mode in ("trunk", "blocks", "fp32"):
            xrface.set_compute_dtype(torch.float32 if mode == "fp32" else torch.bfloat16)
            seq = nn.Sequential(*[_Residual_Block(64) for _ in range(3)])
            for i, b in enumerate(seq):
                b.load_state_dict(G.det_state_dict(b.state_dict(), 10 + i))
            seq.to(DEV)
            if flat:
                parallel.FlatParams(seq.parameters()).zero_grad()
            ops._cfg["res_trunk"] = 1 if mode == "trunk" else 0
            xin = x.clone().requires_grad_(True)
            y = ops.leave(res_trunk(seq, ops.enter(xin), 3))
            y.backward(gout)
            ops.join_side_stream()
            torch.cuda.synchronize()
            res[mode] = (y.detach().float(), xin.grad.detach().float(), {k: p.grad.detach().float().clone() for k, p in seq.named_parameters()})
    finally:
        ops._cfg["res_trunk"] = 1
        xrface.set_compute_dtype(torch.float32)
    cos = lambda a, b: float((a.double().flatten() @ b.double().flatten()) / (a.double().norm() * b.double().norm() + 1e-30))
    yt, dxt, gt = res["trunk"]
    yb, dxb, gb = res["blocks"]
    yr, dxr, gr = res["fp32"]
    assert cos(yt, yb) > 0.99995 and rel(yt, yb) < 3e-2        # same forward arithmetic up to the order of the fp32 statistic atomics
    ct, cb = cos(dxt, dxr), cos(dxb, dxr)
    assert ct > 0.998 and ct > cb - 1e-3, (ct, cb)
    for k in gr:
        ct, cb = cos(gt[k], gr[k]), cos(gb[k], gr[k])
        assert ct > 0.99 and ct > cb - 5e-3, (k, ct, cb)
        assert cos(gt[k], gb[k]) > 0.995, (k, cos(gt[k], gb[k]))


@pytest.mark.parametrize("n,h,w", [(3, 20, 24), (2, 112, 112), (5, 7, 7)])
def test_direct_conv_with_prelu_second_output(n, h, w):
    """xr_conv64_direct_prelu (conv1 of the 64-channel IR units, model_irse.py:57-59): y against the fp32 reference convolution of
    the bf16 operands, and the second output bit for bit = PReLU of the STORED y (what xr_conv_igemm's ep2_out delivers)."""
    import xrface
    from xrface import ops
    from xrface._lib import lib, ptr, stream
    g = torch.Generator().manual_seed(n * 100 + h)
    x = torch.randn(n, h, w, 64, generator=g).bfloat16().to(DEV)
    wt = (torch.randn(64, 64, 3, 3, generator=g) / 24).to(DEV)
    alpha = (torch.rand(64, generator=g) * 0.5).to(DEV)
    xrface.set_compute_dtype(torch.bfloat16)
    try:
        pk, kg = ops._packed(wt, "fwd", torch.bfloat16, 64, 1, 9, 64, 64, 64 * 9, 0, 1, 9)
        y, p2 = torch.empty_like(x), torch.empty_like(x)
        lib.xr_conv64_direct_prelu(ptr(x), ptr(pk), ptr(y), ptr(p2), ptr(alpha), n, h, w, stream())
        torch.cuda.synchronize()
    finally:
        xrface.set_compute_dtype(torch.float32)
    ref = torch.nn.functional.conv2d(x.float().permute(0, 3, 1, 2).cpu(), wt.bfloat16().float().cpu(), padding=1).permute(0, 2, 3, 1)
    err = float((y.float().cpu() - ref).abs().max() / ref.abs().max())
    assert err < 1e-2, err
    yf = y.float()
    want = torch.where(yf > 0, yf, yf * alpha.view(1, 1, 1, 64)).bfloat16()
    assert torch.equal(p2, want)
