"""GPU: every HIP op against a plain PyTorch fp32 CPU reference of the same op, through the C ABI.
Tolerances: XR_F32 mode (split-bf16 MFMA, fp32 accumulate) 2e-4 relative to max-abs -- well inside the
1e-3 of BASELINE.json's north_star; XR_BF16 mode 2e-2 (bf16 has 8 significant bits)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import detgen as G

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = {torch.float32: 2e-4, torch.bfloat16: 2.5e-2}


def rnd(key, *shape, scale=1.0):
    return torch.from_numpy((G.normal(key, int(np.prod(shape))) * scale).reshape(shape).astype(np.float32))


def rel(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def to_buf(x_nchw, dtype):
    from xrface import ops
    return ops.enter(x_nchw.to(DEV), dtype)


def from_buf(buf, C):
    return buf.float().permute(0, 3, 1, 2)[:, :C].cpu()


CONV_CASES = [
    # N, C, H, W, K, R, stride, pad, bias
    (2, 64, 20, 20, 64, 3, 1, 1, False),
    (2, 3, 24, 24, 64, 3, 1, 1, True),
    (1, 64, 14, 14, 3, 3, 1, 1, True),
    (2, 64, 28, 28, 128, 3, 2, 1, False),
    (3, 128, 9, 9, 128, 3, 1, 1, False),
    (2, 3, 56, 56, 128, 7, 4, 3, True),
    (2, 3, 40, 40, 64, 7, 2, 3, False),
    (2, 64, 16, 16, 128, 1, 2, 0, False),
    (2, 128, 7, 7, 97, 1, 1, 0, True),
    (1, 192, 12, 12, 64, 3, 1, 1, True),
    (2, 256, 7, 7, 512, 3, 2, 1, False),
]


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("case", CONV_CASES)
def test_conv2d_fwd_bwd(case, dtype):
    from xrface import ops
    N, C, H, W, K, R, stride, pad, bias = case
    x = rnd(f"cx{case}", N, C, H, W)
    w = rnd(f"cw{case}", K, C, R, R, scale=(C * R * R) ** -0.5)
    b = rnd(f"cb{case}", K, scale=0.1) if bias else None
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    y_ref = F.conv2d(xr, wr, br, stride, pad)
    gy = rnd(f"cg{case}", *y_ref.shape)
    y_ref.backward(gy)

    xg = x.to(DEV).requires_grad_(True)
    wg = w.to(DEV).requires_grad_(True)
    bg = b.to(DEV).requires_grad_(True) if bias else None
    buf = ops.enter(xg, dtype)
    yb = ops.conv2d(buf, wg, bg, stride, pad)
    assert yb.shape == (N, y_ref.shape[2], y_ref.shape[3], ops.r8(K))
    if ops.r8(K) != K:
        assert float(yb[..., K:].abs().max()) == 0.0, "channel padding must be zero"
    y = ops.leave(yb, K)
    tol = TOL[dtype]
    assert rel(y, y_ref) < tol
    y.backward(gy.to(DEV))
    assert rel(xg.grad, xr.grad) < tol
    assert rel(wg.grad, wr.grad) < tol
    if bias:
        assert rel(bg.grad, br.grad) < tol


@pytest.mark.parametrize("case", [CONV_CASES[0], CONV_CASES[3], CONV_CASES[5], CONV_CASES[8], CONV_CASES[10]])
def test_conv2d_two_plane_fp32_mode(case):
    """XR_F32X2 (ops.set_compute_dtype("fp32x2")): fp32 tensors, operands split into hi + lo bf16 planes, three plane-pair MFMAs per
    product.  Forward, input gradient and weight gradient against the fp32 CPU convolution: ~16 significand bits per operand
    -> a few 1e-5 relative to max-abs (bar 1e-4; the three-plane mode sits at ~1e-6, bf16 at ~1e-2), and measurably MORE than
    the three-plane mode on the same operands (i.e. the two-plane kernels really ran)."""
    from xrface import ops
    N, C, H, W, K, R, stride, pad, bias = case
    x = rnd(f"cx{case}", N, C, H, W)
    w = rnd(f"cw{case}", K, C, R, R, scale=(C * R * R) ** -0.5)
    xr, wr = x.clone().requires_grad_(True), w.clone().requires_grad_(True)
    y_ref = F.conv2d(xr, wr, None, stride, pad)
    gy = rnd(f"cg{case}", *y_ref.shape)
    y_ref.backward(gy)
    errs = {}
    for mode in ("fp32x2", torch.float32):
        ops.set_compute_dtype(mode)
        xg, wg = x.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
        y = ops.leave(ops.conv2d(ops.enter(xg, torch.float32), wg, None, stride, pad), K)
        y.backward(gy.to(DEV))
        errs[mode] = (rel(y, y_ref), rel(xg.grad, xr.grad), rel(wg.grad, wr.grad))
    assert max(errs["fp32x2"]) < 1e-4, errs
    assert max(errs[torch.float32]) < 2e-5, errs
    assert min(errs["fp32x2"]) > 1.5 * max(errs[torch.float32]) or max(errs["fp32x2"]) < 2e-6, errs


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("shape", [(3, 64, 20, 12, 64), (2, 128, 16, 16, 128), (2, 64, 8, 8, 128)])
def test_optional_strided_conv_forms(shape, dtype):
    """Two alternative forms of the stage-opening layers, off by default because they measured no faster, kept correct:
    the 3x3 / stride-2 input gradient as one dense 2x2-window GEMM with a depth-to-space epilogue (xr_conv_dgrad_s2; PReLU-backward
    and BatchNorm-backward epilogues addressed on the up-sampled grid) and the 1x1 / stride-2 shortcut convolution as sub-sample +
    stride-1 GEMM.  A bottleneck_IR_SE stage opening (model_irse.py:69-91) with both forms on against the default forms."""
    import copy
    import xrface
    from xrface import ops
    from xrface.model.model_irse import bottleneck_IR_SE
    n, cin, h, w, depth = shape
    xrface.set_compute_dtype(dtype)
    try:
        torch.manual_seed(9)
        blk0 = bottleneck_IR_SE(cin, depth, 2).to(DEV).train()
        x0 = rnd(f"s2f{shape}", n, cin, h, w)
        res = {}
        for mode in (0, 1):
            ops._cfg["dgrad_s2"] = ops._cfg["conv1x1_subsample"] = mode
            blk = copy.deepcopy(blk0)
            x = x0.to(DEV).requires_grad_(True)
            out = blk(x)
            out.square().mean().backward()
            torch.cuda.synchronize()
            res[mode] = [out.detach().float().cpu(), x.grad.cpu()] + [p.grad.cpu() for p in blk.parameters()]
        tol = 2e-3 if dtype == torch.float32 else 8e-2
        for a, b in zip(res[1], res[0]):
            assert rel(a, b) < tol, rel(a, b)
    finally:
        ops._cfg["dgrad_s2"] = ops._cfg["conv1x1_subsample"] = 0
        xrface.set_compute_dtype(torch.float32)


CONV8_CASES = [
    # N, C, H, W, K, R, stride, pad, bias  -- the 8-wave 256x256 kernel forced on (xr_tune knob 7 = 2)
    (2, 64, 14, 14, 256, 3, 1, 1, False),    # 392 rows: one full + one ragged row tile
    (3, 128, 10, 10, 264, 3, 1, 1, True),    # two column tiles, the second 8 wide; bias
    (2, 64, 28, 28, 128, 3, 2, 1, False),    # stride-2 gather (dgrad falls back to the class-mode kernel)
    (5, 192, 8, 8, 72, 1, 1, 0, True),       # 1x1, three channel chunks per tap
    (1, 64, 16, 16, 512, 3, 1, 1, False),    # exactly one row tile, two column tiles
]


@pytest.mark.parametrize("deep", [0, 1])   # knob 12: prefetch schedule (3-4 / 5-6 phases ahead)
@pytest.mark.parametrize("cfg", [3, 4, 5, 6, 7, 8])   # xr_tune knob 7: force tile configuration 256x256, 256x128, 128x256, 224x256, 512x128, 448x128
@pytest.mark.parametrize("case", CONV8_CASES)
def test_conv2d_8wave_kernel(case, cfg, deep):
    """xr_conv8.hip (8-wave ping-pong tile) against the fp32 CPU conv: forward, input gradient (same kernel, transposed
    gather) and -- unchanged kernel, sanity only -- the weight gradient."""
    from xrface import ops
    from xrface._lib import lib
    N, C, H, W, K, R, stride, pad, bias = case
    x = rnd(f"c8x{case}", N, C, H, W).bfloat16().float()
    w = rnd(f"c8w{case}", K, C, R, R, scale=(C * R * R) ** -0.5)
    b = rnd(f"c8b{case}", K, scale=0.1) if bias else None
    xr = x.clone().requires_grad_(True)
    wr = w.clone().requires_grad_(True)
    br = b.clone().requires_grad_(True) if bias else None
    y_ref = F.conv2d(xr, wr, br, stride, pad)
    gy = rnd(f"c8g{case}", *y_ref.shape)
    y_ref.backward(gy)
    lib.xr_tune(7, cfg)
    lib.xr_tune(12, deep)
    try:
        xg = x.to(DEV).requires_grad_(True)
        wg = w.to(DEV).requires_grad_(True)
        bg = b.to(DEV).requires_grad_(True) if bias else None
        yb = ops.conv2d(ops.enter(xg, torch.bfloat16), wg, bg, stride, pad)
        y = ops.leave(yb, K)
        y.backward(gy.to(DEV))
        torch.cuda.synchronize()
    finally:
        lib.xr_tune(7, 1)
        lib.xr_tune(12, 1)
    tol = TOL[torch.bfloat16]
    assert rel(y, y_ref) < tol
    assert rel(xg.grad, xr.grad) < tol
    assert rel(wg.grad, wr.grad) < tol
    # and bit-for-bit stable against the 4-wave kernel up to accumulation order: compare on the same inputs
    lib.xr_tune(7, 0)
    try:
        y4 = ops.leave(ops.conv2d(ops.enter(x.to(DEV), torch.bfloat16), w.to(DEV), b.to(DEV) if bias else None, stride, pad), K)
    finally:
        lib.xr_tune(7, 1)
    assert rel(y, y4) < 1e-2


@pytest.mark.parametrize("shape", [(256, 256, 256, 14, 1), (128, 128, 128, 28, 1), (64, 256, 512, 14, 1), (32, 128, 256, 28, 2)])
def test_conv2d_8wave_race_screen(shape):
    """The 8-wave kernel orders its LDS-DMA prefetch by counted vmcnt + raw barriers only, so a wrong count would show up
    as a rare wrong tile.  Screen at the benchmark's full sizes: 60 back-to-back launches (forward and transposed gather,
    both prefetch schedules) must reproduce the first launch bit for bit, and the first launch must agree with the 4-wave
    kernel up to accumulation order."""
    from xrface import ops
    from xrface._lib import lib, ptr, stream, dt
    N, C, K, H, st = shape
    Ho = (H + 2 - 3) // st + 1
    g = torch.Generator(device=DEV).manual_seed(7)
    x = torch.randn(N, H, H, C, device=DEV, generator=g).bfloat16()
    dy = torch.randn(N, Ho, Ho, K, device=DEV, generator=g).bfloat16()
    w = torch.randn(K, C, 3, 3, device=DEV, generator=g) * 0.05
    pk, kg = ops._packed(w, "fwd", torch.bfloat16, K, 1, 9, C, C, C * 9, 0, 1, 9)
    pkd, kgd = ops._packed(w, "dgrad", torch.bfloat16, C, 1, 9, K, K, 9, 0, 1, C * 9)

    def fwd(out):
        lib.xr_conv_igemm(dt(x), ptr(x), ptr(pk), None, ptr(out), N, H, H, C, Ho, Ho, K, 3, 3, st, 1, 0, kg, K, None, 0, None, None,
                          None, 1, None, None, None, stream())

    def dgrad(out):
        lib.xr_conv_igemm(dt(x), ptr(dy), ptr(pkd), None, ptr(out), N, Ho, Ho, K, H, H, C, 3, 3, st, 1, 1, kgd, C, None, 0, None, None,
                          None, 1, None, None, None, stream())

    try:
        lib.xr_tune(7, 0)
        y4, dx4 = torch.empty_like(dy), torch.empty_like(x)
        fwd(y4); dgrad(dx4)
        for deep in (0, 1):
            lib.xr_tune(7, 2)
            lib.xr_tune(12, deep)
            y0, dx0 = torch.empty_like(dy), torch.empty_like(x)
            fwd(y0); dgrad(dx0)
            assert rel(y0, y4) < 1e-2 and rel(dx0, dx4) < 1e-2
            bad = 0
            ys, dxs = [torch.empty_like(dy) for _ in range(4)], [torch.empty_like(x) for _ in range(4)]
            for it in range(60):
                fwd(ys[it % 4]); dgrad(dxs[it % 4])
                if it % 4 == 3:
                    bad += sum(int(not torch.equal(t, y0)) for t in ys) + sum(int(not torch.equal(t, dx0)) for t in dxs)
            assert bad == 0, f"{bad} launches differ from the first one (prefetch schedule deep={deep})"
    finally:
        lib.xr_tune(7, 1)
        lib.xr_tune(12, 1)


def test_conv2d_8wave_prelu_backward_epilogue():
    """dgrad of conv(prelu(y)) with the PReLU backward fused into the 8-wave kernel's epilogue."""
    from xrface import ops
    from xrface._lib import lib
    N, C, H, K = 2, 256, 14, 64
    y0 = rnd("c8py", N, C, H, H).bfloat16().float()
    al = rnd("c8pa", C, scale=0.3).abs() + 0.05
    w = rnd("c8pw", K, C, 3, 3, scale=(C * 9) ** -0.5)
    yr, ar, wr = y0.clone().requires_grad_(True), al.clone().requires_grad_(True), w.clone().requires_grad_(True)
    out_ref = F.conv2d(F.prelu(yr, ar), wr, None, 1, 1)
    g = rnd("c8pg", *out_ref.shape)
    out_ref.backward(g)
    res = {}
    for knob in (3, 4, 5, 6, 7, 8, 0):
        lib.xr_tune(7, knob)
        try:
            yg, ag, wg = y0.to(DEV).requires_grad_(True), al.to(DEV).requires_grad_(True), w.to(DEV).requires_grad_(True)
            out = ops.leave(ops.prelu_conv2d(ops.enter(yg, torch.bfloat16), ag, wg, 1, 1), K)
            out.backward(g.to(DEV))
            torch.cuda.synchronize()
        finally:
            lib.xr_tune(7, 1)
        res[knob] = (out.detach().cpu(), yg.grad.cpu(), ag.grad.cpu())
        assert rel(out, out_ref) < TOL[torch.bfloat16]
        assert rel(yg.grad, yr.grad) < TOL[torch.bfloat16]
        assert rel(ag.grad, ar.grad) < 4e-2
    for knob in (3, 4, 5, 6, 7, 8):
        assert rel(res[knob][1], res[0][1]) < 1e-2


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_conv_transpose2d_k7s4(dtype):
    from xrface import ops
    N, Cin, Cout, H = 2, 64, 64, 7
    x = rnd("dx", N, Cin, H, H)
    w = rnd("dw", Cin, Cout, 7, 7, scale=0.05)
    b = rnd("db", Cout, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    y_ref = F.conv_transpose2d(xr, wr, br, stride=4, padding=2, output_padding=1)
    gy = rnd("dg", *y_ref.shape)
    y_ref.backward(gy)
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.leave(ops.conv_transpose2d(ops.enter(xg, dtype), wg, bg, 4, 2, 1), Cout)
    assert y.shape == y_ref.shape
    tol = TOL[dtype]
    assert rel(y, y_ref) < tol
    y.backward(gy.to(DEV))
    assert rel(xg.grad, xr.grad) < tol
    assert rel(wg.grad, wr.grad) < tol
    assert rel(bg.grad, br.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_linear_flatten_order(dtype):
    from xrface import ops
    N, C, H, K = 5, 64, 7, 512
    x = rnd("lx", N, C, H, H)
    w = rnd("lw", K, C * H * H, scale=(C * H * H) ** -0.5)
    b = rnd("lb", K, scale=0.1)
    xr, wr, br = (t.clone().requires_grad_(True) for t in (x, w, b))
    y_ref = F.linear(xr.flatten(1), wr, br)
    gy = rnd("lg", N, K)
    y_ref.backward(gy)
    xg, wg, bg = (t.to(DEV).requires_grad_(True) for t in (x, w, b))
    y = ops.leave2d(ops.linear_nhwc(ops.enter(xg, dtype), wg, bg))
    tol = TOL[dtype]
    assert rel(y, y_ref) < tol
    y.backward(gy.to(DEV).to(y.dtype))
    assert rel(xg.grad, xr.grad) < tol
    assert rel(wg.grad, wr.grad) < tol
    assert rel(bg.grad, br.grad) < tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode,act,with_res", [("in", "prelu", True), ("in", None, False), ("bn", "relu", True),
                                               ("bn", "prelu", False), ("bn_eval", None, True), ("none", "prelu", False),
                                               ("in_noaffine", "prelu", True)])
def test_norm_act(mode, act, with_res, dtype):
    from xrface import ops
    N, C, H, W = 3, 64, 10, 12
    x = rnd("nx" + mode, N, C, H, W) * 1.5 + 0.3
    res = rnd("nr" + mode, N, C, H, W) if with_res else None
    gamma, beta = rnd("ng", C) * 0.3 + 1.0, rnd("nb", C) * 0.2
    alpha = rnd("na", C) * 0.1 + 0.3
    rm, rv = rnd("nm", C) * 0.1, rnd("nv", C).abs() + 0.5
    if dtype == torch.bfloat16:  # compare on the same bf16-rounded inputs
        x, res = x.bfloat16().float(), (res.bfloat16().float() if with_res else None)
    leaves = [t.clone().requires_grad_(True) for t in (x, gamma, beta, alpha)]
    xr, gr, br, ar = leaves
    rr = res.clone().requires_grad_(True) if with_res else None
    rm_ref, rv_ref = rm.clone(), rv.clone()
    if mode == "in":
        z = F.instance_norm(xr, None, None, gr, br, True, 0.0, 1e-5)
    elif mode == "in_noaffine":
        z = F.instance_norm(xr, None, None, None, None, True, 0.0, 1e-5)
    elif mode == "bn":
        z = F.batch_norm(xr, rm_ref, rv_ref, gr, br, True, 0.1, 1e-5)
    elif mode == "bn_eval":
        z = F.batch_norm(xr, rm_ref, rv_ref, gr, br, False, 0.1, 1e-5)
    else:
        z = xr
    if with_res:
        z = z + rr
    y_ref = F.prelu(z, ar) if act == "prelu" else (F.relu(z) if act == "relu" else z)
    gy = rnd("ngy" + mode, N, C, H, W)
    y_ref.backward(gy)

    xg, gg, bg, ag = (t.detach().clone().to(DEV).requires_grad_(True) for t in (x, gamma, beta, alpha))
    rg = res.to(DEV).requires_grad_(True) if with_res else None
    rmg, rvg = rm.to(DEV), rv.to(DEV)
    buf = ops.enter(xg, dtype)
    rbuf = ops.enter(rg, dtype) if with_res else None
    kind = {"in": "in", "in_noaffine": "in", "bn": "bn", "bn_eval": "bn", "none": "none"}[mode]
    affine = mode in ("in", "bn", "bn_eval")
    yb = ops.norm_act(buf, gg if affine else None, bg if affine else None, rmg if kind == "bn" else None,
                      rvg if kind == "bn" else None, rbuf, ag if act == "prelu" else None, kind, act,
                      training=(mode != "bn_eval"))
    y = ops.leave(yb)
    tol = TOL[dtype]
    assert rel(y, y_ref) < tol
    y.backward(gy.to(DEV).to(y.dtype))
    gtol = tol * 3
    assert rel(xg.grad, xr.grad) < gtol
    if with_res:
        assert rel(rg.grad, rr.grad) < gtol
    if affine:
        assert rel(gg.grad, gr.grad) < gtol
        assert rel(bg.grad, br.grad) < gtol
    if act == "prelu":
        assert rel(ag.grad, ar.grad) < gtol
    if mode == "bn":
        assert rel(rmg, rm_ref) < 1e-4 and rel(rvg, rv_ref) < 1e-4


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("mode", ["bn", "in", "bn_chained"])
def test_norm_statistics_of_an_offset_input(mode, dtype):
    """Batch / instance statistics of an input with |mean| >> std (mean ~ 1e3, std ~ 1 in fp32; bf16 inputs are quantised to the
    bf16 grid first, spacing 4-8 at 1e3, both sides see the same values) against aten::batch_norm / instance_norm in fp32
    (model_irse.py:56-60 BatchNorm semantics, model/FSRnet.py:81 InstanceNorm): E[x^2] - mean^2 on raw fp32 sums loses the
    variance entirely here (the clamp then gives invstd = 1 / sqrt(eps)); the pivoted sums must not.  ``bn_chained``: the
    statistics come out of the elementwise pass that produced the tensor (xr_affine_act_stats_pivot, TailLink / offer_stats)."""
    from xrface import ops
    N, C, H, W = 6, 64, 12, 10
    off = (rnd("off", C) * 300.0 + 1000.0).reshape(1, C, 1, 1)
    x = rnd("offx" + mode, N, C, H, W) * (1.0 if dtype == torch.float32 else 16.0) + off
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    gamma, beta = rnd("og", C) * 0.3 + 1.0, rnd("ob", C) * 0.2
    rm, rv = torch.zeros(C), torch.ones(C)
    xr = x.clone().requires_grad_(True)
    rm_ref, rv_ref = rm.clone(), rv.clone()
    if mode == "in":
        y_ref = F.instance_norm(xr, None, None, gamma, beta, True, 0.0, 1e-5)
    else:
        y_ref = F.batch_norm(xr, rm_ref, rv_ref, gamma, beta, True, 0.1, 1e-5)
    gy = rnd("ogy", N, C, H, W)
    y_ref.backward(gy)
    xg = x.clone().to(DEV).requires_grad_(True)
    rmg, rvg = rm.to(DEV), rv.to(DEV)
    gg, bg = gamma.to(DEV), beta.to(DEV)
    if mode == "bn_chained":
        # producer: an identity elementwise pass (scale 1, shift 0) that offers the statistics of its output to the BatchNorm
        ident = ops.enter(xg, dtype).detach()
        osum = ops.zeros_f32((2, N, C), ident.device)
        opiv = torch.empty((N, C), dtype=torch.float32, device=DEV)
        buf = torch.empty_like(ident)
        from xrface._lib import ACT_NONE, dt, lib, ptr, stream
        lib.xr_affine_act_stats_pivot(dt(ident), ptr(ident.detach()), None, None, None, None, ACT_NONE, ptr(buf), ptr(osum), ptr(opiv), N,
                                      H * W, C, 0, stream())
        sl = ops.StatsLink()
        sl.deliver(buf, osum, opiv)
        yb = ops.norm_act(buf, gg, bg, rmg, rvg, mode="bn", training=True, slink=sl)
        assert sl.sums is None       # taken
    else:
        yb = ops.norm_act(ops.enter(xg, dtype), gg, bg, rmg if mode == "bn" else None, rvg if mode == "bn" else None,
                          mode="in" if mode == "in" else "bn", training=True)
    y = ops.leave(yb)
    # outputs are O(1): absolute error against the fp32 reference (bf16 mode: the OUTPUT is rounded to bf16, 2^-8 relative)
    tol = 2e-3 if dtype == torch.float32 else 3e-2
    err = float((y.detach().float().cpu() - y_ref.detach()).abs().max())
    assert err < tol, err
    if mode != "in":
        assert rel(rmg, rm_ref) < 1e-5 and rel(rvg, rv_ref) < (2e-3 if dtype == torch.float32 else 1e-4), (rel(rmg, rm_ref), rel(rvg, rv_ref))
    if mode != "bn_chained":
        y.backward(gy.to(DEV).to(y.dtype))
        gerr = float((xg.grad.float().cpu() - xr.grad).abs().max() / xr.grad.abs().max())
        assert gerr < (2e-2 if dtype == torch.float32 else 6e-2), gerr


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_se_scale_add(dtype):
    from xrface import ops
    N, C, H = 3, 64, 9
    r = rnd("ser", N, C, H, H)
    sc = rnd("ses", N, C, H, H)
    w1 = rnd("sew1", C // 16, C, 1, 1, scale=0.3)
    w2 = rnd("sew2", C, C // 16, 1, 1, scale=0.5)
    if dtype == torch.bfloat16:
        r, sc = r.bfloat16().float(), sc.bfloat16().float()
    rr, sr, w1r, w2r = (t.clone().requires_grad_(True) for t in (r, sc, w1, w2))
    s = torch.sigmoid(F.conv2d(F.relu(F.conv2d(rr.mean((2, 3), keepdim=True), w1r)), w2r))
    y_ref = rr * s + sr
    gy = rnd("seg", N, C, H, H)
    y_ref.backward(gy)
    rg, sg, w1g, w2g = (t.to(DEV).requires_grad_(True) for t in (r, sc, w1, w2))
    y = ops.leave(ops.se_scale_add(ops.enter(rg, dtype), w1g, w2g, ops.enter(sg, dtype)))
    tol = TOL[dtype]
    assert rel(y, y_ref) < tol
    y.backward(gy.to(DEV).to(y.dtype))
    for a, b in ((rg, rr), (sg, sr), (w1g, w1r), (w2g, w2r)):
        assert rel(a.grad, b.grad) < 3 * tol


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_resampling_ops(dtype):
    from xrface import ops
    N, C, H = 2, 128, 12
    x = rnd("rsx", N, C, H, H)
    if dtype == torch.bfloat16:
        x = x.bfloat16().float()
    # subsample == MaxPool2d(1, 2)
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 1, 2)
    gy = rnd("rsg", *y_ref.shape)
    y_ref.backward(gy)
    xg = x.to(DEV).requires_grad_(True)
    y = ops.leave(ops.subsample(ops.enter(xg, dtype), 2))
    assert rel(y, y_ref) < 1e-6
    y.backward(gy.to(DEV).to(y.dtype))
    assert rel(xg.grad, xr.grad) < TOL[dtype]
    # maxpool 2x2
    xr = x.clone().requires_grad_(True)
    y_ref = F.max_pool2d(xr, 2, 2)
    gy = rnd("rsg2", *y_ref.shape)
    y_ref.backward(gy)
    xg = x.to(DEV).requires_grad_(True)
    y = ops.leave(ops.maxpool2(ops.enter(xg, dtype)))
    assert rel(y, y_ref) < 1e-6
    y.backward(gy.to(DEV).to(y.dtype))
    assert rel(xg.grad, xr.grad) < TOL[dtype]
    # up1 + nearest-up(low)
    low = rnd("rsl", N, C, H // 2, H // 2)
    if dtype == torch.bfloat16:
        low = low.bfloat16().float()
    xr, lr_ = x.clone().requires_grad_(True), low.clone().requires_grad_(True)
    y_ref = xr + F.interpolate(lr_, scale_factor=2)
    gy = rnd("rsg3", *y_ref.shape)
    y_ref.backward(gy)
    xg, lg = x.to(DEV).requires_grad_(True), low.to(DEV).requires_grad_(True)
    y = ops.leave(ops.upadd2(ops.enter(xg, dtype), ops.enter(lg, dtype)))
    assert rel(y, y_ref) < TOL[dtype]
    y.backward(gy.to(DEV).to(y.dtype))
    assert rel(xg.grad, xr.grad) < TOL[dtype] and rel(lg.grad, lr_.grad) < TOL[dtype]
    # cat
    a, b = rnd("cata", N, 128, 6, 6), rnd("catb", N, 64, 6, 6)
    ag, bg = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    y = ops.leave(ops.cat2(ops.enter(ag, torch.float32), ops.enter(bg, torch.float32)))
    assert rel(y, torch.cat((a, b), 1)) < 1e-7
    gy = rnd("catg", N, 192, 6, 6)
    y.backward(gy.to(DEV))
    assert rel(ag.grad, gy[:, :128]) < 1e-7 and rel(bg.grad, gy[:, 128:]) < 1e-7


def test_dropout_mask_and_stream():
    from xrface import ops
    x = rnd("dox", 4, 512, 7, 7).to(DEV).requires_grad_(True)
    buf = ops.enter(x, torch.float32)
    mask = (torch.from_numpy(G.uniform01("dom", buf.numel()).reshape(buf.shape)) < 0.5).to(DEV)
    y = ops.dropout(buf, 0.5, True, mask=mask)
    assert rel(y, buf.detach() * mask * 2.0) < 1e-7
    y.sum().backward()
    assert rel(x.grad.permute(0, 2, 3, 1), mask.float() * 2.0) < 1e-7
    y2 = ops.dropout(buf.detach(), 0.5, True, seed=123)
    keep = float((y2 != 0).float().mean())
    assert 0.47 < keep < 0.53
    assert torch.equal(y2, ops.dropout(buf.detach(), 0.5, True, seed=123))
    assert ops.dropout(buf, 0.5, False) is buf


def test_losses_against_reference_fixture():
    """Inputs/targets/expected values come from the reference's own loss modules (tests/golden/losses.npz)."""
    from tests.helpers import load_gold
    from xrface.loss.loss import CrossEntropyLoss2d, MSELoss_Landmark, MSELossFunc
    st = load_gold("losses.npz")
    for nm, mod in (("mse97", MSELossFunc()), ("landmark", MSELoss_Landmark()), ("nll2d", CrossEntropyLoss2d())):
        x = torch.from_numpy(st[nm + "/in"]).to(DEV).requires_grad_(True)
        t = torch.from_numpy(st[nm + "/target"]).to(DEV)
        loss = mod(x, t)
        (loss * 1.7).backward()
        assert abs(loss.item() - float(st[nm + "/loss"])) <= 1e-4 * abs(float(st[nm + "/loss"])), nm
        assert rel(x.grad, torch.from_numpy(st[nm + "/grad"]) * 1.7) < 1e-4, nm


def test_cross_entropy_rows():
    from xrface import ops
    x = rnd("cex", 16, 512)
    t = G.synth_labels(16, 512)
    xr = x.clone().requires_grad_(True)
    l_ref = F.cross_entropy(xr, t)
    l_ref.backward()
    xg = x.to(DEV).requires_grad_(True)
    l = ops.cross_entropy(xg, t.to(DEV))
    l.backward()
    assert abs(l.item() - l_ref.item()) < 1e-5 * abs(l_ref.item())
    assert rel(xg.grad, xr.grad) < 1e-4


def test_fused_optimizer_mask_semantics():
    """wd_mask bit 0 = weight decay on, bit 1 = skip the element (a parameter without gradient is left alone, as a stock
    optimizer skips .grad None -- including its weight decay)."""
    from xrface._lib import lib, ptr, stream
    m = 3000
    p0, g = rnd("mkp", m).to(DEV), rnd("mkg", m).to(DEV)
    mask = torch.ones(m, dtype=torch.uint8, device=DEV)
    mask[1000:2000] = 0
    mask[2000:] = 2
    g[2000:] = 0
    for kind in ("sgd", "rmsprop", "adam"):
        pa, pb = p0.clone(), p0.clone()
        sa, sb, va, vb = (torch.zeros(m, device=DEV) for _ in range(4))
        if kind == "sgd":
            lib.xr_sgd_step(ptr(pa), ptr(g), ptr(sa), m, 0.1, 0.9, 1e-2, ptr(mask), 1, stream())
            lib.xr_sgd_step(ptr(pb), ptr(g), ptr(sb), m, 0.1, 0.9, 1e-2, None, 1, stream())
            lib.xr_sgd_step(ptr(vb.copy_(p0)), ptr(g), ptr(va), m, 0.1, 0.9, 0.0, None, 1, stream())
        elif kind == "rmsprop":
            lib.xr_rmsprop_step(ptr(pa), ptr(g), ptr(sa), m, 5e-3, 0.99, 1e-8, 1e-2, ptr(mask), stream())
            lib.xr_rmsprop_step(ptr(pb), ptr(g), ptr(sb), m, 5e-3, 0.99, 1e-8, 1e-2, None, stream())
            lib.xr_rmsprop_step(ptr(vb.copy_(p0)), ptr(g), ptr(va), m, 5e-3, 0.99, 1e-8, 0.0, None, stream())
        else:
            s2a, s2b, s2c = (torch.zeros(m, device=DEV) for _ in range(3))
            lib.xr_adam_step(ptr(pa), ptr(g), ptr(sa), ptr(s2a), m, 1e-3, 0.5, 0.999, 1e-8, 1e-2, 1, None, 0, ptr(mask), stream())
            lib.xr_adam_step(ptr(pb), ptr(g), ptr(sb), ptr(s2b), m, 1e-3, 0.5, 0.999, 1e-8, 1e-2, 1, None, 0, None, stream())
            lib.xr_adam_step(ptr(vb.copy_(p0)), ptr(g), ptr(va), ptr(s2c), m, 1e-3, 0.5, 0.999, 1e-8, 0.0, 1, None, 0, None, stream())
        torch.cuda.synchronize()
        assert torch.equal(pa[:1000], pb[:1000]), kind                 # decay on: same as no mask
        assert torch.equal(pa[1000:2000], vb[1000:2000]), kind         # decay off: same as weight_decay = 0
        assert torch.equal(pa[2000:], p0[2000:]), kind                 # skipped: untouched ...
        assert not torch.equal(pb[2000:], p0[2000:]), kind             # ... while the unmasked update decays it


def test_fused_optimizers_match_torch_optim():
    from xrface._lib import lib, ptr, stream
    n = 10007
    p0, g = rnd("opp", n), rnd("opg", n)
    for kind in ("sgd", "rmsprop", "adam"):
        pr = p0.clone().requires_grad_(True)
        if kind == "sgd":
            opt = torch.optim.SGD([pr], lr=0.1, momentum=0.9, weight_decay=1e-3)
        elif kind == "rmsprop":
            opt = torch.optim.RMSprop([pr], lr=5e-3, alpha=0.99, weight_decay=1e-5)
        else:
            opt = torch.optim.Adam([pr], lr=1e-3, betas=(0.5, 0.999), weight_decay=1e-5)
        pg = p0.to(DEV)
        s1, s2 = torch.zeros(n, device=DEV), torch.zeros(n, device=DEV)
        for step in range(1, 4):
            pr.grad = g * step
            opt.step()
            gg = (g * step).to(DEV)
            if kind == "sgd":
                lib.xr_sgd_step(ptr(pg), ptr(gg), ptr(s1), n, 0.1, 0.9, 1e-3, None, int(step == 1), stream())
            elif kind == "rmsprop":
                lib.xr_rmsprop_step(ptr(pg), ptr(gg), ptr(s1), n, 5e-3, 0.99, 1e-8, 1e-5, None, stream())
            else:
                lib.xr_adam_step(ptr(pg), ptr(gg), ptr(s1), ptr(s2), n, 1e-3, 0.5, 0.999, 1e-8, 1e-5, step, None, 0, None, stream())
        assert rel(pg, pr) < 2e-6, kind


def test_pairdist_and_roc_against_reference_fixture():
    """dist vs the reference's numpy value (fp32 summation order differs: 1e-6 rel); the threshold sweep is
    integer-exact on a given dist array, and the whole calculate_roc matches the reference fixture."""
    from tests.helpers import load_gold
    from xrface.utils.utils import calculate_roc, pair_dist, roc_histograms
    from oracle import cpu_ref as R
    st = load_gold("roc.npz")
    p = int(st["p"])
    e1, e2, same = G.synth_pairs(p, 512, seed=0)
    dist = pair_dist(e1, e2).cpu().numpy()
    assert np.abs(dist - st["dist"]).max() / st["dist"].max() < 2e-6
    thresholds = np.arange(0, 12000, 3)
    hist = roc_histograms(torch.from_numpy(st["dist"]).to(DEV), same, st["fold_id"], thresholds, 10).cpu().numpy()
    for f in range(10):
        sel = st["fold_id"] == f
        tp, fp, tn, fn = R.confusion_at(thresholds, st["dist"][sel], same[sel])
        assert np.array_equal(np.cumsum(hist[f, 1])[:-1], tp) and np.array_equal(np.cumsum(hist[f, 0])[:-1], fp)
    # the K-fold sweep itself, on the device (xr_roc_sweep), fed with the REFERENCE's dist: every output bit-identical to what
    # the reference's calculate_roc returned (integer counts, fp64 rates, first-maximum argmax, fold-order means)
    from xrface.utils.utils import roc_sweep
    hist_d = roc_histograms(torch.from_numpy(st["dist"]).to(DEV), same, st["fold_id"], thresholds, 10)
    tpr_x, fpr_x, acc_x, best_x = roc_sweep(hist_d, len(thresholds), 10)
    assert np.array_equal(tpr_x, st["tpr"]) and np.array_equal(fpr_x, st["fpr"])
    assert acc_x.mean() == float(st["acc"]) and np.array_equal(thresholds[best_x].astype(np.float64), st["best"])
    assert np.array_equal(hist_d.cpu().numpy(), np.cumsum(hist, axis=2))        # hist is left as its prefix sums
    # end to end (distances from the HIP kernel differ from numpy's in the last bits: a few pairs cross a threshold)
    tpr, fpr, acc, best = calculate_roc(thresholds, e1, e2, same, nrof_folds=10, fold_id=st["fold_id"])
    assert np.abs(tpr - st["tpr"]).max() < 5e-3 and np.abs(fpr - st["fpr"]).max() < 5e-3
    assert abs(acc - float(st["acc"])) < 5e-3 and np.abs(best - st["best"]).max() <= 6
    # edge cases: a single fold has no train split -- the reference's KFold(n_splits=1) raises ValueError, so does this;
    # ragged: three pairs, two folds, one threshold, one fold without positives (tpr 0 by the reference's convention)
    with pytest.raises(ValueError):
        calculate_roc(np.array([1.0]), e1[:1], e2[:1], same[:1], nrof_folds=1, fold_id=np.zeros(1, np.int32))
    fid = np.array([0, 1, 1], np.int32)
    lab = np.array([True, False, False])
    t1, f1, a1, b1 = calculate_roc(np.array([1.0e9]), e1[:3], e2[:3], lab, nrof_folds=2, fold_id=fid)
    tr, fr, ar, br = R.calculate_roc(np.array([1.0e9]), e1[:3], e2[:3], lab, [(np.where(fid != f)[0], np.where(fid == f)[0]) for f in range(2)])
    assert t1.shape == (1,) and b1.shape == (2,)
    assert np.array_equal(t1, tr) and np.array_equal(f1, fr) and a1 == ar and np.array_equal(b1, br)


def test_calculate_roc_default_fold_count_matches_oracle():
    """calculate_roc with the reference's DEFAULT nrof_folds = 50 (utils/utils.py:26) and with 200 folds: the device sweep
    (xr_roc_sweep, up to 256 folds) against the oracle's numpy evaluation fed with the SAME distances and the same fold
    membership.  The default call draws its folds from sklearn's KFold(shuffle=True) on the global numpy RNG exactly like the
    reference -- pinned here with np.random.seed."""
    from sklearn.model_selection import KFold
    from xrface.utils.utils import calculate_roc, pair_dist
    from oracle import cpu_ref as R
    p = 3000
    e1, e2, same = G.synth_pairs(p, 512, seed=3)
    thresholds = np.arange(0, 12000, 3)
    dist = pair_dist(e1, e2).cpu().numpy()
    for folds_n in (50, 200):
        np.random.seed(77)
        folds = list(KFold(n_splits=folds_n, shuffle=True).split(np.arange(p)))
        np.random.seed(77)
        if folds_n == 50:
            tpr, fpr, acc, best = calculate_roc(thresholds, e1, e2, same)            # default argument, internal KFold draw
        else:
            tpr, fpr, acc, best = calculate_roc(thresholds, e1, e2, same, nrof_folds=folds_n)
        r_tpr, r_fpr, r_acc, r_best = R.calculate_roc_from_dist(thresholds, dist, same, folds)
        assert best.shape == (folds_n,)
        assert np.array_equal(tpr, r_tpr) and np.array_equal(fpr, r_fpr) and acc == r_acc and np.array_equal(best, r_best)
    with pytest.raises(RuntimeError):
        calculate_roc(thresholds, e1[:600], e2[:600], same[:600], nrof_folds=300)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_deterministic_mode_unit_step_repeats_bit_for_bit(dtype):
    """Two bottleneck_IR_SE units + a residual FSRNet block under xrface.set_deterministic(True): forward outputs, every parameter
    gradient and the input gradient are bit-identical between two runs (one reduction block per image-group, per-tile epilogue
    partial rows folded in order, single-group weight-gradient sums), and agree with the default mode to rounding.  In the default
    mode the same comparison is NOT exact (fp32 atomics order), which is what makes this test meaningful."""
    import xrface
    from xrface import ops
    from xrface.model.FSRnet import _Residual_Block
    from xrface.model.model_irse import bottleneck_IR_SE
    torch.manual_seed(11)
    mods = torch.nn.ModuleList([bottleneck_IR_SE(64, 64, 2), bottleneck_IR_SE(64, 128, 2), _Residual_Block(64)]).to(DEV).train()
    x = (rnd("detx", 6, 64, 40, 40) * 0.7).to(DEV)
    gy = rnd("detg", 6, 128, 10, 10).to(DEV)
    gz = rnd("detz", 6, 64, 40, 40).to(DEV)

    def run():
        for p_ in mods.parameters():
            p_.grad = None
        xg = x.clone().requires_grad_(True)
        buf = ops.enter(xg, dtype)
        z = mods[2].f(buf)
        y = mods[1].f(mods[0].f(z))
        out, mid = ops.leave(y), ops.leave(z)
        torch.autograd.backward([out, mid], [gy.to(out.dtype), gz.to(mid.dtype)])
        torch.cuda.synchronize()
        return [out.detach().float().clone(), xg.grad.clone()] + [p_.grad.clone() for p_ in mods.parameters()]

    xrface.set_deterministic(True)
    try:
        a, b = run(), run()
    finally:
        xrface.set_deterministic(False)
    for i, (u, v) in enumerate(zip(a, b)):
        assert torch.equal(u, v), f"deterministic mode: tensor #{i} differs between two runs"
    c = run()
    # (default mode: the direct 64-channel kernels and fused epilogues; deterministic mode: implicit GEMM + standalone passes -- two
    # valid bf16 evaluations that round at different places; measured up to 7e-2 of max-abs on the input gradient)
    tol = 1.5e-1 if dtype == torch.bfloat16 else 2e-3
    for i, (u, v) in enumerate(zip(a, c)):
        assert rel(u, v) < tol, (i, rel(u, v))


def test_arcface_head_against_fp64_restatement():
    """ArcFace is absent from the reference (parity unpinned): checked against the fp64 restatement in the oracle."""
    from oracle import cpu_ref as R
    from xrface.loss.loss import ArcFaceHead, CrossEntropyLoss
    n, d, classes = 32, 512, 1000
    emb = rnd("afe", n, d)
    w = rnd("afw", classes, d, scale=0.05)
    tgt = G.synth_labels(n, classes)
    e64, w64 = emb.double().requires_grad_(True), w.double().requires_grad_(True)
    ref = R.arcface_logits(e64, w64, tgt)
    l_ref = F.cross_entropy(ref, tgt)
    l_ref.backward()
    head = ArcFaceHead(d, classes).to(DEV)
    with torch.no_grad():
        head.weight.copy_(w.to(DEV))
    eg = emb.to(DEV).requires_grad_(True)
    logits = head(eg, tgt.to(DEV))
    loss = CrossEntropyLoss()(logits, tgt.to(DEV))
    loss.backward()
    assert rel(logits, ref) < 2e-4
    assert abs(loss.item() - l_ref.item()) < 1e-3 * abs(l_ref.item())
    assert rel(eg.grad, e64.grad) < 2e-3
    assert rel(head.weight.grad, w64.grad) < 2e-3


def test_mmd_against_fp64_restatement():
    """MMD is imported-but-undefined upstream (parity unpinned): checked against the oracle's fp64 restatement."""
    from oracle import cpu_ref as R
    from xrface.loss.loss import MMD
    n, d = 24, 512
    a, b = rnd("mma", n, d) * 0.3, rnd("mmb", n, d) * 0.3 + 0.05
    a64, b64 = a.double().requires_grad_(True), b.double().requires_grad_(True)
    ref = R.mmd_gaussian(a64, b64)
    ref.backward()
    ag, bg = a.to(DEV).requires_grad_(True), b.to(DEV).requires_grad_(True)
    val = MMD(ag, bg)
    (val * 2.0).backward()
    assert abs(val.item() - ref.item()) < 1e-4 * abs(ref.item()) + 1e-7
    assert rel(ag.grad, a64.grad * 2.0) < 1e-3 and rel(bg.grad, b64.grad * 2.0) < 1e-3


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("cin,depth,stride,hw", [(64, 64, 1, 20), (64, 128, 2, 28), (256, 256, 1, 14)])
def test_bn_backward_reduction_fused_into_conv_dgrad(cin, depth, stride, hw, dtype):
    """bottleneck_IR_SE with the first BatchNorm's backward reductions taken in the first convolution's dgrad epilogue
    (ops.BnLink) vs the separate reduction pass: same input gradient and parameter gradients; the link must really fire."""
    import copy
    import xrface
    from xrface import ops
    from xrface.model.model_irse import bottleneck_IR_SE
    xrface.set_compute_dtype(dtype)
    try:
        torch.manual_seed(5)
        blk0 = bottleneck_IR_SE(cin, depth, stride).to(DEV).train()
        x0 = rnd(f"bl{cin}{depth}", 8 if hw > 14 else 64, cin, hw, hw)
        res = {}
        taken = []
        orig_take = ops.BnLink.take

        def spy(self, dy):
            r = orig_take(self, dy)
            taken.append(r is not None)
            return r
        ops.BnLink.take = spy
        ops._cfg["ir_block"] = 0     # the op-level composition is under test here (the block-level path has its own test)
        for mode in (0, 1):
            ops._cfg["fuse_bn_reduce"] = mode
            blk = copy.deepcopy(blk0)
            x = x0.to(DEV).requires_grad_(True)
            out = blk(x)
            out.square().mean().backward()
            torch.cuda.synchronize()
            res[mode] = [x.grad.cpu()] + [p.grad.cpu() for p in blk.parameters()]
        assert taken == [False, True], taken
        # fp32: two different summation orders of the same reductions feeding an ill-conditioned small-batch BatchNorm backward
        # (bf16: the noise floor is set by the smallest gradient tensors -- max-abs ~5e-6 here -- on which two summation orders of
        # bf16-rounded data differ by up to 5.5e-2 of their own max-abs)
        tol = 1e-3 if dtype == torch.float32 else 8e-2
        for a, b in zip(res[1], res[0]):
            assert rel(a, b) < tol
    finally:
        ops.BnLink.take = orig_take
        ops._cfg["fuse_bn_reduce"] = 1
        ops._cfg["ir_block"] = 1
        xrface.set_compute_dtype(torch.float32)


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("c0,c1,stride,hw,n", [(64, 64, 1, 20, 8), (64, 128, 2, 28, 6), (256, 256, 1, 14, 32), (256, 512, 2, 14, 5)])
def test_chained_ir_se_units_share_statistics_and_backward_sums(c0, c1, stride, hw, n, dtype):
    """Two bottleneck_IR_SE units in a row (model_irse.py:69-91).  With ops.TailLink the first unit's tail pass also takes the
    statistics of its output (xr_affine_act_stats: the second unit's opening BatchNorm skips its statistics pass) and that
    BatchNorm's backward apply also takes the first tail's per-image sums (xr_affine_act_bwd_apply_red: the tail skips its reduce
    pass).  Same outputs, running statistics, input gradient and parameter gradients as the unchained passes; both links must
    really fire; an extra consumer of the unit output (a feature tap) must make the tail fall back to its own pass."""
    import copy
    import xrface
    from xrface import ops
    from xrface.model.model_irse import bottleneck_IR_SE
    from xrface.ops import enter, leave
    xrface.set_compute_dtype(dtype)
    orig_take, orig_stake = ops.TailLink.take, ops.StatsLink.take
    try:
        torch.manual_seed(11)
        u0 = [bottleneck_IR_SE(c0, c0, 1).to(DEV).train(), bottleneck_IR_SE(c0, c1, stride).to(DEV).train()]
        x0 = rnd(f"chain{c0}{c1}", n, c0, hw, hw)
        fired = {"red": [], "stats": []}

        def spy(self, dout):
            r = orig_take(self, dout)
            fired["red"].append(r is not None)
            return r

        def sspy(self, x):
            r = orig_stake(self, x)
            fired["stats"].append(r is not None)
            return r
        ops.TailLink.take, ops.StatsLink.take = spy, sspy
        res = {}
        for mode in (0, 1, 2):   # 2: chained, with a tap on the first unit's output
            ops._cfg["chain_units"] = 1 if mode else 0
            units = copy.deepcopy(u0)
            x = x0.to(DEV).requires_grad_(True)
            mid = units[0].f(enter(x))
            out = leave(units[1].f(mid))
            loss = out.square().mean()
            if mode == 2:
                loss = loss + 0.0 * leave(mid).sum()
            loss.backward()
            torch.cuda.synchronize()
            res[mode] = ([out.detach().cpu(), x.grad.cpu()] + [p.grad.cpu() for u in units for p in u.parameters()] +
                         [units[1].res_layer[0].running_mean.cpu(), units[1].res_layer[0].running_var.cpu()])
        # backward order is unit 1 then unit 0.  unchained: no TailLink exists; chained: the second tail has no successor, the
        # first takes; tapped: the delivery is refused
        assert fired["red"] == [False, True, False, False], fired
        # the output statistics are taken in both chained runs (+ the shortcut conv -> BatchNorm StatsLink of a widening unit)
        assert fired["stats"].count(True) == 2 + 3 * (c0 != c1), fired
        # The kernels themselves are checked bit for bit in test_fused_statistics_and_second_reduction_kernels; here the two paths
        # sum the same reductions in a different order (1e-7 on a statistic), and that is enough to flip the sign of a
        # pre-activation that sits within 1e-7 of the PReLU kink: ONE output-gradient element then changes its slope and the
        # difference spreads to a weight-gradient row and everything upstream (tools/probe/chain_noise.py shows the same between
        # two runs of the UNCHAINED path: L2 5e-4 / max-abs 1e-2).  Hence L2-relative bounds, not max-abs ones.
        l2 = lambda a, b: float((a.double() - b.double()).norm() / max(float(b.double().norm()), 1e-30))
        # Gradients that are analytically near zero (the shortcut BatchNorm's bias under this loss: sum of a normalised tensor) are
        # pure rounding noise in bf16: those are held to an absolute bound on the scale of the largest parameter gradient.
        tol, atol = (5e-3, 1e-5) if dtype == torch.float32 else (4e-2, 1e-3)
        rms = lambda t: float(t.double().norm()) / t.numel() ** 0.5
        top = max(rms(b) for b in res[0][2:-2])
        for k in (1, 2):
            for a, b in zip(res[k], res[0]):
                assert l2(a, b) < tol or rms(a - b) < atol * top, (k, tuple(a.shape), l2(a, b), rms(a - b), top)
    finally:
        ops.TailLink.take, ops.StatsLink.take = orig_take, orig_stake
        ops._cfg["chain_units"] = 1
        xrface.set_compute_dtype(torch.float32)


def test_lr_synthesis_and_heatmaps_match_pil_and_reference_fixture():
    """SURVEY 8f-3 on the device: xr_lr_synth == the PIL calls of FHN_loader.py:65-66 bit for bit (uint8), its normalised output ==
    ToTensor + Normalize bit for bit (float32); xr_heatmap == the reference's generate_hm to float32 rounding."""
    from tests.helpers import load_gold
    from oracle import cpu_ref as R
    from xrface.utils.loader_ops import gaussian_k, generate_hm, lr_from_hr
    st = load_gold("loader.npz")
    hr, lr_ref, scale = st["hr_u8"], st["lr_u8"], st["scale"]
    lr_u8, lr_n = lr_from_hr(hr, scale=scale)                      # one scale per image, as the loader draws them
    assert np.array_equal(lr_u8.cpu().numpy(), lr_ref)
    for i in range(len(hr)):
        assert torch.equal(lr_n[i].cpu(), R.to_tensor_normalize(lr_ref[i]))
    # every scale of scale_list on every image, against the restatement (pinned to PIL by the CPU suite)
    for sc in (2, 4, 8):
        got, _ = lr_from_hr(hr, scale=sc, normalize=False)
        for i in range(len(hr)):
            assert np.array_equal(got[i].cpu().numpy(), R.lr_from_hr_u8(hr[i], 128 // sc)), (sc, i)
    # ragged: non-square crop, odd sizes, low edge equal to one side (that pass is the identity), single image
    rng = np.random.RandomState(3)
    odd = (rng.rand(2, 45, 70, 3) * 255).astype(np.uint8)
    for low in (45, 17, 5):
        got, _ = lr_from_hr(odd, scale=128 / low, normalize=False)
        for i in range(2):
            assert np.array_equal(got[i].cpu().numpy(), R.lr_from_hr_u8(odd[i], low)), low
    with pytest.raises(ValueError):
        lr_from_hr(odd, scale=128 / 4)          # 70 / 4 > 16: beyond the tap tables
    with pytest.raises(ValueError):
        lr_from_hr(odd, scale=128 / 46)         # larger than the image
    # heat-maps: fp64 bumps, float32 running sum in landmark order; exp() of the device library vs numpy's may differ in the last
    # bit of the double, which can move a float32 sum by one ulp
    hm = generate_hm(112, 112, st["lm68"], s=2.0).cpu().numpy()
    assert hm.shape == (2, 112, 112) and np.abs(hm - st["hm68"]).max() <= 2.4e-7 * max(1.0, float(st["hm68"].max()))
    hm2 = generate_hm(112, 112, st["lm194"], s=1.3).cpu().numpy()
    assert np.abs(hm2 - st["hm194"]).max() <= 2.4e-7 * max(1.0, float(st["hm194"].max()))
    print(f"[heatmap] exact elements: {(hm == st['hm68']).mean():.4f} / {(hm2 == st['hm194']).mean():.4f}")
    g = gaussian_k(30.5, 40.25, 2.0, width=112, height=112).cpu().numpy()
    assert np.abs(g - R.gaussian_k(30.5, 40.25, 2.0, 112, 112).astype(np.float32)).max() <= 1.2e-7 and abs(g[40, 30] - np.exp(-(0.25 + 0.0625) / 8)) < 1e-6


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
@pytest.mark.parametrize("n,hw,c", [(32, 196, 256), (5, 49, 512), (6, 784, 128), (3, 400, 64), (2, 37, 24)])
def test_fused_statistics_and_second_reduction_kernels(n, hw, c, dtype):
    """xr_affine_act_stats == xr_affine_act + the sums of its stored output; xr_affine_act_bwd_apply_red == xr_affine_act_bwd_apply
    (same dx bit for bit) + the per-image sums of (dx, dx * y2) -- against float64 sums of the stored values."""
    from xrface._lib import lib, ptr, stream, dt
    g = torch.Generator().manual_seed(n * 1000 + c)
    mk = lambda *s: torch.randn(*s, generator=g).to(DEV)
    x, res, dy, y2, add = (mk(n, hw, c).to(dtype) for _ in range(5))
    cA, cB = mk(n, c), mk(n, c)
    # forward: per-image coefficients + residual, as the IR-SE tail launches it
    y_ref, y = torch.empty_like(x), torch.empty_like(x)
    stats = torch.zeros(2, n, c, device=DEV)
    lib.xr_affine_act(dt(x), ptr(x), ptr(cA), ptr(cB), ptr(res), None, 0, ptr(y_ref), n, hw, c, 1, stream())
    lib.xr_affine_act_stats(dt(x), ptr(x), ptr(cA), ptr(cB), ptr(res), None, 0, ptr(y), ptr(stats), n, hw, c, 1, stream())
    torch.cuda.synchronize()
    assert torch.equal(y, y_ref)
    yd = y_ref.double()
    for got, want, mag in ((stats[0], yd.sum(1), yd.abs().sum(1)), (stats[1], (yd * yd).sum(1), (yd * yd).sum(1))):
        assert float(((got.double() - want).abs() / mag).max()) < 2e-6
    # backward: one statistics group over the batch (coefficients per channel), shortcut gradient added, sums per image
    sc, sh, coef = mk(1, c), mk(1, c), mk(3, 1, c)
    dx_ref, dx = torch.empty_like(x), torch.empty_like(x)
    red2 = torch.zeros(2, n, c, device=DEV)
    lib.xr_affine_act_bwd_apply(dt(x), ptr(x), ptr(sc), ptr(sh), None, None, 0, ptr(dy), ptr(coef), ptr(dx_ref), None, 1, n * hw, c, 1,
                                ptr(add), stream())
    lib.xr_affine_act_bwd_apply_red(dt(x), ptr(x), ptr(sc), ptr(sh), None, None, 0, ptr(dy), ptr(coef), ptr(dx), None, n, hw, c,
                                    ptr(add), ptr(y2), ptr(red2), stream())
    torch.cuda.synchronize()
    assert torch.equal(dx, dx_ref)
    dd, wd = dx_ref.double(), y2.double()
    for got, want, mag in ((red2[0], dd.sum(1), dd.abs().sum(1)), (red2[1], (dd * wd).sum(1), (dd * wd).abs().sum(1))):
        assert float(((got.double() - want).abs() / mag).max()) < 2e-6


@pytest.mark.parametrize("c,hw,n", [(128, 28, 6), (256, 14, 32), (512, 7, 16)])
def test_ir_se_unit_block_level_path_equals_op_level_path(c, hw, n):
    """xr_ir_block_fwd / xr_ir_block_bwd (one C call per identity-shortcut bottleneck_IR_SE unit each way, ops._IrSeUnit) against the
    op-level composition it replaces (BN1 + BnLink, conv -> PReLU -> conv, BatchNorm + SE tail + TailLink): three chained units in
    bf16 -- forward output, input gradient, every parameter gradient and the BatchNorm running statistics.  Same kernels in the same
    order, so the two differ only by the order of the fp32 atomics (times bf16 re-rounding); the yardstick is the fp32 parity mode
    on the same weights: the block-level path must sit as close to it as the op-level path does (some of these gradients -- the SE
    fc1 weight -- are ill-conditioned in bf16: 10-25 % off the fp32 value on EITHER path).  The block path must really have been
    taken, and the links between block-level and op-level units (statistics forward, tail sums backward) must work both ways."""
    import copy
    import xrface
    from xrface import ops
    from xrface.model.model_irse import bottleneck_IR_SE
    calls = {"fwd": 0}
    orig = ops._IrSeUnit.forward

    def spy(ctx, *a):
        calls["fwd"] += 1
        return orig(ctx, *a)
    try:
        ops._IrSeUnit.forward = staticmethod(spy)
        torch.manual_seed(21)
        units0 = torch.nn.ModuleList([bottleneck_IR_SE(c, c, 1) for _ in range(3)]).to(DEV).train()
        x0 = rnd(f"irb{c}", n, c, hw, hw)
        res = {}
        # modes: fp32 parity mode (op-level: the reference) / bf16 op-level / bf16 block-level / bf16 mixed (unit 1 op-level)
        for mode, dtype in (("f32", torch.float32), ("ops", torch.bfloat16), ("block", torch.bfloat16), ("mixed", torch.bfloat16)):
            xrface.set_compute_dtype(dtype)
            units = copy.deepcopy(units0)
            x = x0.to(DEV).requires_grad_(True)
            y = ops.enter(x, dtype)
            for i, u in enumerate(units):
                ops._cfg["ir_block"] = int(mode == "block" or (mode == "mixed" and i != 1))
                y = u.f(y)
            out = ops.leave(y)
            out.float().square().mean().backward()
            torch.cuda.synchronize()
            stats = [b.clone() for nme, b in units.named_buffers() if "running" in nme]
            res[mode] = [out.detach().float().cpu(), x.grad.cpu()] + [p.grad.float().cpu() for p in units.parameters()] + [t.cpu() for t in stats]
        assert calls["fwd"] == 3 + 2, calls
        for other in ("block", "mixed"):
            for i, (a, o, r) in enumerate(zip(res[other], res["ops"], res["f32"])):
                e_blk, e_ops = rel(a, r), rel(o, r)
                assert e_blk < max(5e-2, 2.5 * e_ops), (other, i, e_blk, e_ops)   # (two samples of a noisy quantity each)
        # forward outputs and running statistics do not depend on the atomics' order beyond rounding
        assert rel(res["block"][0], res["ops"][0]) < 1e-2
        for a, b in zip(res["block"][-12:], res["ops"][-12:]):
            assert rel(a, b) < 3e-3     # (statistics of bf16 tensors that already differ in their last bits)
    finally:
        ops._IrSeUnit.forward = orig
        ops._cfg["ir_block"] = 1
        xrface.set_compute_dtype(torch.float32)


def test_input_layer_offers_statistics_to_the_first_unit():
    """Backbone.f_input (conv -> BatchNorm -> PReLU, model_irse.py:141-143) takes the statistics of its output in the same pass;
    the first unit's opening BatchNorm must pick them up and produce the running statistics / outputs of the separate pass."""
    import copy
    import xrface
    from xrface import ops
    from xrface.model.model_irse import Backbone, bottleneck_IR_SE
    from xrface.ops import enter, leave
    orig = ops.StatsLink.take
    try:
        torch.manual_seed(3)
        net0 = Backbone([112, 112], 50, 'ir_se')
        il0, u0 = net0.input_layer.to(DEV).train(), net0.body[0].to(DEV).train()
        x = rnd("inl", 4, 3, 24, 24).to(DEV)
        res, hits = {}, []

        def spy(self, t):
            r = orig(self, t)
            hits.append(r is not None)
            return r
        ops.StatsLink.take = spy
        for mode in (0, 1):
            ops._cfg["chain_units"] = mode
            net = copy.deepcopy(net0)
            net.input_layer, net.body = copy.deepcopy(il0), torch.nn.Sequential(copy.deepcopy(u0))
            y = net.f_input(enter(x))
            assert (getattr(y, "_xr_tail", None) is not None) == bool(mode)
            out = leave(net.body[0].f(y))
            torch.cuda.synchronize()
            bn = net.body[0].res_layer[0]
            res[mode] = [out.detach().cpu(), bn.running_mean.cpu(), bn.running_var.cpu()]
        # takes per run: the input layer's conv -> BatchNorm link, then (chained only) the offered statistics
        assert hits == [True, True, True], hits
        for a, b in zip(res[1], res[0]):
            assert rel(a, b) < 1e-5
    finally:
        ops.StatsLink.take = orig
        ops._cfg["chain_units"] = 1


WGRAD8_CASES = [
    # N, C, H, W, K, R, stride, pad, split
    (8, 256, 14, 14, 256, 3, 1, 1, 14),
    (4, 256, 14, 14, 512, 3, 1, 1, 7),
    (8, 512, 7, 7, 512, 3, 1, 1, 3),
    (4, 256, 28, 28, 256, 3, 2, 1, 5),
    (3, 128, 9, 9, 192, 3, 1, 1, 2),     # C % 256 != 0: not eligible even when forced -- must fall back to the 4-wave kernel
    (2, 72, 11, 13, 136, 3, 1, 1, 3),    # (same)
    (4, 256, 14, 14, 512, 1, 1, 0, 4),   # 1x1
    (1, 256, 6, 6, 256, 3, 1, 1, 1),     # a single ragged stage
    (2, 512, 10, 10, 384, 3, 1, 1, 2),   # three row tiles, two taps' worth of column tiles per tap
    (3, 256, 14, 14, 264, 3, 1, 1, 3),   # ragged last row tile (264 = 2 x 128 + 8)
]


@pytest.mark.parametrize("case", WGRAD8_CASES)
def test_wgrad_8wave_ring_kernel(case):
    """xr_wgrad8.hip (8 waves, 128 x 256 tile, three-stage LDS ring, counted vmcnt) forced on (xr_tune knob 13 = 2) against the
    4-wave sliced kernel on the same operands through the C ABI: both cut the pixels into the same slices and add the same
    16-pixel MFMA steps in the same order, so every slab must agree bit for bit; then against the fp32 reference; twice (race
    screen: the ring's only ordering is one s_barrier per stage)."""
    from xrface import ops
    from xrface._lib import lib, ptr, stream, dt
    N, C, H, W, K, R, stride, pad, split = case
    Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - R) // stride + 1
    x = rnd(f"w8x{case}", N, C, H, W).bfloat16()
    gy = rnd(f"w8g{case}", N, K, Ho, Wo).bfloat16()
    Cp, Kp = ops.r8(C), ops.r8(K)
    kg = ops.kg_of(R * R, Cp)
    xb = torch.zeros(N, H, W, Cp, dtype=torch.bfloat16); xb[..., :C] = x.permute(0, 2, 3, 1)
    gb = torch.zeros(N, Ho, Wo, Kp, dtype=torch.bfloat16); gb[..., :K] = gy.permute(0, 2, 3, 1)
    xb, gb = xb.to(DEV), gb.to(DEV)
    out = {}
    try:
        for knob in (0, 2, 2):
            lib.xr_tune(13, knob)
            slabs = torch.full((split, K, kg), float("nan"), device=DEV)
            n = lib.xr_conv_wgrad(dt(xb), ptr(xb), ptr(gb), ptr(slabs), N, H, W, Cp, Ho, Wo, K, R, R, stride, pad, 0, Kp, kg, split,
                                  stream())
            torch.cuda.synchronize()
            out.setdefault(knob, []).append((n, slabs[:n].cpu()))
    finally:
        lib.xr_tune(13, 1)
    (n0, s0), (n8, s8), (_, s8b) = out[0][0], out[2][0], out[2][1]
    assert n0 == n8 and not torch.isnan(s8).any()
    assert torch.equal(s8, s8b)
    assert torch.equal(s8, s0)
    dw = s8.sum(0)[:, :R * R * Cp].reshape(K, R * R, Cp)[:, :, :C].permute(0, 2, 1).reshape(K, C, R, R)
    ref = torch.nn.grad.conv2d_weight(x.float(), (K, C, R, R), gy.float(), stride=stride, padding=pad)
    assert rel(dw, ref) < 2e-5
