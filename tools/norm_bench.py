#!/usr/bin/env python
"""Micro-benchmark of the norm/activation family on the IR-SE-50 activation shapes (batch 256, bf16): achieved GB/s of the
statistics pass, forward apply, backward reduce and backward apply, per tuning variant (VARIANTS="8=2048,9=2048;8=512,...")."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
from xrface._lib import lib, ptr, stream, dt

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 256))
SHAPES = [(56, 64), (28, 128), (14, 256), (7, 512)]


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


variants = [v for v in os.environ.get("VARIANTS", "8=2048,9=2048").split(";")]
for H, C in SHAPES:
    M = N * H * H
    x = torch.randn(M, C, device=dev).bfloat16()
    dy = torch.randn(M, C, device=dev).bfloat16()
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    nbytes = M * C * 2
    for pg in (32, N):
        rows = M // pg
        sums = torch.zeros(3, pg, C, device=dev)
        sc = torch.ones(pg, C, device=dev); sh = torch.zeros(pg, C, device=dev)
        coef = torch.ones(3, pg, C, device=dev)
        line = f"{H:3d}x{H:<3d} C={C:3d} G={pg:3d} "
        for v in variants:
            for kv in v.split(","):
                k, val = kv.split("=")
                lib.xr_tune(int(k), int(val))
            t_st = timeit(lambda: lib.xr_group_stats(dt(x), ptr(x), ptr(sums), pg, rows, C, stream()))
            t_fw = timeit(lambda: lib.xr_affine_act(dt(x), ptr(x), ptr(sc), ptr(sh), None, None, 0, ptr(y), pg, rows, C, 1, stream()))
            t_br = timeit(lambda: lib.xr_affine_act_bwd_reduce(dt(x), ptr(x), ptr(sc), ptr(sh), None, None, 0, ptr(dy), ptr(sums), pg, rows, C, 1, stream()))
            t_ba = timeit(lambda: lib.xr_affine_act_bwd_apply(dt(x), ptr(x), ptr(sc), ptr(sh), None, None, 0, ptr(dy), ptr(coef), ptr(dx), None, pg, rows, C, 1, None, stream()))
            line += f" | stats {nbytes / t_st / 1e9:5.0f} fwd {2 * nbytes / t_fw / 1e9:5.0f} bred {2 * nbytes / t_br / 1e9:5.0f} bapp {3 * nbytes / t_ba / 1e9:5.0f} GB/s"
        print(line)

# chained IR-SE units (ops.TailLink): separate passes vs the fused ones, in microseconds.  VARIANTS_CHAIN="8=1024,9=1024;..."
print("\nchained units: tail apply + next statistics  |  opening-norm apply (with shortcut gradient) + tail reduce")
for H, C in SHAPES:
    M = N * H * H
    HW = H * H
    x = torch.randn(M, C, device=dev).bfloat16()
    dy = torch.randn(M, C, device=dev).bfloat16()
    y2 = torch.randn(M, C, device=dev).bfloat16()
    sc_t = torch.randn(M, C, device=dev).bfloat16()
    y = torch.empty_like(x)
    dx = torch.empty_like(x)
    cA, cB = torch.ones(N, C, device=dev), torch.zeros(N, C, device=dev)
    sc1, sh1, coef1 = torch.ones(1, C, device=dev), torch.zeros(1, C, device=dev), torch.ones(3, 1, C, device=dev)
    s2, s3 = torch.zeros(2, N, C, device=dev), torch.zeros(3, N, C, device=dev)
    s32 = torch.zeros(2, 32, C, device=dev)
    for v in os.environ.get("VARIANTS_CHAIN", "8=1024,9=1024;8=2048,9=2048;8=4096,9=4096").split(";"):
        for kv in v.split(","):
            k, val = kv.split("=")
            lib.xr_tune(int(k), int(val))
        t_fw = timeit(lambda: lib.xr_affine_act(dt(x), ptr(x), ptr(cA), ptr(cB), ptr(sc_t), None, 0, ptr(y), N, HW, C, 1, stream()))
        t_st = timeit(lambda: lib.xr_group_stats(dt(y), ptr(y), ptr(s32), 32, M // 32, C, stream()))
        t_fs = timeit(lambda: lib.xr_affine_act_stats(dt(x), ptr(x), ptr(cA), ptr(cB), ptr(sc_t), None, 0, ptr(y), ptr(s2), N, HW, C, 1, stream()))
        t_ba = timeit(lambda: lib.xr_affine_act_bwd_apply(dt(x), ptr(x), ptr(sc1), ptr(sh1), None, None, 0, ptr(dy), ptr(coef1), ptr(dx), None, 1, M, C, 1, ptr(sc_t), stream()))
        t_br = timeit(lambda: lib.xr_affine_act_bwd_reduce(dt(y2), ptr(y2), None, None, None, None, 0, ptr(dx), ptr(s3), N, HW, C, 1, stream()))
        t_bf = timeit(lambda: lib.xr_affine_act_bwd_apply_red(dt(x), ptr(x), ptr(sc1), ptr(sh1), None, None, 0, ptr(dy), ptr(coef1), ptr(dx), None, N, HW, C, ptr(sc_t), ptr(y2), ptr(s2), stream()))
        print(f"{H:3d}x{H:<3d} C={C:3d} [{v:16s}]  fwd {t_fw * 1e6:6.1f} + stats {t_st * 1e6:6.1f} = {(t_fw + t_st) * 1e6:6.1f} us -> fused {t_fs * 1e6:6.1f} us"
              f"  |  apply {t_ba * 1e6:6.1f} + reduce {t_br * 1e6:6.1f} = {(t_ba + t_br) * 1e6:6.1f} us -> fused {t_bf * 1e6:6.1f} us")

# the per-unit glue of the IR-SE tail (latency chains on the critical path of every unit): xr_bnse_fwd, xr_bnse_bwd
print("\nIR-SE tail glue (microseconds per call)")
for H, C in SHAPES:
    Cr, HW = C // 16, H * H
    f = lambda *s: torch.randn(*s, device=dev)
    sum_y, a, b, w1, w2 = f(N, C), f(C), f(C), f(Cr, C) * 0.1, f(C, Cr) * 0.1
    pooled, hidden, s_, cA, cB = f(N, C), f(N, Cr), f(N, C), f(N, C), f(N, C)
    S1, S2, gamma, mean, invstd = f(N, C), f(N, C), f(C), f(C), f(C).abs() + 0.5
    dpre2, dhid, dp, coef, dg, db = f(N, C), f(N, Cr), f(N, C), f(3, N, C), torch.zeros(C, device=dev), torch.zeros(C, device=dev)
    t_f = timeit(lambda: lib.xr_bnse_fwd(ptr(sum_y), ptr(a), ptr(b), ptr(w1), ptr(w2), ptr(pooled), ptr(hidden), ptr(s_), ptr(cA), ptr(cB), N, C, Cr, HW, stream()), reps=50)
    t_b = timeit(lambda: lib.xr_bnse_bwd(ptr(S1), ptr(S2), ptr(sum_y), ptr(a), ptr(b), ptr(w1), ptr(w2), ptr(hidden), ptr(s_), ptr(gamma), ptr(mean), ptr(invstd), ptr(dpre2), ptr(dhid), ptr(dp), ptr(coef), ptr(dg), ptr(db), N, C, Cr, HW, 1, stream()), reps=50)
    print(f"{H:3d}x{H:<3d} C={C:3d} Cr={Cr:2d}:  fwd {t_f * 1e6:6.1f} us   bwd (excite + coeffs) {t_b * 1e6:6.1f} us")
