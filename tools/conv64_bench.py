#!/usr/bin/env python
"""Micro-benchmark of the direct 64->64 3x3 convolution (xr_conv64_direct) against the implicit-GEMM kernel on the FSRNet /
stage-1 shapes; prints TFLOP/s fwd / dgrad, plain and with the fused on-load transform + statistics."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
from xrface import ops
from xrface._lib import lib, ptr, stream

dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for N, H in ((256, 112), (128, 112), (256, 56), (256, 28)):
    x = torch.randn(N, H, H, 64, device=dev).bfloat16()
    w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    y = torch.empty_like(x)
    pk, _ = ops._packed(w, "fwd", torch.bfloat16, 64, 1, 9, 64, 64, 576, 0, 1, 9)
    pkd, _ = ops._packed(w, "dgrad", torch.bfloat16, 64, 1, 9, 64, 64, 9, 0, 1, 576)
    sc = torch.rand(N, 64, device=dev) + 0.5; sh = torch.randn(N, 64, device=dev); al = torch.rand(64, device=dev)
    stats = torch.zeros(2, N, 64, device=dev)
    fl = 2.0 * N * H * H * 64 * 64 * 9
    row = f"N={N} {H}x{H} {fl / 1e9:6.1f} GF:"
    for wide in (0,):
        lib.xr_tune(13, wide)
        a = timeit(lambda: lib.xr_conv64_direct(ptr(x), ptr(pk), None, ptr(y), N, H, H, 0, None, None, None, None, None, stream()))
        b = timeit(lambda: lib.xr_conv64_direct(ptr(x), ptr(pkd), None, ptr(y), N, H, H, 1, None, None, None, None, None, stream()))
        c = timeit(lambda: lib.xr_conv64_direct(ptr(x), ptr(pk), None, ptr(y), N, H, H, 0, ptr(sc), ptr(sh), ptr(al), ptr(stats), None, stream()))
        row += f"  direct(w{wide}) fwd {fl / a / 1e9:6.0f} dgrad {fl / b / 1e9:6.0f} fwd+norm+stats {fl / c / 1e9:6.0f} TF/s ({a * 1e3:.0f} us)"
    lib.xr_tune(13, 0)
    d = timeit(lambda: lib.xr_conv_igemm(0, ptr(x), ptr(pk), None, ptr(y), N, H, H, 64, H, H, 64, 3, 3, 1, 1, 0, 576, 64, None, 0, None, None, None, 1, None, None, None, stream()))
    e = timeit(lambda: lib.xr_conv_igemm(0, ptr(x), ptr(pkd), None, ptr(y), N, H, H, 64, H, H, 64, 3, 3, 1, 1, 1, 576, 64, None, 0, None, None, None, 1, None, None, None, stream()))
    row += f"  igemm fwd {fl / d / 1e9:6.0f} dgrad {fl / e / 1e9:6.0f} TF/s ({d * 1e3:.0f} us)"
    print(row, flush=True)
