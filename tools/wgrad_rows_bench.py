#!/usr/bin/env python
"""Micro-benchmark of the row-walking weight gradient (xr_conv_wgrad_rows) against the sliced implicit GEMM (xr_conv_wgrad) on the
IR-SE-50 body shapes at batch 256; TFLOP/s of the kernel alone (slab sums excluded on both sides)."""
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


N = int(os.environ.get("N", 256))
for C, K, H, s in ((256, 256, 14, 1), (128, 128, 28, 1), (64, 64, 56, 1), (64, 128, 56, 1), (128, 256, 28, 1), (256, 512, 14, 1),
                   (64, 64, 112, 2), (128, 128, 56, 2), (256, 256, 28, 2), (64, 64, 112, 1)):
    Ho = H // s
    x = torch.randn(N, H, H, C, device=dev).bfloat16()
    dy = torch.randn(N, Ho, Ho, K, device=dev).bfloat16()
    kg = 9 * C
    cap = max(1, 256 // ((K // 64) * (C // 64)))
    slabs = torch.empty(cap, K, kg, device=dev)
    fl = 2.0 * N * Ho * Ho * K * C * 9
    ns = [0]
    def rows():
        ns[0] = lib.xr_conv_wgrad_rows(ptr(x), ptr(dy), ptr(slabs), N, H, H, C, K, s, cap, stream())
    a = timeit(rows)
    split = ops._wgrad_split(N * Ho * Ho, K, kg)
    slabs2 = torch.empty(split, K, kg, device=dev)
    ns2 = [0]
    def sliced():
        ns2[0] = lib.xr_conv_wgrad(0, ptr(x), ptr(dy), ptr(slabs2), N, H, H, C, Ho, Ho, K, 3, 3, s, 1, 0, K, kg, split, stream())
    b = timeit(sliced)
    print(f"{C:3d}->{K:3d} @{H}x{H} s{s} {fl / 1e9:6.1f} GF: rows {fl / a / 1e9:6.0f} TF/s ({a * 1e3:.0f} us, {ns[0]} slabs) | sliced "
          f"{fl / b / 1e9:6.0f} TF/s ({b * 1e3:.0f} us, {ns2[0]} slabs)", flush=True)
