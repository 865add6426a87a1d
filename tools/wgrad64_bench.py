#!/usr/bin/env python
"""Micro-benchmark of the direct 64->64 3x3 weight gradient (xr_conv64_wgrad, + slab sum) against the sliced implicit GEMM
(xr_conv_wgrad) on the FSRNet / stage-1 shapes; TFLOP/s of the kernel alone and with xr_unpack_wgrad."""
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


for N, H in ((256, 112), (128, 112), (256, 56), (64, 112)):
    x = torch.randn(N, H, H, 64, device=dev).bfloat16()
    dy = torch.randn(N, H, H, 64, device=dev).bfloat16()
    sc = torch.rand(N, 64, device=dev) + 0.5; sh = torch.randn(N, 64, device=dev); al = torch.rand(64, device=dev)
    dw = torch.zeros(64, 64, 3, 3, device=dev)
    slabs = torch.empty(256, 64, 576, device=dev)
    fl = 2.0 * N * H * H * 64 * 64 * 9
    ns = [0]
    def direct(xf=False):
        ns[0] = lib.xr_conv64_wgrad(ptr(x), ptr(dy), ptr(slabs), N, H, H, 256, ptr(sc) if xf else None, ptr(sh) if xf else None,
                                    ptr(al) if xf else None, stream())
    unpack = lambda: lib.xr_unpack_wgrad(ptr(slabs), ptr(dw), 64, 1, 9, 64, 64, 576, 576, 0, 1, 9, 1, ns[0], stream())
    a = timeit(direct)
    b = timeit(lambda: direct(True))
    c = timeit(lambda: (direct(), unpack()))
    split = ops._wgrad_split(N * H * H, 64, 576)
    slabs2 = torch.empty(split, 64, 576, device=dev)
    ns2 = [0]
    def sliced():
        ns2[0] = lib.xr_conv_wgrad(0, ptr(x), ptr(dy), ptr(slabs2), N, H, H, 64, H, H, 64, 3, 3, 1, 1, 0, 64, 576, split, stream())
    d = timeit(sliced)
    e = timeit(lambda: (sliced(), lib.xr_unpack_wgrad(ptr(slabs2), ptr(dw), 64, 1, 9, 64, 64, 576, 576, 0, 1, 9, 1, ns2[0], stream())))
    print(f"N={N} {H}x{H} {fl / 1e9:6.1f} GF: direct {fl / a / 1e9:6.0f} TF/s ({a * 1e3:.0f} us, {ns[0]} slabs)  +transform {fl / b / 1e9:6.0f}"
          f"  +unpack {fl / c / 1e9:6.0f} ({c * 1e3:.0f} us) | sliced {fl / d / 1e9:6.0f} TF/s ({d * 1e3:.0f} us, {ns2[0]} slabs) +unpack "
          f"{fl / e / 1e9:6.0f} ({e * 1e3:.0f} us)", flush=True)
