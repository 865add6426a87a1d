#!/usr/bin/env python
"""Experiment: 64 -> 3 and 3 -> 64 3x3 convolutions at 112 x 112 (FSRNet image heads / entry convolutions), 128 x 32 vs 128 x 64 tiles
of the narrow implicit-GEMM kernel (knob 15 = 1 restores the 64-column tile)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "cross-resolution-face-recognition_amd"))
import torch
from xrface import ops
from xrface._lib import lib, ptr, stream
dev = torch.device("cuda:0")
N, H = int(os.environ.get("N", 256)), 112


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


x = torch.randn(N, H, H, 64, device=dev).bfloat16()
w = torch.randn(3, 64, 3, 3, device=dev) * 0.05
y = torch.empty(N, H, H, 8, device=dev, dtype=torch.bfloat16)
pk, kg = ops._packed(w, "fwd", torch.bfloat16, 3, 1, 9, 64, 64, 576, 0, 1, 9)
fwd = lambda: lib.xr_conv_igemm(0, ptr(x), ptr(pk), None, ptr(y), N, H, H, 64, H, H, 3, 3, 3, 1, 1, 0, kg, 8, None, 0, None, None, None, 1, None, None, None, stream())
ref = None
for knob in (1, 0):
    lib.xr_tune(15, knob)
    us = timeit(fwd)
    torch.cuda.synchronize()
    if ref is None:
        ref = y.clone()
    print(f"64->3 fwd  tile 128x{'64' if knob else '32'}: {us:.0f} us  (same result: {torch.equal(ref[..., :3], y[..., :3])})", flush=True)
lib.xr_tune(15, 0)
