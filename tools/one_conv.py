#!/usr/bin/env python
"""Run one conv shape (fwd, dgrad, wgrad) a few times -- target for rocprofv3 --pmc."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
from xrface import ops
from xrface._lib import lib, ptr, stream, dt
dev = torch.device("cuda:0")
N, C, K, H, st = (int(v) for v in os.environ.get("SHAPE", "256,256,256,14,1").split(","))
dtype = torch.bfloat16
Ho = (H + 2 - 3) // st + 1
x = torch.randn(N, H, H, C, device=dev).to(dtype)
w = torch.randn(K, C, 3, 3, device=dev) * 0.05
dy = torch.randn(N, Ho, Ho, K, device=dev).to(dtype)
y = torch.empty_like(dy); dx = torch.empty_like(x)
pk, kg = ops._packed(w, "fwd", dtype, K, 1, 9, C, C, C * 9, 0, 1, 9)
pkd, kgd = ops._packed(w, "dgrad", dtype, C, 1, 9, K, K, 9, 0, 1, C * 9)
split = ops._wgrad_split(N * Ho * Ho, K, kg)
slab = torch.zeros(split, K, kg, device=dev)
for _ in range(int(os.environ.get("REPS", 3))):
    lib.xr_conv_igemm(dt(x), ptr(x), ptr(pk), None, ptr(y), N, H, H, C, Ho, Ho, K, 3, 3, st, 1, 0, kg, K, None, 0, None, None, None, 1, None, None, None, stream())
    lib.xr_conv_igemm(dt(x), ptr(dy), ptr(pkd), None, ptr(dx), N, Ho, Ho, K, H, H, C, 3, 3, st, 1, 1, kgd, C, None, 0, None, None, None, 1, None, None, None, stream())
    lib.xr_conv_wgrad(dt(x), ptr(x), ptr(dy), ptr(slab), N, H, H, C, Ho, Ho, K, 3, 3, st, 1, 0, K, kg, split, stream())
torch.cuda.synchronize()
