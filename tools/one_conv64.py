#!/usr/bin/env python
"""Run the direct 64-channel kernels (fwd / dgrad with fusions, weight gradient) on one shape a few times -- target for
rocprofv3 --pmc.  SHAPE=N,H (default 128,112)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
from xrface import ops
from xrface._lib import lib, ptr, stream
dev = torch.device("cuda:0")
N, H = (int(v) for v in os.environ.get("SHAPE", "128,112").split(","))
x = torch.randn(N, H, H, 64, device=dev).bfloat16()
dy = torch.randn(N, H, H, 64, device=dev).bfloat16()
x2 = torch.randn(N, H, H, 64, device=dev).bfloat16()
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
y = torch.empty_like(x)
pk, _ = ops._packed(w, "fwd", torch.bfloat16, 64, 1, 9, 64, 64, 576, 0, 1, 9)
pkd, _ = ops._packed(w, "dgrad", torch.bfloat16, 64, 1, 9, 64, 64, 9, 0, 1, 576)
sc = torch.rand(N, 64, device=dev) + 0.5; sh = torch.randn(N, 64, device=dev); al = torch.rand(64, device=dev)
stats = torch.zeros(2, N, 64, device=dev)
slabs = torch.empty(256, 64, 576, device=dev)
red = torch.zeros(3, N, 64, device=dev)
for _ in range(int(os.environ.get("REPS", 3))):
    lib.xr_conv64_direct(ptr(x), ptr(pk), None, ptr(y), N, H, H, 0, None, None, None, ptr(stats), None, stream())               # conv1 fwd
    lib.xr_conv64_direct(ptr(x), ptr(pk), None, ptr(y), N, H, H, 0, ptr(sc), ptr(sh), ptr(al), ptr(stats), None, stream())     # conv2 fwd
    lib.xr_conv64_direct(ptr(dy), ptr(pkd), None, ptr(y), N, H, H, 1, None, None, None, None, None, stream())                  # dgrad
    lib.xr_conv64_direct(ptr(dy), ptr(pkd), None, ptr(y), N, H, H, 1, None, None, None, None, ptr(x), stream())                # dgrad + add
    lib.xr_conv64_direct_bwdred(ptr(dy), ptr(pkd), ptr(y), N, H, H, 1, ptr(x), ptr(sc), ptr(sh), ptr(al), ptr(red), stream())  # dgrad + IN-bwd sums
    lib.xr_conv64_direct_tailred(ptr(dy), ptr(pkd), ptr(y), N, H, H, 1, ptr(x), ptr(x2), ptr(x), ptr(sc), ptr(sh), ptr(al), ptr(red), stream())  # chained tail
    lib.xr_conv64_wgrad(ptr(x), ptr(dy), ptr(slabs), N, H, H, 256, None, None, None, stream())
    lib.xr_conv64_wgrad(ptr(x), ptr(dy), ptr(slabs), N, H, H, 256, ptr(sc), ptr(sh), ptr(al), stream())
torch.cuda.synchronize()
