#!/usr/bin/env python
"""Isolated timing of the 3x3 / stride-2 input gradient: class-mode strided gather (xr_conv_igemm, transposed) vs the dense 2x2-window
GEMM with depth-to-space epilogue (xr_conv_dgrad_s2), IR-SE-50 stage-opening shapes at batch 256."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
from xrface import ops
from xrface._lib import lib, ptr, stream
dev = torch.device("cuda:0")


def timeit(fn, reps=10):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


N = 256
for C, H in ((64, 112), (128, 56), (256, 28), (512, 14)):
    Ho = H // 2
    w = torch.randn(C, C, 3, 3, device=dev) * 0.05
    dy = torch.randn(N, Ho, Ho, C, device=dev).bfloat16()
    dx = torch.empty(N, H, H, C, device=dev, dtype=torch.bfloat16); dx2 = torch.empty_like(dx)
    pk, kg = ops._packed(w, "dgrad", torch.bfloat16, C, 1, 9, C, C, 9, 0, 1, C * 9)
    pk2, kg2 = ops._packed_s2(w, torch.bfloat16)
    a = timeit(lambda: lib.xr_conv_igemm(0, ptr(dy), ptr(pk), None, ptr(dx), N, Ho, Ho, C, H, H, C, 3, 3, 2, 1, 1, kg, C, None, 0, None, None, None, 1,
                                         None, None, None, stream()))
    b = timeit(lambda: lib.xr_conv_dgrad_s2(0, ptr(dy), ptr(pk2), ptr(dx2), N, Ho, Ho, C, C, kg2, None, None, None, 1, None, None, stream()))
    err = float((dx.float() - dx2.float()).abs().max() / dx.float().abs().max())
    fl = 2.0 * N * Ho * Ho * C * C * 9
    print(f"C={C} {H}->{Ho}: class-mode {a:7.1f} us ({fl / a / 1e6:6.0f} TF/s)   dense 2x2 + d2s {b:7.1f} us ({fl / b / 1e6:6.0f} TF/s algorithmic, "
          f"{fl * 16 / 9 / b / 1e6:6.0f} executed)   max diff {err:.2e}", flush=True)
