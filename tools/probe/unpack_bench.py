#!/usr/bin/env python
"""xr_unpack_wgrad (slab sum + conversion to the parameter layout) per layer shape and slices-per-group (xr_tune knob 10)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
from xrface._lib import lib, ptr, stream

dev = torch.device("cuda:0")
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)


def timeit(fn, reps=10):
    ts = []
    for _ in range(reps + 2):
        flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]


for K, C, ns in ((64, 64, 256), (256, 256, 14), (512, 512, 3), (128, 128, 4), (128, 128, 56), (64, 128, 256)):
    kg = (9 * C + 63) // 64 * 64
    slabs = torch.randn(ns, K, kg, device=dev)
    dst = torch.zeros(K, C, 3, 3, device=dev)
    row = f"K={K:3d} C={C:3d} slices={ns:3d} ({ns * K * kg * 4 / 1e6:6.1f} MB):"
    for spg in (0, 4, 8, 16, 32, 64):
        lib.xr_tune(10, spg)
        ms = timeit(lambda: lib.xr_unpack_wgrad(ptr(slabs), ptr(dst), K, 1, 9, C, C, kg, C * 9, 0, 1, 9, 1, ns, stream()))
        row += f"  spg={spg:2d} {ms * 1e3:6.1f} us"
    print(row)
lib.xr_tune(10, 0)
