#!/usr/bin/env python
"""Weight gradient of the wide IR-SE-50 layers at batch 256 (bf16): the 4-wave sliced kernel (xr_tune knob 13 = 0) vs the
8-wave ring kernel (xr_wgrad8.hip, knob 13 = 2), per slice count.  Prints microseconds and TFLOP/s; COLD=1 flushes L2 / MALL
between launches with a 512 MB memset (operands then come from HBM, as inside the training step)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
from xrface import ops
from xrface._lib import lib, ptr, stream, dt

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 256))
SHAPES = [(256, 256, 14, 1), (256, 512, 14, 1), (512, 512, 7, 1), (256, 256, 28, 2), (512, 512, 14, 2), (128, 256, 28, 1), (128, 128, 28, 1)]
cold = bool(os.environ.get("COLD"))
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev) if cold else None


def timeit(fn, reps=10):
    for _ in range(2):
        fn()
    ts = []
    for _ in range(reps):
        if cold:
            flush.zero_()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts.sort()
    return ts[len(ts) // 2]


for C, K, H, st in SHAPES:
    Ho = (H + 2 - 3) // st + 1
    x = torch.randn(N, H, H, C, device=dev).bfloat16()
    dy = torch.randn(N, Ho, Ho, K, device=dev).bfloat16()
    kg = ops.kg_of(9, C)
    flops = 2.0 * N * Ho * Ho * K * C * 9
    base = ops._wgrad_split(N * Ho * Ho, K, kg)
    row = f"{C:3d}->{K:3d} @{H:2d} s{st} {flops / 1e9:6.1f} GF |"
    for knob, splits in ((0, (base,)), (2, sorted({max(1, base // 2), base, base * 2}))):
        for split in splits:
            slab = torch.zeros(split, K, kg, device=dev)
            lib.xr_tune(13, knob)
            ms = timeit(lambda: lib.xr_conv_wgrad(dt(x), ptr(x), ptr(dy), ptr(slab), N, H, H, C, Ho, Ho, K, 3, 3, st, 1, 0, K, kg, split, stream()))
            row += f"  {'4w' if knob == 0 else '8w'} x{split:<3d} {ms * 1e3:6.1f} us {flops / ms / 1e9:5.0f} TF |"
    print(row)
lib.xr_tune(13, 1)
