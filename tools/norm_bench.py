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
