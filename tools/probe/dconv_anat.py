#!/usr/bin/env python
"""Anatomy of the direct 64 -> 64 convolution (xr_conv64.hip): time per 16 x 16 tile with parts of the kernel switched off
(tuning knob 14: bit 0 no halo loads, bit 1 no accumulator -> LDS epilogue, bit 2 no output stores; results are then wrong)
and with the tensors resident in L2 / Infinity Cache (small N) instead of streamed from HBM."""
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


H = 112
for N in [int(v) for v in os.environ.get('NS', '128,32,8').split(',')]:
    x = torch.randn(N, H, H, 64, device=dev).bfloat16(); y = torch.empty_like(x); c1 = torch.randn_like(x); dz = torch.randn_like(x)
    w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
    pk, _ = ops._packed(w, "fwd", torch.bfloat16, 64, 1, 9, 64, 64, 576, 0, 1, 9)
    sc = torch.rand(N, 64, device=dev) + 0.5; sh = torch.randn(N, 64, device=dev); al = torch.rand(64, device=dev)
    stats = torch.zeros(2, N, 64, device=dev); red = torch.zeros(3, N, 64, device=dev)
    tiles = N * 49
    tpc = -(-tiles // 256)
    fl = 2.0 * N * H * H * 64 * 576
    variants = {
        "plain": lambda: lib.xr_conv64_direct(ptr(x), ptr(pk), None, ptr(y), N, H, H, 0, None, None, None, None, None, stream()),
        "stats": lambda: lib.xr_conv64_direct(ptr(x), ptr(pk), None, ptr(y), N, H, H, 0, None, None, None, ptr(stats), None, stream()),
        "norm+stats": lambda: lib.xr_conv64_direct(ptr(x), ptr(pk), None, ptr(y), N, H, H, 0, ptr(sc), ptr(sh), ptr(al), ptr(stats), None, stream()),
        "add": lambda: lib.xr_conv64_direct(ptr(x), ptr(pk), None, ptr(y), N, H, H, 1, None, None, None, None, ptr(c1), stream()),
        "bwdred(EP3)": lambda: lib.xr_conv64_direct_bwdred(ptr(x), ptr(pk), ptr(y), N, H, H, 1, ptr(c1), ptr(sc), ptr(sh), ptr(al), ptr(red), stream()),
        "tailred(EP4)": lambda: lib.xr_conv64_direct_tailred(ptr(x), ptr(pk), ptr(y), N, H, H, 1, ptr(dz), ptr(c1), ptr(x), ptr(sc), ptr(sh), ptr(al), ptr(red), stream()),
    }
    print(f"N={N}: {tiles} tiles, {tpc} per CU, tensor {N * H * H * 128 / 1e6:.0f} MB, {fl / 1e9:.0f} GFLOP", flush=True)
    for name, fn in variants.items():
        row = f"  {name:14s}"
        for dbg in [int(v) for v in os.environ.get('DBGS', '0,1,4,5,2,7').split(',')]:
            lib.xr_tune(14, dbg)
            us = timeit(fn)
            row += f"  dbg{dbg}: {us:7.1f} us ({us / tpc:5.2f}/tile, {fl / us / 1e6:5.0f} TF/s)"
        lib.xr_tune(14, 0)
        print(row, flush=True)
