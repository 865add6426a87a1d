#!/usr/bin/env python
"""Per-shape micro-benchmark of the conv kernels (fwd / dgrad / wgrad) on the IR-SE-50 layer shapes at batch 256,
for A/B-ing kernel variants in one process (xr_tune knobs).  Prints TFLOP/s per shape and variant."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
from xrface import ops
from xrface._lib import lib, ptr, stream, dt

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 256))
SHAPES = [  # (C, K, H, stride)
    (64, 64, 112, 1), (64, 64, 112, 2), (64, 64, 56, 1), (64, 128, 56, 1), (128, 128, 56, 2), (128, 128, 28, 1),
    (128, 256, 28, 1), (256, 256, 28, 2), (256, 256, 14, 1), (256, 512, 14, 1), (512, 512, 14, 2), (512, 512, 7, 1),
]


def timeit(fn, reps=8):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    variants = [tuple(int(q) for q in v.split(",")) for v in os.environ.get("VARIANTS", "3,1,0").split(";")]
    variants = [tuple(list(v) + [0] * (5 - len(v))) for v in variants]  # knobs 0, !2, 3, 4, 7
    dtype = torch.bfloat16
    print(f"{'shape':28s} " + " ".join(f"{'fwd/dgr/wgr %d,%d,%d,%d,%d' % v:>26s}" for v in variants))
    tot = {v: [0.0, 0.0, 0.0] for v in variants}
    for C, K, H, st in SHAPES:
        Ho = (H + 2 - 3) // st + 1
        x = torch.randn(N, H, H, C, device=dev).to(dtype)
        w = torch.randn(K, C, 3, 3, device=dev) * 0.05
        dy = torch.randn(N, Ho, Ho, K, device=dev).to(dtype)
        y = torch.empty_like(dy)
        dx = torch.empty_like(x)
        pk, kg = ops._packed(w, "fwd", dtype, K, 1, 9, C, C, C * 9, 0, 1, 9)
        pkd, kgd = ops._packed(w, "dgrad", dtype, C, 1, 9, K, K, 9, 0, 1, C * 9)
        split = ops._wgrad_split(N * Ho * Ho, K, kg)
        slab = torch.zeros(split, K, kg, device=dev)
        split = ops._wgrad_split(N * Ho * Ho, K, kg)
        flops = 2.0 * N * Ho * Ho * K * C * 9
        row = f"{C:3d}->{K:3d} @{H:3d} s{st} {flops/1e9:6.1f}GF "
        ref = None
        for v in variants:
            lib.xr_tune(0, v[0]); lib.xr_tune(2, 1 - v[1]); lib.xr_tune(3, v[2]); lib.xr_tune(4, v[3]); lib.xr_tune(7, v[4])
            f = timeit(lambda: lib.xr_conv_igemm(dt(x), ptr(x), ptr(pk), None, ptr(y), N, H, H, C, Ho, Ho, K, 3, 3, st, 1, 0, kg, K, None, 0, None, None, None, 1, None, None, None, stream()))
            d = timeit(lambda: lib.xr_conv_igemm(dt(x), ptr(dy), ptr(pkd), None, ptr(dx), N, Ho, Ho, K, H, H, C, 3, 3, st, 1, 1, kgd, C, None, 0, None, None, None, 1, None, None, None, stream()))
            if os.environ.get("EP"):  # dgrad with the fused PReLU-backward epilogue instead of the plain one
                sp = int(os.environ.get('EP')); al = torch.full((C,), 0.25, device=dev); dal = torch.zeros(sp, C, device=dev)
                d = timeit(lambda: lib.xr_conv_igemm(dt(x), ptr(dy), ptr(pkd), None, ptr(dx), N, Ho, Ho, K, H, H, C, 3, 3, st, 1, 1, kgd, C, None, 0, ptr(x), ptr(al), ptr(dal), sp, None, None, None, stream()))
            g = timeit(lambda: lib.xr_conv_wgrad(dt(x), ptr(x), ptr(dy), ptr(slab), N, H, H, C, Ho, Ho, K, 3, 3, st, 1, 0, K, kg, split, stream()))
            torch.cuda.synchronize()
            cur = (y.float().clone(), dx.float().clone())
            if ref is None:
                ref = cur
            else:
                err = max(float((cur[0] - ref[0]).abs().max() / ref[0].abs().max()),
                          float((cur[1] - ref[1]).abs().max() / ref[1].abs().max()))
                if err > 1e-2:
                    row += f" !!DIFF {err:.3g}"
            for i, tms in enumerate((f, d, g)):
                tot[v][i] += tms
            row += "   " + "/".join(f"{flops / (tms * 1e-3) / 1e12:6.0f}" for tms in (f, d, g))
        print(row)
    for v in variants:
        print(f"variant {v}: total ms fwd {tot[v][0]:.3f} dgrad {tot[v][1]:.3f} wgrad {tot[v][2]:.3f}")


if __name__ == "__main__":
    main()
