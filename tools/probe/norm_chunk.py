#!/usr/bin/env python
"""Experiment: does running the two passes of an InstanceNorm backward (reduce, then apply) per image chunk keep the apply pass's
re-reads in the 256 MB Infinity Cache?  CHUNK images per (reduce, coeffs, apply) triple vs one triple over the whole batch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
from xrface._lib import lib, ptr, stream, dt, ACT_PRELU
dev = torch.device("cuda:0")
N = int(os.environ.get("N", 128)); H = 112; C = 64; HW = H * H
x = torch.randn(N, H, H, C, device=dev).bfloat16(); res = torch.randn_like(x); dy = torch.randn_like(x)
dx = torch.empty_like(x); dres = torch.empty_like(x)
sc = torch.rand(N, C, device=dev) + 0.5; sh = torch.randn(N, C, device=dev); al = torch.rand(C, device=dev)
gm = torch.rand(C, device=dev); mean = torch.randn(N, C, device=dev); invstd = torch.rand(N, C, device=dev) + 0.5
red = torch.zeros(3, N, C, device=dev); coef = torch.empty(3, N, C, device=dev)
dg = torch.zeros(C, device=dev); db = torch.zeros(C, device=dev); da = torch.zeros(C, device=dev)
d = dt(x)


def run(chunk, with_res=True):
    for n0 in range(0, N, chunk):
        nc = min(chunk, N - n0)
        r = red[:, n0:n0 + nc].contiguous() if chunk < N else red
        r.zero_()
        cf = torch.empty(3, nc, C, device=dev)
        rs = res[n0:n0 + nc] if with_res else None
        lib.xr_affine_act_bwd_reduce(d, ptr(x[n0:n0 + nc]), ptr(sc[n0:n0 + nc]), ptr(sh[n0:n0 + nc]), ptr(rs), ptr(al), ACT_PRELU,
                                     ptr(dy[n0:n0 + nc]), ptr(r), nc, HW, C, 1, stream())
        lib.xr_norm_bwd_coeffs(ptr(r), ptr(gm), ptr(mean[n0:n0 + nc]), ptr(invstd[n0:n0 + nc]), ptr(cf), ptr(dg), ptr(db), ptr(da), nc, HW, C, 1,
                               stream())
        lib.xr_affine_act_bwd_apply(d, ptr(x[n0:n0 + nc]), ptr(sc[n0:n0 + nc]), ptr(sh[n0:n0 + nc]), ptr(rs), ptr(al), ACT_PRELU,
                                    ptr(dy[n0:n0 + nc]), ptr(cf), ptr(dx[n0:n0 + nc]), ptr(dres[n0:n0 + nc]) if with_res else None, nc, HW, C, 1, None,
                                    stream())


def timeit(fn, reps=6):
    for _ in range(2):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3


T = N * HW * C * 2 / 1e6
for with_res in (True, False):
    row = f"N={N} T={T:.0f} MB  {'tail (x, res, dy -> dx, dres: 3+5 T)' if with_res else 'mid (x, dy -> dx: 2+3 T)'}:"
    for chunk in (N, 64, 32, 16, 8):
        if chunk > N:
            continue
        us = timeit(lambda: run(chunk, with_res))
        row += f"  chunk {chunk}: {us:.0f} us"
    print(row, flush=True)
