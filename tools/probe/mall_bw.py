#!/usr/bin/env python
"""Streaming bandwidth by working-set size: a 16-B-per-lane elementwise pass (xr_affine_act, bf16, one read + one write) over buffers
from 8 MB to 1 GB, repeated back to back -- where does the 256 MB Infinity Cache show?"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
from xrface._lib import lib, ptr, stream, dt, ACT_PRELU, ACT_NONE
dev = torch.device("cuda:0")
C = 64
al = torch.rand(C, device=dev)
for mb in (4, 8, 16, 32, 64, 96, 128, 192, 256, 512, 1024):
    rows = mb * (1 << 20) // (C * 2)
    x = torch.randn(rows, C, device=dev).bfloat16()
    y = torch.empty_like(x)
    for name, fn, passes in (("act x->y", lambda: lib.xr_affine_act(dt(x), ptr(x), None, None, None, ptr(al), ACT_PRELU, ptr(y), 1, rows, C, 0, stream()), 2),
                             ("in place x->x", lambda: lib.xr_affine_act(dt(x), ptr(x), None, None, None, ptr(al), ACT_PRELU, ptr(x), 1, rows, C, 0, stream()), 2),
                             ("torch copy", lambda: y.copy_(x), 2)):
        for _ in range(3):
            fn()
        reps = max(4, min(200, 4096 // mb))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        print(f"{mb:5d} MB  {name:14s} {us:9.1f} us  {passes * mb * 1.048576 / us * 1e3 / 1e3:7.2f} TB/s", flush=True)
