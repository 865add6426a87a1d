#!/usr/bin/env python
"""Row-walking weight gradient of the stride-2 64 -> 64 layer at 112 x 112, batch 256: warm (back-to-back) vs cold (L2 / MALL
flushed by a 1 GB memset before every launch) -- the in-step launch takes 438 us, the warm one 113 us."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
from xrface._lib import lib, ptr, stream
dev = torch.device("cuda:0")
flush = torch.empty(1 << 30, dtype=torch.uint8, device=dev)
for C, K, H, s, N in ((64, 64, 112, 2, 256), (64, 64, 112, 1, 256), (128, 128, 56, 2, 256), (128, 128, 28, 1, 256)):
    Ho = H // s
    x = torch.randn(N, H, H, C, device=dev).bfloat16(); dy = torch.randn(N, Ho, Ho, K, device=dev).bfloat16()
    cap = max(1, 256 // ((K // 64) * (C // 64)))
    slabs = torch.empty(cap, K, 9 * C, device=dev)
    fn = lambda: lib.xr_conv_wgrad_rows(ptr(x), ptr(dy), ptr(slabs), N, H, H, C, K, s, cap, stream())
    for cold in (0, 1):
        ts = []
        for _ in range(8):
            if cold: flush.zero_()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1))
        ts.sort()
        print(f"{C}->{K} @{H} s{s}: {'cold' if cold else 'warm'} {ts[len(ts)//2]*1e3:7.1f} us (min {ts[0]*1e3:.1f})")
