#!/usr/bin/env python
"""Experiment (round 3): does running the 64-channel residual trunk (3 blocks x 3 passes, forward + backward) DEPTH-FIRST over image
chunks -- so that every tensor a kernel reads was written by the kernel just before it and is still in the 256 MB Infinity
Cache -- beat the breadth-first order?  InstanceNorm is per image, so chunks are independent.  Both orders are captured into ONE
HIP graph each (no host launch cost in the comparison; single stream, weight gradients included)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
import xrface
from xrface import ops, parallel
from xrface.graph import GraphedStep
from xrface.model import FSRnet

dev = torch.device("cuda:0")
N = int(os.environ.get("N", 128)); H = int(os.environ.get("H", 112))
xrface.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net = FSRnet.Course_SR_Network().to(dev)
blocks = list(net.residual)
flat = parallel.FlatParams([p for b in blocks for p in b.parameters()])
x = torch.randn(N, H, H, 64, device=dev).bfloat16()
gy = torch.randn(N, H, H, 64, device=dev).bfloat16() * 0.01


def make(chunk, fwd_only=False):
    def fn(x_, gy_):
        flat.grad.zero_()
        for n0 in range(0, N, chunk):
            xc = x_[n0:n0 + chunk].detach().requires_grad_(True)
            y = ops.res_trunk64(xc, blocks, 3)
            if not fwd_only:
                y.backward(gy_[n0:n0 + chunk])
        return flat.grad
    return fn


def timed(gs, reps=5):
    for _ in range(2):
        gs(x, gy)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        gs(x, gy)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


T = N * H * H * 64 * 2 / 1e6
print(f"N={N} H={H}: tensor {T:.0f} MB", flush=True)
ref = None
for fwd_only in (False, True):
    for chunk in [c for c in (N, 64, 32, 16, 8, 4) if c <= N]:
        gs = GraphedStep(make(chunk, fwd_only), [x, gy], warmup=1)
        ms = timed(gs)
        g = gs(x, gy).clone()
        if ref is None:
            ref = g
        cos = float(torch.nn.functional.cosine_similarity(g.flatten(), ref.flatten(), dim=0)) if not fwd_only else float("nan")
        print(f"  {'fwd only' if fwd_only else 'fwd + bwd'}  chunk {chunk:4d} ({T * chunk / N:6.1f} MB per tensor): {ms:8.3f} ms   grad cos vs unchunked {cos:.5f}",
              flush=True)
        gs.close()
        del gs
        torch.cuda.empty_cache()
