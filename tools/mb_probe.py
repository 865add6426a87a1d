#!/usr/bin/env python
"""Experiment: FHN step (C3) with the batch split into micro-batches that run on their own streams (InstanceNorm is per image, so
the halves are independent): does the GPU overlap the MFMA-bound convolutions of one half with the HBM-bound norm passes of the other?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
import xrface
from xrface import ops, parallel
from xrface.loss.loss import CrossEntropyLoss2d, MSELoss_Landmark, MSELossFunc
from xrface.model import FSRnet

dev = torch.device("cuda:0")
xrface.set_compute_dtype(torch.bfloat16)
n = int(os.environ.get("N", 128)); parts = int(os.environ.get("PARTS", 2))
torch.manual_seed(0)
g = torch.Generator(device=dev); g.manual_seed(5)
lo = torch.randn(n, 3, 14, 14, device=dev, generator=g)
hr = torch.nn.functional.interpolate(lo, size=(112, 112), mode="bilinear").clamp_(-1, 1).contiguous()
lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 7), size=(112, 112), mode="bilinear").contiguous()
hm = torch.rand(n, 28, 28, device=dev); par = torch.randint(0, 11, (n, 1, 28, 28), device=dev)
fhn = {"coarse": FSRnet.Course_SR_Network().to(dev), "prior": FSRnet.Prior_Estimation_Network().to(dev),
       "encoder": FSRnet.Fine_SR_Encoder().to(dev), "decoder": FSRnet.Fine_SR_Decoder().to(dev)}
flats = {k: parallel.FlatParams(fhn[k].parameters()) for k in fhn}
opts = {k: parallel.FusedRMSprop(flats[k], lr=1e-5, weight_decay=1e-5) for k in fhn}
mse97, lmk_loss, ce2d = MSELossFunc(), MSELoss_Landmark(), CrossEntropyLoss2d()
streams = [torch.cuda.Stream(dev) for _ in range(parts)]


def step(P):
    for o in opts.values():
        o.zero_grad()
    cur = torch.cuda.current_stream(dev)
    if P == 1:
        _, coarse = fhn["coarse"](lr)
        c_in = coarse.detach()
        pf, lmk, p_ = fhn["prior"](c_in)
        ef = fhn["encoder"](c_in)
        sr = fhn["decoder"](torch.cat((ops.grad_scale(pf, 0.1), ef), 1))
    else:
        lrs = lr.chunk(P)
        for s in streams[:P]:
            s.wait_stream(cur)
        co, pfs, lm, pa, efs, srs = [None] * P, [None] * P, [None] * P, [None] * P, [None] * P, [None] * P
        for i in range(P):
            with torch.cuda.stream(streams[i]):
                _, co[i] = fhn["coarse"](lrs[i])
        for i in range(P):
            with torch.cuda.stream(streams[i]):
                pfs[i], lm[i], pa[i] = fhn["prior"](co[i].detach())
        for i in range(P):
            with torch.cuda.stream(streams[i]):
                efs[i] = fhn["encoder"](co[i].detach())
        for i in range(P):
            with torch.cuda.stream(streams[i]):
                srs[i] = fhn["decoder"](torch.cat((ops.grad_scale(pfs[i], 0.1), efs[i]), 1))
        for s in streams[:P]:
            cur.wait_stream(s)
        coarse, lmk, p_, sr = torch.cat(co), torch.cat(lm), torch.cat(pa), torch.cat(srs)
    loss = 12.0 * mse97(coarse, hr) + 10.0 * mse97(sr, hr) + lmk_loss(lmk, hm) + ce2d(p_, par)
    loss.backward()
    for o in opts.values():
        o.step()
    return loss


for P in (1, parts, 1, parts):
    for _ in range(2):
        step(P)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(4):
        l = step(P)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 4 * 1e3
    print(f"parts={P}: {dt:.1f} ms/step  loss {float(l):.4f}", flush=True)
