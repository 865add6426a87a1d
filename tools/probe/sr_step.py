#!/usr/bin/env python
"""Experiment: the SR-variant perceptual-loss step (SUPER_RESOLUTION/train_FHN.py:251-308; 4 x depth-4 bottleneck hourglass prior =
475 convolutions of 64 channels, thousands of small launches) eager vs replayed as one HIP graph."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "cross-resolution-face-recognition_amd"))
import torch, xrface
from xrface import ops, parallel, steps
from xrface.graph import GraphedStep
from xrface.model import FSRnet_sr as M, model_irse
dev = torch.device("cuda:0"); xrface.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
nets = {"coarse": M.Coarse_SR_Network().to(dev), "encoder": M.Fine_SR_Encoder().to(dev), "prior": M.Prior_Estimation_Network().to(dev),
        "decoder": M.Fine_SR_Decoder().to(dev)}
bb = model_irse.IR_50([112, 112]).to(dev).eval()
for p in bb.parameters():
    p.requires_grad_(False)
flats = {"coarse": parallel.FlatParams(nets["coarse"].parameters()), "prior": parallel.FlatParams(nets["prior"].parameters()),
         "encdec": parallel.FlatParams(list(nets["encoder"].parameters()) + list(nets["decoder"].parameters()))}
opts = {k: parallel.FusedAdam(f, lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-5) for k, f in flats.items()}
hr = torch.randn(n, 3, 112, 112, device=dev).clamp_(-1, 1)
lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 7), size=(112, 112), mode="bilinear").contiguous()
hm = torch.rand(n, 112, 112, device=dev); par = torch.randint(0, 13, (n, 1, 112, 112), device=dev)
out = torch.zeros(3, device=dev)


def st(lr_, hr_, hm_, par_):
    for o in opts.values():
        o.zero_grad()
    l, _ = steps.fhn_perceptual_step(nets, bb, lr_, hr_, hm_, par_, optimizers=opts)
    out.copy_(torch.stack([l["coarse"].float(), l["prior"].float(), l["encdec"].float()]))
    return out


def timeit(fn, reps=3):
    for _ in range(2):
        fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


e = timeit(lambda: st(lr, hr, hm, par))
print(f"SR perceptual step N={n}: eager {e:.1f} ms = {n / e * 1e3:.0f} images/s  losses {out.tolist()}", flush=True)
gs = GraphedStep(st, [lr, hr, hm, par], warmup=2)
g = timeit(lambda: gs(lr, hr, hm, par), reps=5)
print(f"SR perceptual step N={n}: graph replay {g:.1f} ms = {n / g * 1e3:.0f} images/s  losses {out.tolist()}", flush=True)
gs.close()
