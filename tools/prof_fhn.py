#!/usr/bin/env python
"""One FHN step workload for rocprofv3 (BASELINE config 3 shape: N = 32 pairs, bf16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
import xrface
from xrface import steps
from xrface.model import FSRnet
dev = torch.device("cuda:0")
xrface.set_compute_dtype(torch.bfloat16)
n = int(os.environ.get("FHN_N", 32))
nets = {k: c().to(dev) for k, c in (("coarse", FSRnet.Course_SR_Network), ("encoder", FSRnet.Fine_SR_Encoder),
                                    ("prior", FSRnet.Prior_Estimation_Network), ("decoder", FSRnet.Fine_SR_Decoder))}
opts = {k: torch.optim.RMSprop(v.parameters(), lr=1e-4, alpha=0.99, weight_decay=1e-5) for k, v in nets.items()}
g = torch.Generator(device=dev); g.manual_seed(5)
hr = torch.rand(n, 3, 112, 112, device=dev, generator=g) * 2 - 1
lr = hr.clone()
hm = torch.rand(n, 28, 28, device=dev); par = torch.randint(0, 11, (n, 1, 28, 28), device=dev)
for _ in range(int(os.environ.get("STEPS", 4))):
    for o in opts.values():
        o.zero_grad(set_to_none=True)
    steps.fhn_step(nets, lr, hr, hm, par, opts)
torch.cuda.synchronize()
