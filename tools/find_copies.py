#!/usr/bin/env python
"""Which Python call sites issue aten::copy_ / fill / tiny aten kernels inside one IR-SE-50 training step."""
import os, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
import xrface
from xrface import parallel
from xrface.loss.loss import CrossEntropyLoss
from xrface.model.model_irse import IR_SE_50
import bench

dev = torch.device("cuda:0")
xrface.set_compute_dtype(torch.bfloat16)
model = IR_SE_50([112, 112]).to(dev).train()
flat = parallel.FlatParams(model.parameters())
opt = parallel.FusedSGD(flat, lr=0.01, momentum=0.9, weight_decay=5e-4)
crit = CrossEntropyLoss()
x, y = bench.synth_batch(32, dev, 0)


def step():
    opt.zero_grad(); crit(model(x), y).backward(); opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
    step()
    torch.cuda.synchronize()
evs = sorted(prof.events(), key=lambda e: e.time_range.start)
cpu = [e for e in evs if e.device_type == torch.autograd.DeviceType.CPU]
names = collections.Counter()
for e in evs:
    if e.device_type != torch.autograd.DeviceType.CPU and ("emcpy" in e.name or "copyBuffer" in e.name or "emset" in e.name):
        names[e.name] += 1
print(names)
# runtime API calls that are memcpy, with the innermost enclosing aten / autograd op
cnt = collections.Counter()
for e in cpu:
    if "emcpy" in e.name or "emset" in e.name:
        par = e.cpu_parent
        chain = []
        while par is not None and len(chain) < 4:
            chain.append(par.name)
            par = par.cpu_parent
        cnt[(e.name, " < ".join(chain))] += 1
for k, c in cnt.most_common(20):
    print(c, k)
