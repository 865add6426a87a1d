#!/usr/bin/env python
"""Experiment: capture one whole training step (forward + backward + optimizer) in a HIP graph and replay it."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
import xrface
from xrface import parallel, ops
from xrface.loss.loss import CrossEntropyLoss
from xrface.model.model_irse import IR_SE_50
import bench

dev = torch.device("cuda:0")
xrface.set_compute_dtype(torch.bfloat16)
N = int(os.environ.get("N", 256))
torch.manual_seed(0)
model = IR_SE_50([112, 112]).to(dev).train()
flat = parallel.FlatParams(model.parameters())
opt = parallel.FusedSGD(flat, lr=0.01, momentum=0.9, weight_decay=5e-4)
crit = CrossEntropyLoss()
x, y = bench.synth_batch(N, dev, 0)
loss_out = torch.zeros((), device=dev)


def step():
    opt.zero_grad()
    loss = crit(model(x), y)
    loss.backward()
    opt.step()
    loss_out.copy_(loss.detach())


def timed(fn, reps=8):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    t1 = time.perf_counter(); torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, (t1 - t0) / reps * 1e3


s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    for _ in range(3):
        step()
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
print("eager ms (wall, host):", timed(step), "loss", float(loss_out))
ops._zpool.buf = None   # the zero slab must be (re)created inside the capture so that every replay re-zeroes it
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    step()
torch.cuda.synchronize()
print("captured")
g.replay(); torch.cuda.synchronize()
print("replay ms (wall, host):", timed(g.replay), "loss", float(loss_out))
for _ in range(5):
    g.replay()
torch.cuda.synchronize()
print("loss after 5 more replays", float(loss_out))
