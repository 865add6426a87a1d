#!/usr/bin/env python
"""C2 step (IR-SE-50, batch 256, bf16) with a one-rank RCCL group and BucketedAllReduce(force=True): buckets launched from the
backward stream after joining the weight-gradient side stream (comm_stream=False) vs from a launcher stream that waits for both
(comm_stream=True).  One rank moves no bytes: what is measured is the serialisation the join adds to the backward pass."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch, torch.distributed as dist
import xrface
from xrface import parallel
from xrface.loss.loss import CrossEntropyLoss
from xrface.model.model_irse import IR_SE_50
import bench
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
xrface.set_compute_dtype(torch.bfloat16)
model = IR_SE_50([112, 112]).to(dev).train()
flat = parallel.FlatParams(model.parameters_in_execution_order())
opt = parallel.FusedSGD(flat, lr=0.01, momentum=0.9, weight_decay=5e-4)
crit = CrossEntropyLoss()
x, y = bench.synth_batch(256, dev, 0)
mode = os.environ.get("MODE", "launcher")   # none | join | launcher  (one reducer per process: the hooks of two would both fire)
red = None if mode == "none" else parallel.BucketedAllReduce(flat, force=True, comm_stream=(mode == "launcher"))
hp = torch.cuda.Stream(priority=-1)
ts = []
with torch.cuda.stream(hp):
    def step():
        opt.zero_grad(); crit(model(x), y).backward()
        if red is not None:
            red.finish()
        opt.step()
    for _ in range(4):
        step()
    for rnd in range(5):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 5 * 1e3)
ts.sort()
early = red.last_early if red is not None else "-"
print(f"{mode:10s} median {ts[len(ts) // 2]:.3f} ms  min {ts[0]:.3f}   buckets launched during backward: {early} of {len(red.buckets) if red else 0}")
dist.destroy_process_group()
