#!/usr/bin/env python
"""Where does the HOST spend its time while it enqueues one IR-SE-50 training step (C2, batch 256)?  cProfile over 6 steps (after
warm-up); the GPU runs behind, so this is pure enqueue cost."""
import cProfile, os, pstats, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
import xrface
from xrface import parallel
from xrface.loss.loss import CrossEntropyLoss
from xrface.model.model_irse import IR_SE_50
import bench
dev = torch.device("cuda:0")
xrface.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = IR_SE_50([112, 112]).to(dev).train()
flat = parallel.FlatParams(model.parameters_in_execution_order())
opt = parallel.FusedSGD(flat, lr=0.01, momentum=0.9, weight_decay=5e-4)
crit = CrossEntropyLoss()
x, y = bench.synth_batch(int(os.environ.get("N", 256)), dev, 0)


def step():
    opt.zero_grad(); crit(model(x), y).backward(); opt.step()


for _ in range(4):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(6):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("TOP", 40)))
