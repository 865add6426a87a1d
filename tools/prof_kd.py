#!/usr/bin/env python
"""One residual-KD step workload for rocprofv3 (BASELINE config 4 shape: IR-50 teacher + 2 x ResNet-34, bf16)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
import xrface
from xrface import steps
from xrface.model import model_irse, resnet
dev = torch.device("cuda:0")
xrface.set_compute_dtype(torch.bfloat16)
n = int(os.environ.get("KD_N", 256))
teacher = model_irse.IR_50([112, 112]).to(dev).eval()
student, assistant = resnet.ResNet_34().to(dev), resnet.ResNet_34().to(dev)
so = torch.optim.RMSprop(student.parameters(), lr=1e-4, weight_decay=1e-5)
ao = torch.optim.RMSprop(assistant.parameters(), lr=1e-4, weight_decay=1e-5)
g = torch.Generator(device=dev); g.manual_seed(5)
x = torch.rand(n, 3, 112, 112, device=dev, generator=g) * 2 - 1
for _ in range(int(os.environ.get("STEPS", 4))):
    so.zero_grad(set_to_none=True); ao.zero_grad(set_to_none=True)
    steps.kd_step(teacher, student, assistant, x, so, ao)
torch.cuda.synchronize()
