#!/usr/bin/env python
"""Secondary timings for DESIGN.md: BASELINE configs 1, 3, 4 (single GPU) and 5.  Not the graded bench."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import numpy as np
import torch
import xrface
from xrface import parallel, steps
from xrface.graph import GraphedStep
from xrface.model import FSRnet, model_irse, resnet
from xrface.loss.loss import MSELossFunc

dev = torch.device("cuda:0")


def timed(fn, warm=2, reps=5):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


def faces(n):
    g = torch.Generator(device=dev); g.manual_seed(5)
    lo = torch.randn(n, 3, 14, 14, device=dev, generator=g)
    return torch.nn.functional.interpolate(lo, size=(112, 112), mode="bilinear").clamp_(-1, 1).contiguous()


out = {}
# ---- C1: Course_SR_Network fwd+bwd of 12*mse97 (+RMSprop)
for dtype, n in ((torch.float32, 4), (torch.bfloat16, 4), (torch.bfloat16, 64)):
    xrface.set_compute_dtype(dtype)
    net = FSRnet.Course_SR_Network().to(dev)
    opt = torch.optim.RMSprop(net.parameters(), lr=1e-4, alpha=0.99, weight_decay=1e-5)
    hr = faces(n); lr = hr.clone()
    def step():
        opt.zero_grad(set_to_none=True)
        _, img = net(lr)
        (12.0 * MSELossFunc()(img, hr)).backward()
        opt.step()
    ms = timed(step)
    out[f"C1 coarse N={n} {str(dtype).split('.')[-1]}"] = {"ms": round(ms, 3), "img_s": round(n / ms * 1e3, 1)}
    try:  # the same step captured in one HIP graph (RMSprop must be capturable)
        opt = torch.optim.RMSprop(net.parameters(), lr=1e-4, alpha=0.99, weight_decay=1e-5, capturable=True)
        def gstep(lr_, hr_):
            opt.zero_grad(set_to_none=False)
            _, img = net(lr_)
            (12.0 * MSELossFunc()(img, hr_)).backward()
            opt.step()
        gs = GraphedStep(gstep, [lr, hr])
        ms = timed(lambda: gs(lr, hr))
        out[f"C1 coarse N={n} {str(dtype).split('.')[-1]} HIP graph"] = {"ms": round(ms, 3), "img_s": round(n / ms * 1e3, 1)}
    except Exception as e:  # noqa: BLE001
        out[f"C1 coarse N={n} {str(dtype).split('.')[-1]} HIP graph"] = {"error": repr(e)[:200]}

# ---- C3: full FHN step (per-pair gradients), bf16
xrface.set_compute_dtype(torch.bfloat16)
n = int(os.environ.get("FHN_N", 32))
nets = {k: c().to(dev) for k, c in (("coarse", FSRnet.Course_SR_Network), ("encoder", FSRnet.Fine_SR_Encoder),
                                    ("prior", FSRnet.Prior_Estimation_Network), ("decoder", FSRnet.Fine_SR_Decoder))}
opts = {k: torch.optim.RMSprop(v.parameters(), lr=1e-4, alpha=0.99, weight_decay=1e-5) for k, v in nets.items()}
hr = faces(n); lr = hr.clone()
hm = torch.rand(n, 28, 28, device=dev); par = torch.randint(0, 11, (n, 1, 28, 28), device=dev)
ms = timed(lambda: steps.fhn_step(nets, lr, hr, hm, par, opts), warm=2, reps=3)
out[f"C3 FHN step N={n} bf16"] = {"ms": round(ms, 2), "img_s": round(n / ms * 1e3, 1)}
try:
    opts = {k: torch.optim.RMSprop(v.parameters(), lr=1e-4, alpha=0.99, weight_decay=1e-5, capturable=True) for k, v in nets.items()}
    def fstep(lr_, hr_, hm_, par_):
        for o in opts.values():
            o.zero_grad(set_to_none=False)
        steps.fhn_step(nets, lr_, hr_, hm_, par_, opts)
    gs = GraphedStep(fstep, [lr, hr, hm, par], warmup=2)
    ms = timed(lambda: gs(lr, hr, hm, par), warm=1, reps=3)
    out[f"C3 FHN step N={n} bf16 HIP graph"] = {"ms": round(ms, 2), "img_s": round(n / ms * 1e3, 1)}
    del gs
except Exception as e:  # noqa: BLE001
    out[f"C3 FHN step N={n} bf16 HIP graph"] = {"error": repr(e)[:300]}
del nets, opts

# ---- C4-like: residual KD step, teacher IR-50 (frozen) + student/assistant ResNet-34, bf16
n = int(os.environ.get("KD_N", 64))
teacher = model_irse.IR_50([112, 112]).to(dev).eval()
student, assistant = resnet.ResNet_34().to(dev), resnet.ResNet_34().to(dev)
so = torch.optim.RMSprop(student.parameters(), lr=1e-4, weight_decay=1e-5)
ao = torch.optim.RMSprop(assistant.parameters(), lr=1e-4, weight_decay=1e-5)
x = faces(n)
def kd():
    so.zero_grad(set_to_none=True); ao.zero_grad(set_to_none=True)
    steps.kd_step(teacher, student, assistant, x, so, ao)
ms = timed(kd, warm=2, reps=3)
out[f"C4 KD step N={n} bf16"] = {"ms": round(ms, 2), "img_s": round(n / ms * 1e3, 1)}
try:
    so = torch.optim.RMSprop(student.parameters(), lr=1e-4, weight_decay=1e-5, capturable=True)
    ao = torch.optim.RMSprop(assistant.parameters(), lr=1e-4, weight_decay=1e-5, capturable=True)
    def kstep(x_):
        so.zero_grad(set_to_none=False); ao.zero_grad(set_to_none=False)
        steps.kd_step(teacher, student, assistant, x_, so, ao)
    gs = GraphedStep(kstep, [x], warmup=3)
    ms = timed(lambda: gs(x), warm=1, reps=5)
    out[f"C4 KD step N={n} bf16 HIP graph"] = {"ms": round(ms, 2), "img_s": round(n / ms * 1e3, 1)}
    del gs
except Exception as e:  # noqa: BLE001
    out[f"C4 KD step N={n} bf16 HIP graph"] = {"error": repr(e)[:300]}
del teacher, student, assistant

# ---- C5: P = 1e6 pair distances + 4000-threshold / 10-fold ROC
from xrface.utils.utils import calculate_roc, pair_dist
P = int(os.environ.get("PAIRS", 1000000))
g = torch.Generator(device=dev); g.manual_seed(0)
e1 = torch.randn(P, 512, device=dev, generator=g)
same = torch.rand(P, device=dev, generator=g) < 0.5
e2 = torch.where(same[:, None], e1 + 0.5 * torch.randn(P, 512, device=dev, generator=g), torch.randn(P, 512, device=dev, generator=g))
ms = timed(lambda: pair_dist(e1, e2), warm=1, reps=5)
out["C5 pairdist P=1e6"] = {"ms": round(ms, 3), "GB_s": round(P * (2 * 512 * 4 + 4) / ms / 1e6, 1)}
fold = np.random.RandomState(0).randint(0, 10, P).astype(np.int32)
t0 = time.perf_counter()
tpr, fpr, acc, best = calculate_roc(np.arange(0, 12000, 3), e1, e2, same.cpu().numpy(), nrof_folds=10, fold_id=fold)
torch.cuda.synchronize()
out["C5 calculate_roc P=1e6 (incl. host prefix sums)"] = {"ms": round((time.perf_counter() - t0) * 1e3, 1), "acc": round(float(acc), 4)}

# ---- embedding extraction (the forward half of the evaluation path): IR-SE-50 eval forward, bf16
xrface.set_compute_dtype(torch.bfloat16)
n = int(os.environ.get("EMB_N", 256))
net = model_irse.IR_SE_50([112, 112]).to(dev).eval()
xe = faces(n)
with torch.no_grad():
    ms = timed(lambda: net(xe), warm=3, reps=8)
out[f"IR-SE-50 eval forward N={n} bf16"] = {"ms": round(ms, 3), "img_s": round(n / ms * 1e3, 1)}
print(json.dumps(out, indent=1))
