import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch, copy
import xrface
from xrface import ops, parallel
from xrface.model import FSRnet, model_irse
from xrface.steps import c4_step
dev = torch.device("cuda:0")
xrface.set_compute_dtype(torch.bfloat16)
torch.manual_seed(31)
N = int(os.environ.get("N", 256))
fhn = {"coarse": FSRnet.Course_SR_Network().to(dev), "prior": FSRnet.Prior_Estimation_Network().to(dev),
       "encoder": FSRnet.Fine_SR_Encoder().to(dev), "decoder": FSRnet.Fine_SR_Decoder().to(dev)}
student, assistant = model_irse.IR_SE_50([112, 112]).to(dev), model_irse.IR_SE_50([112, 112]).to(dev)
teacher = model_irse.IR_SE_50([112, 112]).to(dev).eval()
for m in (student, assistant):
    m.output_layer[1].p = 0.0
for p_ in teacher.parameters():
    p_.requires_grad_(False)
g = torch.Generator(device=dev); g.manual_seed(8)
lo = torch.randn(N, 3, 14, 14, device=dev, generator=g)
hr = torch.nn.functional.interpolate(lo, size=(112, 112), mode="bilinear").clamp_(-1, 1).contiguous()
lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 7), size=(112, 112), mode="bilinear").contiguous()
fhn_params = [p_ for k in ("coarse", "prior", "encoder", "decoder") for p_ in fhn[k].parameters()]
flats = [parallel.FlatParams(fhn_params), parallel.FlatParams(student.parameters()), parallel.FlatParams(assistant.parameters())]
res = {}
cos = lambda u, v: float(torch.nn.functional.cosine_similarity(u, v, dim=0))
for lock in (1, 0):
    ops._cfg["lockstep"] = lock
    for mode in (1, 0, 1, 0):
        ops._cfg["ir_block"] = mode
        for f in flats:
            f.zero_grad()
        (l1, a1), _ = c4_step(fhn, student, assistant, teacher, lr, hr)
        torch.cuda.synchronize()
        gs = [f.grad.clone() for f in flats]
        key = (lock, mode)
        if key in res:
            print(f"lock={lock} ir_block={mode} repeat: " + " ".join(f"{cos(u, v):.4f}" for u, v in zip(gs, res[key])) + f"  losses {l1.item():.5f} {a1.item():.5f}", flush=True)
        else:
            res[key] = gs
    print(f"lock={lock} block vs ops: " + " ".join(f"{cos(u, v):.4f}" for u, v in zip(res[(lock, 1)], res[(lock, 0)])), flush=True)
    # per-parameter worst offenders of the assistant
    worst = []
    f = flats[2]
    for (n_, p_), o in zip(assistant.named_parameters(), f.offsets):
        a_, b_ = res[(lock, 1)][2][o:o + p_.numel()], res[(lock, 0)][2][o:o + p_.numel()]
        worst.append((cos(a_, b_), n_))
    worst.sort()
    print("   assistant, lowest per-parameter cosines:", [(round(c, 3), n_) for c, n_ in worst[:8]], flush=True)
