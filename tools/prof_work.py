#!/usr/bin/env python
"""One workload for rocprofv3 (--kernel-trace --stats):  WORK=c4|c3|c2  N=<per-GPU batch>  STEPS=<n>.
c4: steps.c4_step (FHN -> IR-SE-50 student + assistant vs frozen teacher), c3: steps.fhn_step_fused, c2: the bench step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
import xrface
from xrface import parallel, steps
from xrface.loss.loss import CrossEntropyLoss
from xrface.model import FSRnet, model_irse

dev = torch.device("cuda:0")
xrface.set_compute_dtype({"bf16": torch.bfloat16, "fp32": torch.float32, "fp32x2": "fp32x2"}[os.environ.get("MODE", "bf16")])   # MODE=bf16|fp32|fp32x2
work = os.environ.get("WORK", "c4")
n = int(os.environ.get("N", 256 if work != "c3" else 128))
nsteps = int(os.environ.get("STEPS", 3))
torch.manual_seed(0)
g = torch.Generator(device=dev); g.manual_seed(5)
lo = torch.randn(n, 3, 14, 14, device=dev, generator=g)
hr = torch.nn.functional.interpolate(lo, size=(112, 112), mode="bilinear").clamp_(-1, 1).contiguous()
lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 7), size=(112, 112), mode="bilinear").contiguous()
mk_fhn = lambda: {"coarse": FSRnet.Course_SR_Network().to(dev), "prior": FSRnet.Prior_Estimation_Network().to(dev),
                  "encoder": FSRnet.Fine_SR_Encoder().to(dev), "decoder": FSRnet.Fine_SR_Decoder().to(dev)}
hp = torch.cuda.Stream(dev, priority=-1)
with torch.cuda.stream(hp):
    if work == "c4":
        fhn = mk_fhn()
        student, assistant = model_irse.IR_SE_50([112, 112]).to(dev), model_irse.IR_SE_50([112, 112]).to(dev)
        teacher = model_irse.IR_SE_50([112, 112]).to(dev).eval()
        for p_ in teacher.parameters():
            p_.requires_grad_(False)
        fp = [p_ for k in ("coarse", "prior", "encoder", "decoder") for p_ in fhn[k].parameters()]
        flats = [parallel.FlatParams(fp), parallel.FlatParams(student.parameters_in_execution_order()),
                 parallel.FlatParams(assistant.parameters_in_execution_order())]
        opts = [parallel.FusedRMSprop(f, lr=1e-5, weight_decay=1e-5) for f in flats]
        fn = lambda: steps.c4_step(fhn, student, assistant, teacher, lr, hr, optimizers=opts)
    elif work == "c3":
        fhn = mk_fhn()
        flats = {k: parallel.FlatParams(fhn[k].parameters()) for k in fhn}
        opts = {k: parallel.FusedRMSprop(flats[k], lr=1e-5, weight_decay=1e-5) for k in fhn}
        hm = torch.rand(n, 28, 28, device=dev); par = torch.randint(0, 11, (n, 1, 28, 28), device=dev)
        fn = lambda: steps.fhn_step_fused(fhn, lr, hr, hm, par, optimizers=opts)
    elif work == "sr":   # SR-variant generators + perceptual IR-50 losses (train_FHN.py:251-308), N = 32 by default
        from xrface.model import FSRnet_sr
        nets = {"coarse": FSRnet_sr.Coarse_SR_Network().to(dev), "encoder": FSRnet_sr.Fine_SR_Encoder().to(dev),
                "prior": FSRnet_sr.Prior_Estimation_Network().to(dev), "decoder": FSRnet_sr.Fine_SR_Decoder().to(dev)}
        bb = model_irse.IR_50([112, 112]).to(dev).eval()
        for p_ in bb.parameters():
            p_.requires_grad_(False)
        flats = {"coarse": parallel.FlatParams(nets["coarse"].parameters()), "prior": parallel.FlatParams(nets["prior"].parameters()),
                 "encdec": parallel.FlatParams(list(nets["encoder"].parameters()) + list(nets["decoder"].parameters()))}
        opts = {k: parallel.FusedAdam(f, lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-5) for k, f in flats.items()}
        hm = torch.rand(n, 112, 112, device=dev); par = torch.randint(0, 13, (n, 1, 112, 112), device=dev)
        def fn():
            for o in opts.values():
                o.zero_grad()
            steps.fhn_perceptual_step(nets, bb, lr, hr, hm, par, optimizers=opts)
    else:
        model = model_irse.IR_SE_50([112, 112]).to(dev).train()
        flat = parallel.FlatParams(model.parameters_in_execution_order())
        opt = parallel.FusedSGD(flat, lr=0.05, momentum=0.9, weight_decay=5e-4)
        y = torch.randint(0, 512, (n,), device=dev); crit = CrossEntropyLoss()
        def fn():
            opt.zero_grad(); crit(model(hr), y).backward(); opt.step()
    import time
    for i in range(nsteps):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize(); print(f"step {i}: {(time.perf_counter() - t0) * 1e3:.1f} ms", flush=True)
