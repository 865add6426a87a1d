#!/usr/bin/env python
"""Per convolution launch site (kind, shape) table of ONE training step of a workload, timed inside the step with HIP event pairs on the
stream each launch runs on (ops._probe_begin; the same machinery bench.py uses for its `kernels` table):
    WORK=c2|c3|c4 [N=<per-GPU batch>] python tools/all_sites.py > profiles/rNN_<work>_all_sites.json"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
import xrface
from xrface import ops, parallel, steps
from xrface.loss.loss import CrossEntropyLoss
from xrface.model import FSRnet, model_irse
import bench

dev = torch.device("cuda:0")
xrface.set_compute_dtype(torch.bfloat16)
work = os.environ.get("WORK", "c4")
n = int(os.environ.get("N", 128 if work == "c3" else 256))
torch.manual_seed(0)
hr, y = bench.synth_batch(n, dev, 11)
lr = bench.synth_lr(hr)
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
    else:
        model = model_irse.IR_SE_50([112, 112]).to(dev).train()
        flat = parallel.FlatParams(model.parameters_in_execution_order())
        opt = parallel.FusedSGD(flat, lr=0.05, momentum=0.9, weight_decay=5e-4)
        crit = CrossEntropyLoss()

        def fn():
            opt.zero_grad(); crit(model(hr), y).backward(); opt.step()
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ops._cfg["probe"] = {"all": {}}
    fn()
    torch.cuda.synchronize()
    table = bench.kernel_table(ops._cfg.pop("probe")["all"], n, 1)
rows = [{k: v for k, v in r.items() if k != "_tag"} for r in table]
print(json.dumps({"workload": work, "per_gpu_batch": n, "note": "one probed step, event pairs around every convolution launch on its own stream; "
                  "launches of different streams overlap, so the per-site totals add up to more than the step",
                  "conv_ms_per_step_probed": round(sum(r["total_ms_per_step"] for r in rows), 2), "sites": rows}, indent=1))
