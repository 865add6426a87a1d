#!/usr/bin/env python
"""In-process A/B of the IR-SE-50 bf16 training step under different tuning knobs (interleaved rounds; one device)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
import xrface
from xrface import parallel, ops
from xrface._lib import lib
from xrface.loss.loss import CrossEntropyLoss
from xrface.model.model_irse import IR_SE_50
import bench

dev = torch.device("cuda:0")
xrface.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
model = IR_SE_50([112, 112]).to(dev).train()
flat = parallel.FlatParams(model.parameters())
opt = parallel.FusedSGD(flat, lr=0.01, momentum=0.9, weight_decay=5e-4)
crit = CrossEntropyLoss()
x, y = bench.synth_batch(int(os.environ.get("N", 256)), dev, 0)


def step():
    opt.zero_grad(); crit(model(x), y).backward(); opt.step()


if os.environ.get("WORKLOAD") == "kd":  # residual KD step (BASELINE config 4 shape): IR-50 teacher + 2 x ResNet-34, N = 64
    from xrface import steps
    from xrface.model import model_irse, resnet
    n = int(os.environ.get("N", 64))
    teacher = model_irse.IR_50([112, 112]).to(dev).eval()
    student, assistant = resnet.ResNet_34().to(dev), resnet.ResNet_34().to(dev)
    so = torch.optim.RMSprop(student.parameters(), lr=1e-4, weight_decay=1e-5)
    ao = torch.optim.RMSprop(assistant.parameters(), lr=1e-4, weight_decay=1e-5)
    xk = bench.synth_batch(n, dev, 0)[0]

    def step():
        so.zero_grad(set_to_none=True); ao.zero_grad(set_to_none=True)
        steps.kd_step(teacher, student, assistant, xk, so, ao)


if os.environ.get("WORKLOAD") == "c4":  # the composed north-star step (BASELINE configs[3]): FHN -> IR-SE-50 student + assistant vs teacher
    from xrface import steps
    from xrface.model import FSRnet, model_irse
    n = int(os.environ.get("N", 256))
    fhn = {"coarse": FSRnet.Course_SR_Network().to(dev), "prior": FSRnet.Prior_Estimation_Network().to(dev),
           "encoder": FSRnet.Fine_SR_Encoder().to(dev), "decoder": FSRnet.Fine_SR_Decoder().to(dev)}
    student, assistant = model_irse.IR_SE_50([112, 112]).to(dev), model_irse.IR_SE_50([112, 112]).to(dev)
    teacher = model_irse.IR_SE_50([112, 112]).to(dev).eval()
    for p_ in teacher.parameters():
        p_.requires_grad_(False)
    fp = [p_ for k in ("coarse", "prior", "encoder", "decoder") for p_ in fhn[k].parameters()]
    flats = [parallel.FlatParams(fp), parallel.FlatParams(student.parameters_in_execution_order()),
             parallel.FlatParams(assistant.parameters_in_execution_order())]
    opts4 = [parallel.FusedRMSprop(f, lr=1e-5, weight_decay=1e-5) for f in flats]
    hr4 = bench.synth_batch(n, dev, 11)[0]
    lr4 = bench.synth_lr(hr4)
    del model, flat, opt

    def step():
        steps.c4_step(fhn, student, assistant, teacher, lr4, hr4, optimizers=opts4)


if os.environ.get("WORKLOAD") == "c3":  # the full FHN step (BASELINE configs[2] per-GPU shape)
    from xrface import steps
    from xrface.model import FSRnet
    n = int(os.environ.get("N", 128))
    fhn = {"coarse": FSRnet.Course_SR_Network().to(dev), "prior": FSRnet.Prior_Estimation_Network().to(dev),
           "encoder": FSRnet.Fine_SR_Encoder().to(dev), "decoder": FSRnet.Fine_SR_Decoder().to(dev)}
    flats3 = {k: parallel.FlatParams(fhn[k].parameters()) for k in fhn}
    opts3 = {k: parallel.FusedRMSprop(flats3[k], lr=1e-5, weight_decay=1e-5) for k in fhn}
    hr3 = bench.synth_batch(n, dev, 12)[0]
    lr3 = bench.synth_lr(hr3)
    hm3 = torch.rand(n, 28, 28, device=dev)
    par3 = torch.randint(0, 11, (n, 1, 28, 28), device=dev)
    del model, flat, opt

    def step():
        steps.fhn_step_fused(fhn, lr3, hr3, hm3, par3, optimizers=opts3)


# VARIANTS="name:knob=val,knob=val;..." ; special key: wb = ops wgrad_blocks
variants = []
for spec in os.environ.get("VARIANTS", "base:;wprio:5=1").split(";"):
    name, kv = spec.split(":")
    variants.append((name, [tuple(p.split("=")) for p in kv.split(",") if p]))
cfg0 = dict(ops._cfg)
defaults = {0: 3, 2: 0, 3: 0, 4: 1, 5: 1, 6: 18, 7: 1, 11: 0, 12: 1, 13: 1}


def apply(kvs):
    for k, v in defaults.items():
        lib.xr_tune(k, v)
    ops._cfg["wgrad_blocks"] = 512
    ops._cfg.update(cfg0)
    for k, v in kvs:
        if k.startswith("cfg."):
            ops._cfg[k[4:]] = int(v)
        elif k == "wb":
            ops._cfg["wgrad_blocks"] = int(v)
        else:
            lib.xr_tune(int(k), int(v))


_ctx = None
if os.environ.get("HIGH_PRIO"):   # run the whole step on a high-priority stream: side-stream work then yields to it
    _hp = torch.cuda.Stream(priority=-1)
    _hp.wait_stream(torch.cuda.current_stream())
    _ctx = torch.cuda.stream(_hp)
    _ctx.__enter__()
for _ in range(3):
    step()
res = {n: [] for n, _ in variants}
cpu = {}
for rnd in range(int(os.environ.get("ROUNDS", 4))):
    for name, kvs in variants:
        apply(kvs)
        for _ in range(int(os.environ.get('SWITCH_WARM', 3))):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            step()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        res[name].append((time.perf_counter() - t0) / 4 * 1e3)
        cpu.setdefault(name, []).append((t1 - t0) / 4 * 1e3)
for name, v in res.items():
    v = sorted(v)
    c = sorted(cpu[name])
    print(f"{name:12s} median {v[len(v)//2]:.3f} ms  min {v[0]:.3f}  max {v[-1]:.3f}   (host enqueue {c[len(c)//2]:.3f} ms/step)")
if os.environ.get("GRAPH"):   # the same step as ONE single-stream HIP-graph replay (no lockstep streams, no side stream inside a capture)
    from xrface.graph import GraphedStep
    apply([])
    gs = GraphedStep(lambda: step(), [], warmup=2, side_stream=bool(os.environ.get("GRAPH_SIDE")))   # GRAPH_SIDE=1: keep the fork / join
    for _ in range(3):
        gs()
    ts = []
    for rnd in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            gs()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) / 4 * 1e3)
    ts.sort()
    print(f"{'graph+side' if os.environ.get('GRAPH_SIDE') else 'graph':12s} median {ts[len(ts)//2]:.3f} ms  min {ts[0]:.3f}  max {ts[-1]:.3f}", flush=True)
    gs.close()
