#!/usr/bin/env python
"""Is the full-FHN step (C3, N = 128) held back by the host?  Eager step time vs the host's enqueue time for it, and the same
step captured into ONE HIP graph (single stream: no side-stream overlap for the weight gradients, but no launch gaps either)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
import xrface
from xrface import ops, parallel, steps
from xrface.graph import GraphedStep
from xrface.model import FSRnet
import bench
dev = torch.device("cuda:0")
xrface.set_compute_dtype(torch.bfloat16)
N = int(os.environ.get("N", 128))
torch.manual_seed(0)
fhn = {"coarse": FSRnet.Course_SR_Network().to(dev), "prior": FSRnet.Prior_Estimation_Network().to(dev),
       "encoder": FSRnet.Fine_SR_Encoder().to(dev), "decoder": FSRnet.Fine_SR_Decoder().to(dev)}
flats = {k: parallel.FlatParams(fhn[k].parameters()) for k in fhn}
opts = {k: parallel.FusedRMSprop(flats[k], lr=1e-5, alpha=0.99, weight_decay=1e-5) for k in fhn}
hr, _ = bench.synth_batch(N, dev, 12)
lr = bench.synth_lr(hr)
hm = torch.rand(N, 28, 28, device=dev)
par = torch.randint(0, 11, (N, 1, 28, 28), device=dev)
lbuf = torch.zeros(4, device=dev)


def step(lr_, hr_, hm_, par_):
    l_, _ = steps.fhn_step_fused(fhn, lr_, hr_, hm_, par_, optimizers=opts)
    lbuf.copy_(torch.stack([l_[k].float() for k in ("coarse", "encoder", "prior", "decoder")]))
    return lbuf


def timed(fn, reps=6):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    hs, ts = [], []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3); hs.append((t1 - t0) * 1e3)
    ts.sort(); hs.sort()
    return ts[len(ts) // 2], hs[len(hs) // 2]


hp = torch.cuda.Stream(dev, priority=-1)
hp.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(hp):
    ms, host = timed(lambda: step(lr, hr, hm, par))
    print(f"C3 N={N} eager (side stream on): {ms:.2f} ms per step, host enqueue {host:.2f} ms", flush=True)
    ops._cfg["wgrad_stream"] = 0
    ms, host = timed(lambda: step(lr, hr, hm, par))
    print(f"C3 N={N} eager single stream:    {ms:.2f} ms per step, host enqueue {host:.2f} ms", flush=True)
    ops._cfg["wgrad_stream"] = 1
    gs = GraphedStep(step, [lr, hr, hm, par], warmup=2)
    ms, host = timed(lambda: gs(lr, hr, hm, par))
    print(f"C3 N={N} one HIP graph (single stream): {ms:.2f} ms per step, host {host:.2f} ms", flush=True)
    gs.close()
    # sub-networks alone, eager: forward + backward of a sum of outputs
    for name in ("prior", "encoder", "coarse", "decoder"):
        net = fhn[name]
        xin = torch.cat((torch.randn(N, 128, 28, 28, device=dev), torch.randn(N, 64, 28, 28, device=dev)), 1) if name == "decoder" else hr

        def sub():
            for o in opts.values():
                o.zero_grad()
            out = net(xin)
            out = out if isinstance(out, (tuple, list)) else (out,)
            sum(o_.float().square().mean() for o_ in out).backward()
        ms, host = timed(sub, 4)
        print(f"   {name:8s} fwd + bwd alone: {ms:.2f} ms, host enqueue {host:.2f} ms", flush=True)
