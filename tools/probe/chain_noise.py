#!/usr/bin/env python
"""Noise floor of two chained IR-SE units in the fp32 parity mode: the same unchained backward run twice (the fp32 atomics of
the reductions land in a different order each time) vs chained against unchained."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch
import xrface
from xrface import ops
from xrface.model.model_irse import bottleneck_IR_SE
from xrface.ops import enter, leave

dev = torch.device("cuda:0")
torch.manual_seed(11)
c0, c1, stride, hw, n = 256, 256, 1, 14, 32
u0 = [bottleneck_IR_SE(c0, c0, 1).to(dev).train(), bottleneck_IR_SE(c0, c1, stride).to(dev).train()]
if os.environ.get("DETGEN"):
    sys.path.insert(0, ROOT)
    from oracle import detgen as G   # diagnostic only: the generator of tests/test_gpu_ops.py
    import numpy as np
    x0 = torch.from_numpy(G.normal(f"chain{c0}{c1}", n * c0 * hw * hw).reshape(n, c0, hw, hw).astype(np.float32))
else:
    x0 = torch.randn(n, c0, hw, hw)


def run(chain):
    ops._cfg["chain_units"] = chain
    units = copy.deepcopy(u0)
    x = x0.to(dev).requires_grad_(True)
    out = leave(units[1].f(units[0].f(enter(x))))
    out.square().mean().backward()
    torch.cuda.synchronize()
    names = ["out", "x.grad"] + [f"u{i}.{k}" for i, u in enumerate(units) for k, _ in u.named_parameters()]
    return names, [out.detach().double().cpu(), x.grad.double().cpu()] + [p.grad.double().cpu() for u in units for p in u.parameters()]


rel = lambda a, b: float((a - b).norm() / b.norm().clamp_min(1e-30))
relmax = lambda a, b: float((a - b).abs().max() / b.abs().max().clamp_min(1e-30))
names, a0 = run(0)
_, a1 = run(0)
_, b0 = run(1)
_, b1 = run(1)
print("a pre-activation within 1e-7 of the PReLU kink may change slope between ANY two runs: look at L2, and at where the worst element sits")
print(f"{'tensor':28s} {'unchained twice':>22s} {'chained twice':>22s} {'chained vs un':>22s}   (L2 / max-abs relative)")
for nm, p, q, r, s in zip(names, a0, a1, b0, b1):
    print(f"{nm:28s} {rel(q, p):10.2e} {relmax(q, p):10.2e}  {rel(s, r):10.2e} {relmax(s, r):10.2e}  {rel(r, p):10.2e} {relmax(r, p):10.2e}")
    if relmax(r, p) > 1e-3 or relmax(q, p) > 1e-3:
        for tag, u, v in (("un/un", q, p), ("ch/un", r, p)):
            d = (u - v).abs()
            i = int(d.argmax())
            idx = tuple(int(k) for k in torch.unravel_index(torch.tensor(i), d.shape))
            big = (d > 0.1 * d.max()).nonzero()
            print(f"    {tag}: worst {float(d.max()):.3e} at {idx} (value {float(v.flatten()[i]):.3e}, max {float(v.abs().max()):.3e}); "
                  f"{len(big)} elements above 10% of it; first: {big[:6].tolist()}")
