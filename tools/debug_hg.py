import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import numpy as np, torch
import xrface
from xrface.model import FSRnet_sr as M
from oracle import cpu_ref as R, detgen as G
xrface.set_compute_dtype(torch.float32)
dev = "cuda:0"
for depth, size in ((1, 16), (2, 32), (4, 112)):
    hg = M.Hourglass(planes=64, depth=depth, block=M.Bottleneck, num_blocks=3)
    sd = G.det_state_dict(hg.state_dict(), 5); hg.load_state_dict(sd); hg.to(dev)
    x = torch.from_numpy(G.normal(f"x{depth}", 2 * 64 * size * size).reshape(2, 64, size, size).astype(np.float32))
    w = torch.from_numpy(G.normal(f"w{depth}", 2 * 64 * size * size).reshape(2, 64, size, size).astype(np.float32))
    xg = x.to(dev).requires_grad_(True)
    out = hg(xg)
    (out * w.to(dev)).sum().backward()
    sdg = R.with_grad({("hg." + k): v for k, v in sd.items()})
    xr = x.clone().requires_grad_(True)
    o2 = R._sr_hourglass(sdg, "hg.hg", xr, depth)
    (o2 * w).sum().backward() if False else None
    names = [k for k, v in sdg.items() if v.requires_grad]
    gs = torch.autograd.grad((o2 * w).sum(), [sdg[k] for k in names] + [xr])
    print(f"depth {depth}: fwd err {float((out.cpu()-o2).abs().max()/o2.abs().max()):.2e}  dx err {float((xg.grad.cpu()-gs[-1]).abs().max()/gs[-1].abs().max()):.2e}")
    got = dict(hg.named_parameters())
    worst = sorted(((float((got[k[3:]].grad.cpu() - g).abs().max() / max(float(g.abs().max()), 1e-12)), k) for k, g in zip(names, gs[:-1])), reverse=True)[:4]
    print("   worst param grads:", [(f"{e:.2e}", k) for e, k in worst])
