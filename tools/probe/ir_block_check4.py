import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch, copy
import xrface
from xrface import ops
from xrface.model.model_irse import bottleneck_IR_SE
dev = "cuda:0"
torch.manual_seed(21)
c, hw, n = 256, 14, 32
units0 = torch.nn.ModuleList([bottleneck_IR_SE(c, c, 1) for _ in range(3)]).to(dev).train()
x0 = torch.randn(n, c, hw, hw)
def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max())
res = {}
for name, dtype, blk in (("f32", torch.float32, 0), ("ops", torch.bfloat16, 0), ("ops2", torch.bfloat16, 0), ("blk", torch.bfloat16, 1), ("blk2", torch.bfloat16, 1)):
    xrface.set_compute_dtype(dtype)
    ops._cfg["ir_block"] = blk
    units = copy.deepcopy(units0)
    x = x0.to(dev).requires_grad_(True)
    y = ops.enter(x, dtype)
    for u in units:
        y = u.f(y)
    out = ops.leave(y)
    out.float().square().mean().backward()
    torch.cuda.synchronize()
    res[name] = {k: p.grad.float().cpu() for k, p in units.named_parameters()}
    res[name]["x"] = x.grad.cpu()
for k in res["f32"]:
    if "5.fc" in k or k == "x" or "1.weight" in k:
        print(f"{k:28s} ops {rel(res['ops'][k], res['f32'][k]):.4f}  ops2 {rel(res['ops2'][k], res['f32'][k]):.4f}  blk {rel(res['blk'][k], res['f32'][k]):.4f}  blk2 {rel(res['blk2'][k], res['f32'][k]):.4f}   blk-vs-ops {rel(res['blk'][k], res['ops'][k]):.4f}  max|g| {float(res['f32'][k].abs().max()):.2e}")
