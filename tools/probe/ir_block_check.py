import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))
import torch, copy
import xrface
from xrface import ops, parallel
from xrface.loss.loss import CrossEntropyLoss
from xrface.model.model_irse import IR_SE_50
dev = torch.device("cuda:0")
xrface.set_compute_dtype(torch.bfloat16)
torch.manual_seed(0)
net0 = IR_SE_50([112, 112]).to(dev).train()
net0.output_layer[1].p = 0.0
N = int(os.environ.get("N", 16))
x = torch.randn(N, 3, 112, 112, device=dev).clamp_(-1, 1)
y = torch.randint(0, 512, (N,), device=dev)
crit = CrossEntropyLoss()
res = {}
for direct in (False, True):
    for side in (0, 1):
        for mode in (1, 0, 1, 0):
            ops._cfg["ir_block"] = mode
            ops._cfg["wgrad_stream"] = side
            net = copy.deepcopy(net0)
            flat = parallel.FlatParams(net.parameters_in_execution_order(), direct=direct)
            flat.zero_grad()
            crit(net(x), y).backward()
            ops.join_side_stream()
            torch.cuda.synchronize()
            g = flat.grad.clone()
            key = (direct, side, mode)
            if key in res:
                cos = float(torch.nn.functional.cosine_similarity(g, res[key], dim=0))
                print(f"direct={direct} side={side} ir_block={mode}: repeat cosine {cos:.5f}")
            else:
                res[key] = g
        cos = float(torch.nn.functional.cosine_similarity(res[(direct, side, 1)], res[(direct, side, 0)], dim=0))
        print(f"direct={direct} side={side}: block vs ops cosine {cos:.5f}", flush=True)
        # per-parameter worst
        worst = []
        for (n_, p_), o in zip([(n_, p_) for n_, p_ in net.named_parameters()], flat.offsets):
            pass
ops._cfg["ir_block"] = 1; ops._cfg["wgrad_stream"] = 1
