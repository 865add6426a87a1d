import os, sys
sys.path.insert(0, "/root/repo/cross-resolution-face-recognition_amd")
import torch
from xrface import ops
from xrface._lib import lib, ptr, stream
dev = torch.device("cuda:0")
def timeit(fn, reps=10):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
N, H = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 112
x = torch.randn(N, H, H, 64, device=dev).bfloat16(); w = torch.randn(64, 64, 3, 3, device=dev) * 0.05; y = torch.empty_like(x)
pk, _ = ops._packed(w, "fwd", torch.bfloat16, 64, 1, 9, 64, 64, 576, 0, 1, 9)
for dbg in (0, 1, 2, 4, 5, 6, 7):
    lib.xr_tune(14, dbg)
    a = timeit(lambda: lib.xr_conv64_direct(ptr(x), ptr(pk), None, ptr(y), N, H, H, 0, None, None, None, None, None, stream()))
    print(f"dbg={dbg:2d}: {a*1e3:7.1f} us  ({a*1e3/(N*49/256):.2f} us/tile)")
lib.xr_tune(14, 0)
