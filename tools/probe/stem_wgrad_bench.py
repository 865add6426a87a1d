import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "cross-resolution-face-recognition_amd"))
import torch
from xrface import ops
from xrface._lib import lib, ptr, stream, dt
dev = torch.device("cuda:0")
N, C, K, H = 256, 8, 64, 112
x = torch.randn(N, H, H, C, device=dev).bfloat16(); dy = torch.randn(N, H, H, K, device=dev).bfloat16()
kg = ops.kg_of(9, C); split = ops._wgrad_split(N * H * H, K, kg)
slab = torch.zeros(2048, K, kg, device=dev)
def timeit(fn, reps=10):
    for _ in range(2): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / reps
for split in (256, 512, 1024, 2048):
    ms = timeit(lambda: lib.xr_conv_wgrad(dt(x), ptr(x), ptr(dy), ptr(slab), N, H, H, C, H, H, K, 3, 3, 1, 1, 0, K, kg, split, stream()))
    print(os.environ.get("XR_STEM_TALL"), "split", split, f"{ms*1e3:.1f} us")
ref = torch.nn.grad.conv2d_weight(x[:2].float().permute(0,3,1,2), (K, C, 3, 3), dy[:2].float().permute(0,3,1,2), stride=1, padding=1)
print(os.environ.get("XR_STEM_TALL"), "kg", kg, "split", split, f"{ms*1e3:.1f} us")
