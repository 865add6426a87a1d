#!/usr/bin/env python
"""Package power while the convolution kernels and the headline step run (sampled from sysfs / rocm-smi by a child process that never
touches the GPU runtime): is the sustained MFMA rate a power limit?   python tools/probe/power_watch.py"""
import glob, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "cross-resolution-face-recognition_amd"))

SAMPLER = r'''
import glob, sys, time
def rd(p):
    try:
        return open(p).read().strip()
    except Exception as e:
        return None
cards = sorted(glob.glob("/sys/class/drm/card*/device/pp_dpm_sclk"))
hw = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_average")) + sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/power1_input"))
fq = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input"))
t_end = time.time() + float(sys.argv[1])
while time.time() < t_end:
    fr = [int(rd(f) or 0) / 1e6 for f in fq]
    pw = [int(rd(h) or 0) / 1e6 for h in hw]
    k = max(range(len(fr)), key=lambda i: fr[i]) if fr else -1
    row = [f"{len(fr)} cards; busiest: freq1 {fr[k]:.0f} MHz" if fr else "no freq1_input"]
    if pw:
        row.append("power " + " ".join(f"{v:.0f}" for v in pw) + " W")
    print(f"{time.time():.2f} " + "; ".join(row), flush=True)
    time.sleep(0.25)
'''

import torch
import xrface
from xrface import ops, parallel
from xrface._lib import lib, ptr, stream, dt
import bench
dev = torch.device("cuda:0")
N, H = 128, 112
x = torch.randn(N, H, H, 64, device=dev).bfloat16()
w = torch.randn(64, 64, 3, 3, device=dev) * 0.05
y = torch.empty_like(x)
pk, _ = ops._packed(w, "fwd", torch.bfloat16, 64, 1, 9, 64, 64, 576, 0, 1, 9)
sc = torch.rand(N, 64, device=dev) + 0.5; sh = torch.randn(N, 64, device=dev); al = torch.rand(64, device=dev)
stats = torch.zeros(2, N, 64, device=dev)
big = torch.empty(1 << 28, device=dev, dtype=torch.float32)   # 1 GiB streaming buffer
# the wide layer of IR-SE-50: 256 -> 256 @14 x 14, batch 256 (8-wave implicit GEMM, 8-wave ring weight gradient)
x8 = torch.randn(256, 14, 14, 256, device=dev).bfloat16(); y8 = torch.empty_like(x8); dy8 = torch.randn_like(x8)
w8 = torch.randn(256, 256, 3, 3, device=dev) * 0.02
kg8 = ops.kg_of(9, 256)
pk8, _ = ops._packed(w8, "fwd", torch.bfloat16, 256, 1, 9, 256, 256, kg8, 0, 1, 9)
split8 = ops._wgrad_split(256 * 196, 256, kg8)
slab8 = torch.zeros(split8, 256, kg8, device=dev)
# the headline training step
xrface.set_compute_dtype(torch.bfloat16)
from xrface.loss.loss import CrossEntropyLoss
from xrface.model.model_irse import IR_SE_50
model = IR_SE_50([112, 112]).to(dev).train()
flat = parallel.FlatParams(model.parameters())
opt = parallel.FusedSGD(flat, lr=0.01, momentum=0.9, weight_decay=5e-4)
crit = CrossEntropyLoss()
xb, yb = bench.synth_batch(256, dev, 0)
def c2_step():
    opt.zero_grad(); crit(model(xb), yb).backward(); opt.step()
for _ in range(3):
    c2_step()
torch.cuda.synchronize()
p = subprocess.Popen([sys.executable, "-c", SAMPLER, "30"], stdout=subprocess.PIPE, text=True)
marks = []
def phase(name, fn, secs=3.0, inner=50):
    torch.cuda.synchronize(); t0 = time.time(); n = 0
    while time.time() - t0 < secs:
        for _ in range(inner):
            fn()
        torch.cuda.synchronize(); n += inner
    marks.append((name, t0, time.time(), (time.time() - t0) / max(n, 1) * 1e6))
phase("idle", lambda: None, 2.0)
phase("direct conv 64->64 @112 plain", lambda: lib.xr_conv64_direct(ptr(x), ptr(pk), None, ptr(y), N, H, H, 0, None, None, None, None, None, stream()))
phase("direct conv norm+stats", lambda: lib.xr_conv64_direct(ptr(x), ptr(pk), None, ptr(y), N, H, H, 0, ptr(sc), ptr(sh), ptr(al), ptr(stats), None, stream()))
phase("igemm8 fwd 256->256 @14", lambda: lib.xr_conv_igemm(0, ptr(x8), ptr(pk8), None, ptr(y8), 256, 14, 14, 256, 14, 14, 256, 3, 3, 1, 1, 0, kg8, 256, None, 0, None, None, None, 1, None, None, None, stream()))
phase("wgrad8 256->256 @14", lambda: lib.xr_conv_wgrad(dt(x8), ptr(x8), ptr(dy8), ptr(slab8), 256, 14, 14, 256, 14, 14, 256, 3, 3, 1, 1, 0, 256, kg8, split8, stream()))
phase("streaming copy 0.5 GiB", lambda: big[: 1 << 27].copy_(big[1 << 27:]))
phase("IR-SE-50 training step", c2_step, 4.0, 10)
out = p.communicate()[0].splitlines()
for name, a, b, us in marks:
    rows = [l.split(" ", 1)[1] for l in out if a + 0.5 <= float(l.split(" ", 1)[0]) <= b]
    pw = [[float(v) for v in r.split("power ")[1].split(" W")[0].split()] for r in rows if "power " in r]
    if pw:
        k = max(range(len(pw[0])), key=lambda i: sum(r[i] for r in pw))   # the card this process runs on: the one drawing most
        mine = [r[k] for r in pw]
        print(f"{name:32s} {us:9.1f} us/launch   package power (card {k}): mean {sum(mine) / len(mine):6.0f} W  max {max(mine):6.0f} W  ({len(mine)} samples)")
    else:
        print(f"{name:32s} {us:9.1f} us/launch   no power samples: {rows[:1]}")
