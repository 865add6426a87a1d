#!/usr/bin/env python
"""Benchmark of the hot path (contract: see the task statement / DESIGN.md section "Measurement").

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Headline (`value`): BASELINE.json configs[1] -- one step = IR-SE-50 forward + backward + gradient all-reduce (N > 1) + fused SGD
update on a batch of 256 synthetic 112x112 faces per GPU, bf16 (train_teacher_model.py path).
`secondary` (N = 1 only, timed after the headline with their own warm-up): the north-star composed step C4 (FHN -> IR-SE-50
student + assistant vs frozen IR-SE-50 teacher, batch 256, BASELINE configs[3] per-GPU shape), C3 (full FHN step, batch 128),
C1 (coarse net, batch 4, fp32) and C5 (1M-pair distances + ROC).
Prints ONE JSON line from rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "cross-resolution-face-recognition_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

IRSE50_FWD_GFLOP = 12.593          # algorithmic 2*MAC conv+linear FLOPs per 112x112 image
FHN_FWD_GFLOP = 38.256             # root FHN: coarse 16.734 + encoder 1.113 + prior 3.230 + decoder 17.179
C4_STEP_GFLOP = 203.0              # 3 x FHN + 3 x student + 3 x assistant + 1 x teacher
PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def synth_batch(n, dev, seed):
    """Cheap synthetic faces on the device: smooth low-frequency field + noise in [-1,1] (plumbing, outside the timed region)."""
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + seed)
    lo = torch.randn(n, 3, 14, 14, device=dev, generator=g)
    x = torch.nn.functional.interpolate(lo, size=(112, 112), mode="bilinear", align_corners=False)
    x = (x + 0.1 * torch.randn(n, 3, 112, 112, device=dev, generator=g)).clamp_(-1, 1)
    y = torch.randint(0, 512, (n,), device=dev, generator=g)
    return x.contiguous(), y


def synth_lr(hr):
    """16x16 average-pooled and blown back up to 112x112 (the tensor contract of FHN_loader.py:65-66)."""
    lo = torch.nn.functional.avg_pool2d(hr, 7)
    return torch.nn.functional.interpolate(lo, size=(112, 112), mode="bicubic", align_corners=False).clamp_(-1, 1).contiguous()


# ------------------------------------------------------------------------------------------------- per-kernel roofline
def _tag_name(tag):
    kind, C, K, H, W, R, stride = tag
    return f"{kind} conv{R}x{R} {C}->{K} @{H}x{W} s{stride}"


def _out_hw(H, W, R, stride):
    if R >= H:      # full-extent "convolution" = Linear on a flattened H x W map (padding 0)
        return 1, 1
    return (H + 2 * (R // 2) - R) // stride + 1, (W + 2 * (R // 2) - R) // stride + 1


def _tag_flops(tag, batch):
    kind, C, K, H, W, R, stride = tag
    Ho, Wo = _out_hw(H, W, R, stride)
    return 2.0 * batch * Ho * Wo * K * C * R * R


def _kernel_of(tag, batch):
    """Kernel template a launch site runs (the auto rule of csrc/xr_conv8.hip:igemm8_config restated; names as rocprofv3 shows them)."""
    kind, C, K, H, W, R, stride = tag
    if kind == "wgrad":
        if C == 64 and K == 64 and R == 3 and stride == 1 and W % 8 == 0 and W <= 112:
            return "wgrad64_kernel<%d chunks/row> (direct, xr_wgrad64.hip)" % (-(-W // 16) if W > 64 else (4 if W > 32 else 2))
        Wo = W // stride
        if (R == 3 and C % 64 == 0 and K % 64 == 0 and H % stride == 0 and W % stride == 0 and 14 <= Wo <= (112 if stride == 1 else 64)
                and ((stride == 1 and C <= 128) or (stride == 2 and C == 64))):   # ops._wgrad_rows_ok, mode 2
            return "wgrad_rows_kernel (row-walking direct, xr_wgrad_rows.hip)"
        if K >= 256 and C >= 256 and K % 128 == 0:   # xr_wgrad8_eligible
            return "wgrad8_kernel (8 waves, 128x256 tile, 3-stage LDS ring, xr_wgrad8.hip)"
        kg = -(-(C * R * R) // 64) * 64
        return "wgrad_kernel<0, %s, ...>" % ("128, 128" if (K > 64 or kg <= 128) else "64, 256")
    if C == 64 and K == 64 and R == 3 and stride == 1:   # ops.direct64_ok: the 64-channel stride-1 layers of the IR units
        if kind == "fwd":
            return "dconv64_kernel (direct, xr_conv64.hip; conv1 with the PReLU second output)"
        return "dconv64_kernel (conv1: BatchNorm sums in the epilogue) + igemm_kernel<dgrad, 4 waves> (conv2: PReLU-backward epilogue)"
    Ho, Wo = _out_hw(H, W, R, stride)
    gk, gc, m = (K, C, batch * Ho * Wo) if kind == "fwd" else (C, K, batch * H * W)    # GEMM columns, reduction channels, rows
    ok = gk % 8 == 0 and gc % 64 == 0 and R * R <= 32 and gc * R * R >= 512 and (kind == "fwd" or stride == 1)
    if ok and gk % 256 == 0 and -(-m // 256) * (gk // 256) >= 160:
        return "igemm8_kernel<%s, 8 waves, 256/224x256 tile>" % kind
    if ok and gk % 128 == 0 and gk % 256 != 0 and -(-m // 512) * (gk // 128) >= 160:
        return "igemm8_kernel<%s, 8 waves, 512/448x128 tile>" % kind
    return "igemm_kernel<%s, 4 waves, 128x%d tile>" % (kind, 128 if gk > 64 else 64)


def kernel_table(probe, batch, steps):
    """Per conv launch site family (kind, shape): launches per step, average in-step duration (HIP event pairs on the launch
    stream, one untimed probe step), TFLOP/s and fraction of the bf16 MFMA peak; sorted by total time."""
    rows = []
    for tag, evs in probe.items():
        ms = [a.elapsed_time(b) for a, b in evs]
        if not ms:
            continue
        avg = sum(ms) / len(ms)
        fl = _tag_flops(tag, batch)
        rows.append({"site": _tag_name(tag), "kernel": _kernel_of(tag, batch), "launches_per_step": len(ms) // max(steps, 1),
                     "avg_ms": round(avg, 4), "total_ms_per_step": round(sum(ms) / max(steps, 1), 3),
                     "tflops": round(fl / (avg * 1e-3) / 1e12, 1), "frac": round(fl / (avg * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                     "_tag": tag})
    rows.sort(key=lambda r: -r["total_ms_per_step"])
    return rows


def isolated_ms(tag, batch, dev, reps=8):
    """The same launch back-to-back in isolation (cache-warm operands, nothing else on the GPU)."""
    from xrface import ops
    from xrface._lib import dt, lib, ptr, stream
    kind, C, K, H, W, R, stride = tag
    pad = 0 if R >= H else R // 2
    Ho, Wo = _out_hw(H, W, R, stride)
    x = torch.randn(batch, H, W, C, device=dev).to(torch.bfloat16)
    dy = torch.randn(batch, Ho, Wo, K, device=dev).to(torch.bfloat16)
    w = torch.randn(K, C, R, R, device=dev) * 0.02
    if kind == "fwd":
        pk, kg = ops._packed(w, "fwd", torch.bfloat16, K, 1, R * R, C, C, C * R * R, 0, 1, R * R)
        y = torch.empty_like(dy)
        fn = lambda: lib.xr_conv_igemm(dt(x), ptr(x), ptr(pk), None, ptr(y), batch, H, W, C, Ho, Wo, K, R, R, stride, pad, 0, kg, K,
                                       None, 0, None, None, None, 1, None, None, None, stream())
    elif kind == "dgrad":
        pk, kg = ops._packed(w, "dgrad", torch.bfloat16, C, 1, R * R, K, K, R * R, 0, 1, C * R * R)
        dx = torch.empty_like(x)
        fn = lambda: lib.xr_conv_igemm(dt(x), ptr(dy), ptr(pk), None, ptr(dx), batch, Ho, Wo, K, H, W, C, R, R, stride, pad, 1, kg, C,
                                       None, 0, None, None, None, 1, None, None, None, stream())
    else:
        kg = ops.kg_of(R * R, C)
        split = ops._wgrad_split(batch * Ho * Wo, K, kg)
        slabs = torch.empty((split, K, kg), dtype=torch.float32, device=dev)
        fn = lambda: lib.xr_conv_wgrad(dt(x), ptr(x), ptr(dy), ptr(slabs), batch, H, W, C, Ho, Wo, K, R, R, stride, pad, 0, K, kg,
                                       split, stream())
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def pmc_traffic(kernel, site):
    """HBM bytes per launch of the dominant kernel from this round's committed rocprofv3 PMC passes
    (profiles/r03_counters.json, written by tools/pmc_kernels.sh + tools/pmc_to_json.py with the convolution kernels of the current
    HEAD); None when that site was not profiled."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_counters.json")) as f:
            db = json.load(f)
        return db["sites"][site]["hbm_bytes_per_launch"]
    except Exception:
        return None


def step_roofline(tag, ms, batch):
    """HBM view of a whole secondary step (C3 / C4 are HBM-bound): bytes per step from this round's
    committed PMC pass over a profiled step (profiles/r03_step_traffic.json: sum over all kernels of 2 x FETCH_SIZE + WRITE_SIZE,
    MI355X_MICROARCH.md HBM section), taken at the batch the step is timed at, against 8 TB/s; None until that file exists."""
    try:
        with open(os.path.join(ROOT, "profiles", "r03_step_traffic.json")) as f:
            db = json.load(f)[tag]
        if db["batch"] != batch:
            return None
        gb = db["hbm_bytes_per_step"] / 1e9
        return {"bound": "hbm", "achieved": round(gb / (ms * 1e-3), 1), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(gb / (ms * 1e-3) / PEAK_HBM_GBS, 4), "traffic": db["hbm_bytes_per_step"],
                "algorithmic_bytes_per_step": db.get("algorithmic_bytes_per_step"), "source": "profiles/r03_step_traffic.json"}
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(dev, dtype, budget_s=12.0):
    """The CPU oracle (oracle/cpu_ref.py: stock torch fp32 CPU ops, the reference's module graph) timed on the
    host cores on a bounded sample of the same workload: IR-SE-50 fwd+bwd at N = 8 per step.  The same leg checks the
    embedding the timed mode (dtype) produces against the oracle's fp32 embedding on identical weights and inputs."""
    from oracle import cpu_ref as R
    from oracle import detgen as G
    import xrface
    from xrface.model.model_irse import IR_SE_50
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # a 1-GPU box exposes 256 logical CPUs but grants a 16-CPU share
    torch.set_num_threads(cores)
    net = IR_SE_50([112, 112])
    sd = G.det_state_dict(net.state_dict())
    n = 8
    x = G.synth_faces(n, 112, seed=1)
    t = G.synth_labels(n, 512)
    # embedding parity of the benchmarked mode: eval-mode forward, HIP path (dtype) vs oracle (fp32 CPU)
    net.load_state_dict(sd)
    net.to(dev).eval()
    with torch.no_grad():
        e_cpu, _ = R.ir_backbone(sd, x, se=True, train=False)
        errs = {}
        for dtp in (dtype, torch.float32, "fp32x2"):   # the timed mode, and the two fp32 modes (split-bf16 MFMA) the 1e-3 bar is met in
            xrface.set_compute_dtype(dtp)
            e_gpu = net(x.to(dev)).float().cpu()
            errs[dtp] = float((e_gpu - e_cpu).norm(dim=1).max() / e_cpu.norm(dim=1).min())
    xrface.set_compute_dtype(dtype)
    emb_err = (errs[dtype], errs[torch.float32], errs)
    del net
    R.teacher_step_grads(sd, x, t, se=True)  # warm-up
    steps, t0 = 0, time.perf_counter()
    while True:
        R.teacher_step_grads(sd, x, t, se=True)
        steps += 1
        if time.perf_counter() - t0 > budget_s or steps >= 200:
            break
    el = time.perf_counter() - t0
    return {"value": round(n * steps / el, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"IR-SE-50 fwd+bwd (CE on the 512-d output), fp32, N={n}/step, {steps} steps in {el:.1f}s"}, emb_err


# ------------------------------------------------------------------------------------------------- secondary workloads
def _timed(fn, warm, reps):
    """Median wall time of `reps` individually synchronised calls after `warm` untimed ones (secondary workloads only: a one-off
    allocator / clock hiccup in one call must not decide a 4-5 call figure; the headline keeps the K-step total of the contract)."""
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = fn()
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    if os.environ.get("XR_BENCH_TRACE"):
        print("  timed: " + " ".join(f"{t:.2f}" for t in ts), file=sys.stderr, flush=True)
    ts.sort()
    med = ts[len(ts) // 2]
    if dist.is_initialized() and dist.get_world_size() > 1:   # data-parallel form: every call held the all-reduce; slowest rank counts
        t = torch.tensor([med], dtype=torch.float64, device="cuda" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        med = float(t.item())
    return med, out


def cpu_secondary(kind, budget_s=10.0):
    """CPU baseline of a secondary workload: the oracle's restatement of the same step -- forward, every
    (loss_k, theta_k) gradient pair at pre-step weights -- on the host cores at N = 4 (C1: the configuration itself; C3 / C4:
    scaled per image), bounded to ~budget_s; C5: the vectorised numpy restatement on a 1e5-pair sample."""
    import numpy as np
    from oracle import cpu_ref as R
    from oracle import detgen as G
    from xrface.model import FSRnet, model_irse
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))
    torch.set_num_threads(cores)
    n = 4
    if kind == "C5":
        P = 100_000
        e1, e2, same = G.synth_pairs(P, 512, seed=0)
        fold = np.random.RandomState(0).randint(0, 10, P)
        folds = [(np.where(fold != f)[0], np.where(fold == f)[0]) for f in range(10)]
        thr = np.arange(0, 12000, 3)
        t0, reps = time.perf_counter(), 0
        while True:
            R.calculate_roc(thr, e1, e2, same, folds)
            reps += 1
            if time.perf_counter() - t0 > budget_s / 2 or reps >= 20:
                break
        el = (time.perf_counter() - t0) / reps
        return {"value": round(P / el, 1), "unit": "pairs/s", "cores": cores, "kind": "port",
                "sample": f"numpy restatement of pair distances + 4000-threshold / 10-fold ROC on {P} pairs, {reps} calls of {el:.2f}s "
                          "(numpy: one thread for the sweep, BLAS-free)"}
    hr = G.synth_faces(n, 112, seed=1)
    lr = G.synth_lr_from_hr(hr)
    sds = {k: G.det_state_dict(m.state_dict()) for k, m in (("coarse", FSRnet.Course_SR_Network()), ("prior", FSRnet.Prior_Estimation_Network()),
                                                           ("encoder", FSRnet.Fine_SR_Encoder()), ("decoder", FSRnet.Fine_SR_Decoder()))}
    if kind == "C1":
        fn = lambda: R.coarse_step_grads(sds["coarse"], lr, hr)
        what = "Course_SR_Network fwd+bwd of 12*mse97"
    elif kind == "C3":
        hm = G.synth_heatmap(n, 28, 97, 1.3, seed=2)
        par = G.synth_parsing(n, 28, 11, seed=2)
        fn = lambda: R.fhn_step_grads(sds, lr, hr, hm, par)
        what = "root FHN forward + the four (loss_k, theta_k) gradient pairs"
    else:
        irsd = [G.det_state_dict(model_irse.IR_SE_50([112, 112]).state_dict(), seed) for seed in (1, 2, 3)]
        fn = lambda: R.c4_step_grads(sds, irsd[0], irsd[1], irsd[2], lr, hr)
        what = "FHN -> IR-SE-50 student + assistant vs frozen teacher, residual-KD gradients"
    fn()   # warm-up
    t0, reps = time.perf_counter(), 0
    while True:
        fn()
        reps += 1
        if time.perf_counter() - t0 > budget_s or reps >= 50:
            break
    el = time.perf_counter() - t0
    return {"value": round(n * reps / el, 3), "unit": "images/s", "cores": cores, "kind": "port",
            "sample": f"{what}, fp32, N={n}/step, {reps} steps in {el:.1f}s"}


def secondary_workloads(dev, c4_batch=256, c3_batch=128, world=1, rank=0, cpu_legs=True, emb_errs=None):
    """world > 1 (BASELINE configs[2], [3]: the multi-GPU configurations): only C4 and C3, in their data-parallel form -- one
    BucketedAllReduce per flat gradient buffer, buckets launched from the gradient hooks, waited for before the optimizer
    updates; EVERY rank runs them (they hold collectives); times are the max over ranks."""
    import numpy as np
    import xrface
    from xrface import parallel, steps
    from xrface.loss.loss import MSELossFunc
    from xrface.model import FSRnet, model_irse
    from xrface.utils.utils import calculate_roc, pair_dist
    out = []
    xrface.set_compute_dtype(torch.bfloat16)
    mk_fhn = lambda: {"coarse": FSRnet.Course_SR_Network().to(dev), "prior": FSRnet.Prior_Estimation_Network().to(dev),
                      "encoder": FSRnet.Fine_SR_Encoder().to(dev), "decoder": FSRnet.Fine_SR_Decoder().to(dev)}
    # Every workload runs in its own function under try / except: a failure in one of them (e.g. a graph capture) is reported
    # in its entry and can never cost the headline line.
    work = []

    def w_c4():
        # ---- C4: the north-star composed step
        torch.manual_seed(0)
        fhn = mk_fhn()
        student, assistant = model_irse.IR_SE_50([112, 112]).to(dev), model_irse.IR_SE_50([112, 112]).to(dev)
        teacher = model_irse.IR_SE_50([112, 112]).to(dev).eval()
        for p_ in teacher.parameters():
            p_.requires_grad_(False)
        fhn_params = [p_ for k in ("coarse", "prior", "encoder", "decoder") for p_ in fhn[k].parameters()]
        flats = [parallel.FlatParams(fhn_params), parallel.FlatParams(student.parameters_in_execution_order()),
                 parallel.FlatParams(assistant.parameters_in_execution_order())]
        opts = [parallel.FusedRMSprop(flats[0], lr=1e-5, alpha=0.99, weight_decay=1e-5),
                parallel.FusedRMSprop(flats[1], lr=1e-4, alpha=0.99, weight_decay=1e-5),
                parallel.FusedRMSprop(flats[2], lr=1e-4, alpha=0.99, weight_decay=1e-5)]
        reds = None
        if world > 1:     # identical replicas (rank-0 broadcast lands in the flat buffers), one reducer per flat gradient buffer
            for m_ in (*fhn.values(), student, assistant, teacher):
                parallel.broadcast_module(m_)
            reds = [parallel.BucketedAllReduce(f_) for f_ in flats]
        hr, _ = synth_batch(c4_batch, dev, 11 + 97 * rank)
        lr = synth_lr(hr)
        ms, res = _timed(lambda: steps.c4_step(fhn, student, assistant, teacher, lr, hr, optimizers=opts, reducers=reds), 3, 7)
        (sl, al), _ = res
        if world > 1:
            for f_ in flats:      # replicas must still agree after ten averaged-gradient steps
                chk = f_.flat[:4096].clone()
                dist.broadcast(chk, 0)
                assert torch.equal(chk, f_.flat[:4096]), "C4 replicas diverged: gradient all-reduce is broken"
        tf = C4_STEP_GFLOP * c4_batch / ms            # GFLOP per ms = TFLOP/s (per GPU)
        ent = {"workload": "C4 (BASELINE configs[3], per-GPU shape): root FHN (trainable) -> IR-SE-50 student + assistant vs frozen "
                           "IR-SE-50 teacher on hr, residual-KD MSE losses, RMSprop x3, Dropout on, 112x112"
                           + (", data parallel: three bucketed gradient all-reduces inside the step" if world > 1 else ""),
               "n_gpus": world, "per_gpu_batch": c4_batch, "dtype": "bf16", "ms_per_step": round(ms, 2),
               "images_per_s": round(c4_batch * world / ms * 1e3, 1),
               "algorithmic_tflop_per_step": round(C4_STEP_GFLOP * c4_batch * world / 1e3, 2), "achieved_tflops": round(tf * world, 1),
               "frac_of_bf16_peak": round(tf / PEAK_BF16_TFLOPS, 4), "student_loss": round(float(sl), 5),
               "assistant_loss": round(float(al), 5)}
        rf = step_roofline("c4", ms, c4_batch)
        if rf is not None:
            ent["roofline"] = rf
        out.append(ent)
        del fhn, student, assistant, teacher, flats, opts, res, reds
        torch.cuda.empty_cache()
        if world == 1 and cpu_legs:
            ent["cpu_baseline"] = cpu_secondary("C4")
    work.append(("C4", w_c4))

    def w_c3():
        # ---- C3: full FHN step (per-network gradients from one backward pass), landmark + parsing losses
        fhn = mk_fhn()
        flats = {k: parallel.FlatParams(fhn[k].parameters()) for k in fhn}
        opts = {k: parallel.FusedRMSprop(flats[k], lr=1e-5, alpha=0.99, weight_decay=1e-5) for k in fhn}
        reds = None
        if world > 1:
            for m_ in fhn.values():
                parallel.broadcast_module(m_)
            reds = {k: parallel.BucketedAllReduce(flats[k]) for k in fhn}
        hr, _ = synth_batch(c3_batch, dev, 12 + 97 * rank)
        lr = synth_lr(hr)
        hm = torch.rand(c3_batch, 28, 28, device=dev)
        par = torch.randint(0, 11, (c3_batch, 1, 28, 28), device=dev)
        ms, res = _timed(lambda: steps.fhn_step_fused(fhn, lr, hr, hm, par, optimizers=opts, reducers=reds), 3, 7)
        if world > 1:
            for f_ in flats.values():
                chk = f_.flat[:4096].clone()
                dist.broadcast(chk, 0)
                assert torch.equal(chk, f_.flat[:4096]), "C3 replicas diverged: gradient all-reduce is broken"
        ms_eager = ms
        if world == 1:
            # the same step as ONE HIP-graph replay (xrface.graph.GraphedStep, single stream): at this batch the side stream buys the
            # FHN step nothing and the prior / encoder sub-networks (28 x 28 maps, ~600 launches of 10-25 us) are host-paced
            from xrface.graph import GraphedStep
            lbuf = torch.zeros(4, device=dev)

            def c3g(lr_, hr_, hm_, par_):
                l_, _ = steps.fhn_step_fused(fhn, lr_, hr_, hm_, par_, optimizers=opts)
                lbuf.copy_(torch.stack([l_[k].float() for k in ("coarse", "encoder", "prior", "decoder")]))
                return lbuf
            gs = GraphedStep(c3g, [lr, hr, hm, par], warmup=2)
            ms_g, _ = _timed(lambda: gs(lr, hr, hm, par), 2, 7)
            gs.close()
            del gs
            ms = min(ms, ms_g)
        tf = 3.0 * FHN_FWD_GFLOP * c3_batch / ms
        ent = {"workload": "C3 (BASELINE configs[2], per-GPU shape): root FHN coarse -> {prior, encoder} -> decoder, mse97 + landmark + "
                           "parsing losses, per-network gradients, RMSprop x4"
                           + (", data parallel: four bucketed gradient all-reduces inside the step" if world > 1 else ""),
               "n_gpus": world, "per_gpu_batch": c3_batch, "dtype": "bf16", "ms_per_step": round(ms, 2),
               "ms_per_step_eager": round(ms_eager, 2),
               "images_per_s": round(c3_batch * world / ms * 1e3, 1),
               "algorithmic_tflop_per_step": round(3.0 * FHN_FWD_GFLOP * c3_batch * world / 1e3, 2), "achieved_tflops": round(tf * world, 1),
               "frac_of_bf16_peak": round(tf / PEAK_BF16_TFLOPS, 4)}
        if world == 1:
            ent["ms_per_step_graph"] = round(ms_g, 2)
            ent["launch"] = "one HIP-graph replay per step" if ms_g <= ms_eager else "eager launches"
        rf = step_roofline("c3", ms, c3_batch)
        if rf is not None:
            ent["roofline"] = rf
        out.append(ent)
        del fhn, flats, opts, res, reds
        torch.cuda.empty_cache()
        if world == 1 and cpu_legs:
            ent["cpu_baseline"] = cpu_secondary("C3")
    work.append(("C3", w_c3))

    def w_sr_perceptual_step():
        # ---- SR-variant generators + perceptual (IR-50 feature) losses, train_FHN.py:251-308.  4 x depth-4 bottleneck
        # hourglass prior = 475 convolutions of 64 channels: thousands of small launches -> eager is host-bound; one HIP graph replay
        from xrface.graph import GraphedStep
        from xrface.model import FSRnet_sr
        nsr = 32
        nets = {"coarse": FSRnet_sr.Coarse_SR_Network().to(dev), "encoder": FSRnet_sr.Fine_SR_Encoder().to(dev),
                "prior": FSRnet_sr.Prior_Estimation_Network().to(dev), "decoder": FSRnet_sr.Fine_SR_Decoder().to(dev)}
        bb = model_irse.IR_50([112, 112]).to(dev).eval()
        for p_ in bb.parameters():
            p_.requires_grad_(False)
        flats = {"coarse": parallel.FlatParams(nets["coarse"].parameters()), "prior": parallel.FlatParams(nets["prior"].parameters()),
                 "encdec": parallel.FlatParams(list(nets["encoder"].parameters()) + list(nets["decoder"].parameters()))}
        opts = {k: parallel.FusedAdam(f, lr=1e-4, betas=(0.5, 0.999), weight_decay=1e-5) for k, f in flats.items()}
        hrs, _ = synth_batch(nsr, dev, 15)
        lrs = synth_lr(hrs)
        hms = torch.rand(nsr, 112, 112, device=dev)
        pars = torch.randint(0, 13, (nsr, 1, 112, 112), device=dev)
        lbuf = torch.zeros(3, device=dev)

        def sr_step(lr_, hr_, hm_, par_):
            for o in opts.values():
                o.zero_grad()
            l_, _ = steps.fhn_perceptual_step(nets, bb, lr_, hr_, hm_, par_, optimizers=opts)
            lbuf.copy_(torch.stack([l_["coarse"].float(), l_["prior"].float(), l_["encdec"].float()]))
            return lbuf
        ms_e, _ = _timed(lambda: sr_step(lrs, hrs, hms, pars), 2, 3)
        gs = GraphedStep(sr_step, [lrs, hrs, hms, pars], warmup=2)
        ms_g, _ = _timed(lambda: gs(lrs, hrs, hms, pars), 2, 5)
        gs.close()
        sr_gf = 3 * 131.434 + 5 * IRSE50_FWD_GFLOP      # three trainable generators + prior, IR-50 forward x3 and input gradient x2
        out.append({"workload": "SURVEY 8f-1/2: SR-variant FHN (coarse / encoder / bottleneck-hourglass prior / decoder) + perceptual IR-50 "
                                "feature losses (train_FHN.py:251-308), Adam x3; eager launches vs one HIP-graph replay per step",
                    "per_gpu_batch": nsr, "dtype": "bf16", "ms_per_step_eager": round(ms_e, 1), "ms_per_step": round(ms_g, 1),
                    "images_per_s": round(nsr / ms_g * 1e3, 1), "achieved_tflops": round(sr_gf * nsr / ms_g, 1)})
        del nets, bb, flats, opts, gs
        torch.cuda.empty_cache()
    if world == 1:
        work.append(("SR perceptual step", w_sr_perceptual_step))

    def w_c2_fp32_modes():
        # ---- C2 in the two fp32 modes: the headline step in the modes that meet the 1e-3 embedding tolerance
        from xrface.loss.loss import CrossEntropyLoss
        for mode, what, dname in ((torch.float32, "C2 in the fp32 parity mode (same step as the headline; every operand split into three bf16 "
                                   "planes, six plane-pair MFMAs per product: fp32-level accuracy -- the mode the 1e-3 parity tests run in)",
                                   "fp32 (split-bf16 MFMA, 3 planes)"),
                                  ("fp32x2", "C2, fp32x2 (same step; fp32 tensors, every operand split into TWO bf16 planes, three plane-pair "
                                   "MFMAs per product: ~16 significand bits -- the cheapest mode inside the north-star 1e-3 embedding tolerance)",
                                   "fp32 storage, 2-plane split-bf16 MFMA")):
            xrface.set_compute_dtype(mode)
            net = model_irse.IR_SE_50([112, 112]).to(dev).train()
            flat = parallel.FlatParams(net.parameters_in_execution_order())
            opt = parallel.FusedSGD(flat, lr=0.05, momentum=0.9, weight_decay=5e-4)
            xb, yb = synth_batch(256, dev, 14)
            ce = CrossEntropyLoss()

            def c2p():
                opt.zero_grad()
                ce(net(xb), yb).backward()
                opt.step()
            ms, _ = _timed(c2p, 1, 3)
            ent = {"workload": what, "per_gpu_batch": 256, "dtype": dname, "ms_per_step": round(ms, 2),
                   "images_per_s": round(256 / ms * 1e3, 1)}
            if emb_errs is not None and mode in emb_errs:
                ent["embedding_rel_l2_vs_cpu"] = round(emb_errs[mode], 7)
            out.append(ent)
            del net, flat, opt
            torch.cuda.empty_cache()
    if world == 1:
        work.append(("C2 fp32 modes", w_c2_fp32_modes))

    def w_c1():
        # ---- C1: coarse net, batch 4, fp32 parity mode (the reference's CPU-runnable case); eager launches and one HIP-graph replay
        from xrface.graph import GraphedStep
        xrface.set_compute_dtype(torch.float32)
        net = FSRnet.Course_SR_Network().to(dev)
        flat = parallel.FlatParams(net.parameters())
        opt = parallel.FusedRMSprop(flat, lr=1e-4, alpha=0.99, weight_decay=1e-5)
        hr4, _ = synth_batch(4, dev, 13)
        lr4 = synth_lr(hr4)
        crit = MSELossFunc()
        lbuf = torch.zeros((), device=dev)

        def c1(lr_, hr_):
            opt.zero_grad()
            _, img = net(lr_)
            loss = 12.0 * crit(img, hr_)
            loss.backward()
            opt.step()
            lbuf.copy_(loss.detach())
            return lbuf
        ms, _ = _timed(lambda: c1(lr4, hr4), 3, 10)
        gs = GraphedStep(c1, [lr4, hr4], warmup=2)
        ms_g, _ = _timed(lambda: gs(lr4, hr4), 3, 20)
        gs.close()
        out.append({"workload": "C1 (BASELINE configs[0]): Course_SR_Network fwd+bwd of 12*mse97 + RMSprop; eager launches vs one HIP-graph "
                                "replay per step", "per_gpu_batch": 4, "dtype": "fp32 (split-bf16 MFMA)", "ms_per_step_eager": round(ms, 3),
                    "ms_per_step": round(ms_g, 3), "images_per_s": round(4 / ms_g * 1e3, 1)})
        del net, opt, flat, gs
        xrface.set_compute_dtype(torch.bfloat16)
        if cpu_legs:
            out[-1]["cpu_baseline"] = cpu_secondary("C1")
    if world == 1:
        work.append(("C1", w_c1))

    def w_c5():
        # ---- C5: P = 1e6 pair distances + 4000-threshold / 10-fold ROC
        P = 1_000_000
        g = torch.Generator(device=dev)
        g.manual_seed(0)
        e1 = torch.randn(P, 512, device=dev, generator=g)
        same = torch.rand(P, device=dev, generator=g) < 0.5
        e2 = torch.where(same[:, None], e1 + 0.5 * torch.randn(P, 512, device=dev, generator=g), torch.randn(P, 512, device=dev, generator=g))
        ms, _ = _timed(lambda: pair_dist(e1, e2), 2, 8)
        gbs = P * (2 * 512 * 4 + 4) / ms / 1e6
        fold = np.random.RandomState(0).randint(0, 10, P).astype(np.int32)
        same_h = same.cpu().numpy()
        thr = np.arange(0, 12000, 3)
        calculate_roc(thr, e1, e2, same_h, nrof_folds=10, fold_id=fold)
        t0 = time.perf_counter()
        _, _, acc, _ = calculate_roc(thr, e1, e2, same_h, nrof_folds=10, fold_id=fold)
        torch.cuda.synchronize()
        roc_ms = (time.perf_counter() - t0) * 1e3
        out.append({"workload": "C5 (BASELINE configs[4]): 1M-pair 512-d squared-L2 distances + 4000-threshold / 10-fold ROC",
                    "pairs": P, "pairdist_ms": round(ms, 3), "pairdist_gb_s": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK_HBM_GBS, 4),
                    "calculate_roc_ms": round(roc_ms, 2), "accuracy": round(float(acc), 4),
                    "pairs_per_s": round(P / (roc_ms * 1e-3), 1)})
        del e1, e2
        if cpu_legs:
            out[-1]["cpu_baseline"] = cpu_secondary("C5")
    if world == 1:
        work.append(("C5", w_c5))

    for name, fn in work:
        try:
            fn()
        except Exception as e:   # noqa: BLE001 -- report, restore the mode, go on
            out.append({"workload": name, "error": f"{type(e).__name__}: {e}"[:300]})
        finally:
            xrface.set_compute_dtype(torch.bfloat16)
            torch.cuda.empty_cache()
    return out

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true")
    ap.add_argument("--c4-batch", type=int, default=256, help="per-GPU batch of the secondary C4 step")
    ap.add_argument("--c3-batch", type=int, default=128, help="per-GPU batch of the secondary C3 step")
    ap.add_argument("--secondary-timeout", type=float, default=420.0, help="N > 1: seconds the data-parallel secondary steps may take")
    ap.add_argument("--all-sites", action="store_true", help="report every conv launch site in `kernels`, not the top 8")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # XR_REHEARSE=1: multi-rank rehearsal on a ONE-GPU box (every rank on cuda:0, gloo transport) -- exercises the
    # data-parallel code path, never used for reported numbers
    rehearse = os.environ.get("XR_REHEARSE", "0") == "1"
    if rehearse:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        if rehearse:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    import xrface
    from xrface import ops, parallel
    from xrface.loss.loss import CrossEntropyLoss
    from xrface.model.model_irse import IR_SE_50

    dtype = torch.bfloat16 if args.dtype == "bf16" else torch.float32
    xrface.set_compute_dtype(dtype)
    torch.manual_seed(0)
    model = IR_SE_50([112, 112]).to(dev)
    model.train()
    parallel.broadcast_module(model)
    flat = parallel.FlatParams(model.parameters_in_execution_order())
    bn_params = [p for n_, p in model.named_parameters() if p.dim() == 1]
    opt = parallel.FusedSGD(flat, lr=0.05, momentum=0.9, weight_decay=5e-4, no_decay=bn_params)
    reducer = parallel.BucketedAllReduce(flat)
    crit = CrossEntropyLoss()
    batches = [synth_batch(args.batch, dev, seed=rank * 97 + i) for i in range(2)]

    def step(i):
        x, y = batches[i % 2]
        opt.zero_grad()
        out = model(x)
        loss = crit(out, y)
        loss.backward()
        reducer.finish()
        opt.step()
        return loss

    # the step runs on a high-priority stream: the side stream that carries the weight gradients (normal priority) then
    # yields to the forward / dgrad / norm-backward chain instead of competing with it (-0.1 ms/step in-process)
    hp = torch.cuda.Stream(dev, priority=-1)
    hp.wait_stream(torch.cuda.current_stream(dev))
    hp_ctx = torch.cuda.stream(hp)
    hp_ctx.__enter__()
    for i in range(args.warmup):
        step(i)
    # one extra UNTIMED step with an event pair around every convolution launch (forward, input gradient, weight gradient;
    # each on the stream it is launched on): the per-site table and the choice of the dominant kernel come from it
    bf = dtype == torch.bfloat16
    table = []
    if bf:
        # EVERY rank runs this step (it contains the gradient all-reduce); only rank 0 brackets its launches with events
        if rank == 0:
            ops._cfg["probe"] = {"all": {}}
        step(args.warmup)
        torch.cuda.synchronize()
        if rank == 0:
            table = kernel_table(ops._cfg.pop("probe")["all"], args.batch, 1)
    if world > 1:
        dist.barrier()
        # replicas must still agree after the warm-up steps (same averaged gradients -> same weights)
        chk = flat.flat[:4096].clone()
        dist.broadcast(chk, 0)
        assert torch.equal(chk, flat.flat[:4096]), "replicas diverged: gradient all-reduce is broken"
    torch.cuda.synchronize()
    dom = table[0] if table else None
    probe = {"tag": dom["_tag"] if dom else None, "events": []}
    if dom is not None:
        ops._cfg["probe"] = probe   # event pairs around the dominant site's launches only, inside the timed region
    t0 = time.perf_counter()
    for i in range(args.steps):
        loss = step(i)
    torch.cuda.synchronize()
    ops._cfg.pop("probe", None)
    if world > 1:
        dist.barrier()
    el = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        el = float(t.item())
    final_loss = float(loss.item())
    assert final_loss == final_loss, "loss is NaN"
    hp_ctx.__exit__(None, None, None)
    torch.cuda.current_stream(dev).wait_stream(hp)

    emb_errs = None     # embedding error of every arithmetic mode vs the CPU oracle (rank 0, N = 1 leg)
    if rank == 0:
        ms = el / args.steps * 1e3
        gb = args.batch * world
        value = gb / (ms * 1e-3)
        step_tflops = 3.0 * IRSE50_FWD_GFLOP * gb / (ms * 1e-3) / 1e3
        line = {
            "metric": "face images/sec (train step, 112x112)", "value": round(value, 1), "unit": "images/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "IR-SE-50 teacher fwd+bwd+SGD, CE on the 512-d output (train_teacher_model.py path), "
                                   "112x112, BN train mode, Dropout on", "global_batch": gb, "per_gpu_batch": args.batch,
                       "parallelism": f"dp{world}", "final_loss": round(final_loss, 4)},
            "step_mfma": {"algorithmic_tflop_per_step": round(3.0 * IRSE50_FWD_GFLOP * gb / 1e3, 3),
                          "achieved_tflops": round(step_tflops, 1), "frac_of_bf16_peak_per_gpu":
                              round(step_tflops / world / PEAK_BF16_TFLOPS, 4)},
        }
        sh = step_roofline("c2", ms, args.batch) if args.dtype == "bf16" else None   # HBM bytes of the whole step (PMC, committed profile)
        if sh is not None:
            line["step_hbm"] = sh
        if dom is not None:
            evs = probe["events"]
            in_ms = sum(a.elapsed_time(b) for a, b in evs) / len(evs) if evs else dom["avg_ms"]
            fl = _tag_flops(dom["_tag"], args.batch)
            iso = isolated_ms(dom["_tag"], args.batch, dev)
            line["roofline"] = {
                "bound": "mfma", "achieved": round(fl / (in_ms * 1e-3) / 1e12, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                "frac": round(fl / (in_ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4), "traffic": pmc_traffic(dom["kernel"], dom["site"]),
                "kernel": dom["kernel"], "site": dom["site"], "selection": "largest total in-step time among all conv launch sites",
                "avg_launch_ms": round(in_ms, 4), "launches_timed_in_step": len(evs), "launches_per_step": dom["launches_per_step"],
                "isolated_launch_ms": round(iso, 4), "isolated_tflops": round(fl / (iso * 1e-3) / 1e12, 1),
                "algorithmic_gflop_per_launch": round(fl / 1e9, 2)}
            line["kernels"] = [{k: v for k, v in r.items() if k != "_tag"} for r in (table if args.all_sites else table[:8])]
            line["conv_ms_per_step_probed"] = round(sum(r["total_ms_per_step"] for r in table), 2)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"], emb_err = cpu_baseline(dev, dtype)
            line["embedding_rel_l2_vs_cpu"] = round(emb_err[0], 6)
            line["embedding_rel_l2_vs_cpu_fp32_parity_mode"] = round(emb_err[1], 7)
            line["embedding_rel_l2_vs_cpu_fp32x2_mode"] = round(emb_err[2]["fp32x2"], 7)
            emb_errs = emb_err[2]
    else:
        line = None
    if not args.no_secondary and bf:
        # N > 1: EVERY rank runs the data-parallel C4 / C3 steps (they hold the gradient all-reduces); rank 0 reports them
        del model, flat, opt, reducer
        torch.cuda.empty_cache()
        if world > 1:
            # the secondary steps hold collectives: if one rank fails or stalls the others would wait for ever and the headline
            # measured above would be lost with them -- a watchdog prints it (without `secondary`) and ends the process
            import threading

            def _watchdog():
                if line is not None:
                    line["secondary"] = [{"error": f"secondary workloads did not finish within {args.secondary_timeout} s on {world} ranks"}]
                    print(json.dumps(line), flush=True)
                os._exit(0)
            wd = threading.Timer(args.secondary_timeout, _watchdog)
            wd.daemon = True
            wd.start()
        try:
            with torch.cuda.stream(hp):      # same stream set-up as the headline: the side work yields to the critical path
                sec = secondary_workloads(dev, c4_batch=args.c4_batch, c3_batch=args.c3_batch, world=world, rank=rank,
                                          cpu_legs=not args.no_cpu_baseline, emb_errs=emb_errs)
            torch.cuda.current_stream(dev).wait_stream(hp)
        except Exception as e:   # noqa: BLE001 -- the headline line is printed whatever happens to the extras
            sec = [{"error": f"{type(e).__name__}: {e}"[:300]}]
        if line is not None:
            line["secondary"] = sec
    if world > 1 and not args.no_secondary and bf:
        wd.cancel()
    if line is not None:
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
