#!/usr/bin/env python
"""Benchmark of the hot path (contract: see the task statement / DESIGN.md section "Measurement").

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

One step = IR-SE-50 forward + backward + gradient all-reduce (N > 1) + fused SGD update on a batch of
synthetic 112x112 faces (BASELINE.json configs[1]: batch 256/GPU, bf16, train_teacher_model.py path).
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
PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X (MI355X_MICROARCH.md)
PEAK_HBM_GBS = 8000.0


def synth_batch(n, dev, seed):
    """Counter-free cheap synthetic faces on the device: smooth low-frequency field + noise in [-1,1]."""
    g = torch.Generator(device=dev)
    g.manual_seed(1000 + seed)
    lo = torch.randn(n, 3, 14, 14, device=dev, generator=g)
    x = torch.nn.functional.interpolate(lo, size=(112, 112), mode="bilinear", align_corners=False)
    x = (x + 0.1 * torch.randn(n, 3, 112, 112, device=dev, generator=g)).clamp_(-1, 1)
    y = torch.randint(0, 512, (n,), device=dev, generator=g)
    return x.contiguous(), y


DOM_TAG = ("fwd", 256, 256, 14, 14, 3, 1)   # conv3x3 256->256 @14x14 stride 1 forward: 26 launches per IR-SE-50 step


def dominant_kernel_roofline(dev, dtype, batch, probe_events, reps=10):
    """Roofline of the dominant kernel (8-wave implicit-GEMM conv on MFMA, xr_conv8.hip igemm8_kernel<false,2,4,3>) at
    its most frequent shape, conv3x3 256->256 @14x14 forward.  `achieved` uses the kernel's average duration INSIDE the
    timed training steps (HIP event pairs recorded on the launch stream around each of its launches; ops._probe_begin);
    the same kernel timed back-to-back in isolation is reported next to it."""
    from xrface import ops
    from xrface._lib import dt, lib, ptr, stream
    N, H, C, K = batch, 14, 256, 256
    x = torch.randn(N, H, H, C, device=dev).to(dtype)
    w = torch.randn(K, C, 3, 3, device=dev) * 0.02
    pk, kg = ops._packed(w, "fwd", dtype, K, 1, 9, C, C, C * 9, 0, 1, 9)
    y = torch.empty(N, H, H, K, device=dev, dtype=dtype)

    def launch():
        lib.xr_conv_igemm(dt(x), ptr(x), ptr(pk), None, ptr(y), N, H, H, C, H, H, K, 3, 3, 1, 1, 0, kg, K, None, 0, None, None, None, 1, None, None, None, stream())
    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        launch()
    e1.record()
    torch.cuda.synchronize()
    iso_ms = e0.elapsed_time(e1) / reps
    if probe_events:
        ms = sum(a.elapsed_time(b) for a, b in probe_events) / len(probe_events)
    else:
        ms = iso_ms
    flops = 2.0 * N * H * H * K * C * 9
    achieved = flops / (ms * 1e-3) / 1e12
    # HBM bytes per launch of this kernel at this shape come from the committed rocprofv3 PMC passes (FETCH_SIZE x2
    # gfx950 correction + WRITE_SIZE; profiles/r01_dominant_kernel_pmc.json) -- bench.py cannot run the profiler itself
    traffic = None
    try:
        with open(os.path.join(ROOT, "profiles", "r01_dominant_kernel_pmc.json")) as f:
            pmc = json.load(f)
        if batch == 256:
            traffic = pmc["hbm_bytes_per_launch"]
    except Exception:
        traffic = None
    return {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
            "kernel": "igemm8_kernel<fwd, 224x256 tile, 8 waves> conv3x3 256->256 @14x14", "avg_launch_ms": round(ms, 4),
            "launches_timed_in_step": len(probe_events), "isolated_launch_ms": round(iso_ms, 4),
            "isolated_tflops": round(flops / (iso_ms * 1e-3) / 1e12, 2), "algorithmic_gflop_per_launch": round(flops / 1e9, 2)}


def cpu_baseline(budget_s=12.0):
    """The CPU oracle (oracle/cpu_ref.py: stock torch fp32 CPU ops, the reference's module graph) timed on the
    host cores on a bounded sample of the same workload: IR-SE-50 fwd+bwd at N = 8 per step."""
    from oracle import cpu_ref as R
    from oracle import detgen as G
    from xrface.model.model_irse import IR_SE_50
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    cores = max(1, min(cores, 16))  # a 1-GPU box exposes 256 logical CPUs but grants a 16-CPU share
    torch.set_num_threads(cores)
    tmpl = {k: v for k, v in IR_SE_50([112, 112]).state_dict().items()}
    sd = G.det_state_dict(tmpl)
    n = 8
    x = G.synth_faces(n, 112, seed=1)
    t = G.synth_labels(n, 512)
    R.teacher_step_grads(sd, x, t, se=True)  # warm-up
    steps, t0 = 0, time.perf_counter()
    while True:
        R.teacher_step_grads(sd, x, t, se=True)
        steps += 1
        if time.perf_counter() - t0 > budget_s or steps >= 200:
            break
    el = time.perf_counter() - t0
    return {"value": round(n * steps / el, 2), "unit": "images/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"IR-SE-50 fwd+bwd (CE on the 512-d output), fp32, N={n}/step, {steps} steps in {el:.1f}s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="per-GPU batch")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
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
    if world > 1:
        dist.barrier()
        # replicas must still agree after the warm-up steps (same averaged gradients -> same weights)
        chk = flat.flat[:4096].clone()
        dist.broadcast(chk, 0)
        assert torch.equal(chk, flat.flat[:4096]), "replicas diverged: gradient all-reduce is broken"
    torch.cuda.synchronize()
    probe = {"tag": DOM_TAG, "events": []}
    if rank == 0 and args.batch == 256 and dtype == torch.bfloat16:
        ops._cfg["probe"] = probe   # event pairs around the dominant kernel's launches (26 per step), timed region only
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
            "roofline": dominant_kernel_roofline(dev, dtype, args.batch, probe["events"]),
        }
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline()
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
