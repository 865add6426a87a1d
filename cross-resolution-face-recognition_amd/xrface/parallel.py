"""Data-parallel layer (new functionality: the reference is single-process).

One process per GPU; gradients are averaged with bucketed all-reduce over RCCL (torch.distributed backend
"nccl" on ROCm) launched from post-accumulate-grad hooks so the collectives overlap the rest of backward.
Parameters and gradients live in two flat fp32 buffers (views re-pointed into them), which makes a bucket
a plain slice -- no packing copies -- and lets the fused optimizers update everything in one launch.
BatchNorm statistics stay per-GPU (the reference has no SyncBN); InstanceNorm is per image anyway.
"""
from __future__ import annotations

import torch
import torch.distributed as dist

from . import ops
from ._lib import lib, ptr, stream


class FlatParams:
    """Re-point ``p.data`` / ``p.grad`` of every trainable parameter into two flat fp32 buffers."""

    def __init__(self, params, align=64, direct=True):
        self.params = [p for p in params if p.requires_grad]
        assert self.params, "no trainable parameters"
        dev = self.params[0].device
        offs, n = [], 0
        for p in self.params:
            offs.append(n)
            n += (p.numel() + align - 1) // align * align
        self.numel = n
        self.offsets = offs
        self.flat = torch.zeros(n, dtype=torch.float32, device=dev)
        self.grad = torch.zeros(n, dtype=torch.float32, device=dev)
        for p, o in zip(self.params, offs):
            v = self.flat[o:o + p.numel()].view(p.shape)
            v.copy_(p.data)
            p.data = v
            p.grad = self.grad[o:o + p.numel()].view(p.shape)
            p._xr_direct = direct  # HIP backward kernels accumulate straight into p.grad (see ops._direct)

    def zero_grad(self):
        """One memset; .grad tensors stay allocated views (autograd accumulates in place)."""
        ops.join_side_stream()
        ops._side["cb"] = False   # a backward pass that raised never ran its end-of-backward callback: re-arm it
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


class BucketedAllReduce:
    """Average ``flat.grad`` across ranks in ~bucket_mb slices, each launched as soon as the last of its
    parameters has accumulated its gradient (reverse registration order ~ reverse execution order).

    ``overlap=False`` defers every launch to ``finish()``: required when FlatParams(direct=True) is combined with a
    model that applies one parameter at several sites (the shared FSRNet trunks), because the direct-accumulation
    path signals per site, not per parameter."""

    def __init__(self, flat: FlatParams, bucket_mb: float = 24.0, group=None, overlap: bool = True):
        self.flat = flat
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.backend = dist.get_backend(group) if dist.is_initialized() else None
        cap = int(bucket_mb * (1 << 20) / 4)
        # buckets over the parameter list walked backwards (last layers finish backward first)
        self.buckets = []  # (lo, hi) element ranges, param index sets
        hi = flat.numel
        cur = []
        lo = hi
        for i in range(len(flat.params) - 1, -1, -1):
            lo = flat.offsets[i]
            cur.append(i)
            if hi - lo >= cap or i == 0:
                self.buckets.append((lo, hi, tuple(cur)))
                hi, cur = lo, []
        self.bucket_of = {}
        for b, (_, _, idx) in enumerate(self.buckets):
            for i in idx:
                self.bucket_of[i] = b
        self._pending = [0] * len(self.buckets)
        self._works = []
        self._hooks = []
        self.enabled = self.world > 1
        if self.enabled and overlap:
            for i, p in enumerate(flat.params):
                h = self._make_hook(i)
                self._hooks.append(p.register_post_accumulate_grad_hook(h))
                p._xr_grad_hook = h  # direct-accumulation path bypasses AccumulateGrad: it calls the hook itself
        self.reset()

    def reset(self):
        self._pending = [len(idx) for (_, _, idx) in self.buckets]
        self._works = []

    def _launch(self, b):
        ops.join_side_stream()   # weight gradients computed on the side stream must have landed in the bucket
        lo, hi, _ = self.buckets[b]
        buf = self.flat.grad[lo:hi]
        if self.backend == "nccl":
            self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        else:
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append((w, buf))

    def _make_hook(self, i):
        def hook(_p):
            b = self.bucket_of[i]
            self._pending[b] -= 1
            if self._pending[b] == 0:  # a parameter used at several sites may fire more than once: launch only once
                self._launch(b)
        return hook

    def finish(self):
        """Launch any bucket whose parameters never produced a gradient this step, then wait for all."""
        if not self.enabled:
            return
        for b, left in enumerate(self._pending):
            if left > 0:
                self._launch(b)
        for w in self._works:
            if isinstance(w, tuple):
                w[0].wait()
                w[1].div_(self.world)
            else:
                w.wait()
        self.reset()


def broadcast_module(module, src=0, group=None):
    """Identical replicas at start: parameters and buffers from rank ``src``."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src, group=group)


class _FlatOptimizer(torch.optim.Optimizer):
    """torch.optim.Optimizer subclass (schedulers / state_dict tooling keep working) that performs the whole
    update in one fused HIP launch over a FlatParams buffer."""

    def __init__(self, flat: FlatParams, defaults):
        self.flat = flat
        super().__init__(flat.params, defaults)
        self._steps = 0

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()


class FusedSGD(_FlatOptimizer):
    """torch.optim.SGD(momentum, weight_decay) semantics (DISTILLATION/train_HRN.py:75-84); ``no_decay`` lists
    parameters excluded from weight decay (the reference keeps BN parameters decay-free, :75-80)."""

    def __init__(self, flat, lr, momentum=0.9, weight_decay=0.0, no_decay=()):
        super().__init__(flat, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.mom = torch.zeros_like(flat.flat)
        self.mask = None
        nd = {id(p) for p in no_decay}
        if nd and weight_decay != 0.0:
            self.mask = torch.ones(flat.numel, dtype=torch.uint8, device=flat.flat.device)
            for p, o in zip(flat.params, flat.offsets):
                if id(p) in nd:
                    self.mask[o:o + p.numel()] = 0

    @torch.no_grad()
    def step(self, closure=None):
        ops.join_side_stream()   # no-op unless a backward pass ended without its end-of-backward join (e.g. it raised)
        g = self.param_groups[0]
        lib.xr_sgd_step(ptr(self.flat.flat), ptr(self.flat.grad), ptr(self.mom), self.flat.numel, g["lr"], g["momentum"],
                        g["weight_decay"], ptr(self.mask), int(self._steps == 0), stream())
        self._steps += 1
        ops.invalidate_weight_cache(self.flat.params)


class FusedRMSprop(_FlatOptimizer):
    """torch.optim.RMSprop(alpha, eps, weight_decay) (Face_Hallucination_sub_Net.py:120-124)."""

    def __init__(self, flat, lr, alpha=0.99, eps=1e-8, weight_decay=0.0):
        super().__init__(flat, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay))
        self.sq = torch.zeros_like(flat.flat)

    @torch.no_grad()
    def step(self, closure=None):
        ops.join_side_stream()   # no-op unless a backward pass ended without its end-of-backward join (e.g. it raised)
        g = self.param_groups[0]
        lib.xr_rmsprop_step(ptr(self.flat.flat), ptr(self.flat.grad), ptr(self.sq), self.flat.numel, g["lr"], g["alpha"],
                            g["eps"], g["weight_decay"], stream())
        self._steps += 1
        ops.invalidate_weight_cache(self.flat.params)


class FusedAdam(_FlatOptimizer):
    """torch.optim.Adam(betas, eps, weight_decay) (SUPER_RESOLUTION/train_FHN.py:115-121)."""

    def __init__(self, flat, lr, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(flat, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))
        self.m = torch.zeros_like(flat.flat)
        self.v = torch.zeros_like(flat.flat)

    @torch.no_grad()
    def step(self, closure=None):
        ops.join_side_stream()   # no-op unless a backward pass ended without its end-of-backward join (e.g. it raised)
        g = self.param_groups[0]
        self._steps += 1
        cap = ops._graph["capturing"]
        lib.xr_adam_step(ptr(self.flat.flat), ptr(self.flat.grad), ptr(self.m), ptr(self.v), self.flat.numel, g["lr"],
                         g["betas"][0], g["betas"][1], g["eps"], g["weight_decay"], self._steps,
                         ptr(ops._graph["tick"]) if cap else None, ops._graph["tick_ref"] if cap else 0, stream())
        ops.invalidate_weight_cache(self.flat.params)
