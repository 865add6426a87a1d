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
            p.__dict__["_xr_touched"] = False
            # gradients that arrive through autograd's AccumulateGrad (stock torch ops, direct=False) count as touched too
            p.register_post_accumulate_grad_hook(_mark_touched)
        self._masks = {}

    def touched_mask(self, no_decay_ids=()):
        """uint8 per element for the fused optimizers (xr_*_step ``wd_mask``): bit 0 = weight decay applies, bit 1 = the
        parameter received no gradient since the last zero_grad() -> left untouched (a stock optimizer skips .grad None).
        Returns None when every parameter was touched and none is decay-free.  One device mask per (touched set, decay-free
        set), kept for the lifetime of this object: a captured HIP graph holds the raw pointer of the mask it was captured
        with, so a mask is never freed or rewritten once handed out (a training loop sees two or three distinct sets)."""
        key = (tuple(p.__dict__.get("_xr_touched", False) for p in self.params), frozenset(no_decay_ids))
        if key not in self._masks:
            if all(key[0]) and not no_decay_ids:
                self._masks[key] = None
            else:
                m = torch.ones(self.numel, dtype=torch.uint8)
                for p, o, t in zip(self.params, self.offsets, key[0]):
                    if not t:
                        m[o:o + p.numel()] = 2
                    elif id(p) in no_decay_ids:
                        m[o:o + p.numel()] = 0
                if self.flat.is_cuda:   # pinned staging: the upload is an asynchronous copy (legal inside a graph capture)
                    m = m.pin_memory()
                dm = torch.empty(self.numel, dtype=torch.uint8, device=self.flat.device)
                dm.copy_(m, non_blocking=True)
                self._masks[key] = dm
                self._mask_hosts = getattr(self, "_mask_hosts", []) + [m]   # keep the pinned source alive until the copy ran
        return self._masks[key]

    def zero_grad(self):
        """One memset; .grad tensors stay allocated views (autograd accumulates in place)."""
        ops.join_side_stream()
        ops._side["cb"] = False   # a backward pass that raised never ran its end-of-backward callback: re-arm it
        self.grad.zero_()
        for p, o in zip(self.params, self.offsets):
            p.__dict__["_xr_touched"] = False
            if p.grad is None or p.grad.data_ptr() != self.grad.data_ptr() + 4 * o:
                p.grad = self.grad[o:o + p.numel()].view(p.shape)


def _mark_touched(p):
    p.__dict__["_xr_touched"] = True


class BucketedAllReduce:
    """Average ``flat.grad`` across ranks in ~bucket_mb slices, each launched as soon as the last of its
    parameters has accumulated its gradient (reverse registration order ~ reverse execution order).

    The direct-accumulation path (ops._direct_done) signals once per USE SITE of a parameter, not once per parameter (the
    shared FSRNet trunks apply one parameter at nine sites).  The first step therefore only COUNTS the signals of every
    parameter (all buckets are launched from ``finish()``); from the second step on a parameter is complete when its
    count reaches the learned one, and a bucket is launched when all of its parameters are complete.  A parameter that
    signals more often than learned after its bucket has gone out makes ``finish()`` raise (and re-learn) instead of
    silently averaging incomplete gradients.  ``overlap=False`` defers every launch to ``finish()``."""

    def __init__(self, flat: FlatParams, bucket_mb: float = 24.0, group=None, overlap: bool = True, force: bool = False,
                 comm_stream: bool = True, tail_mb: float = 1.0):
        self.flat = flat
        self.comm_stream, self._comm = comm_stream, None
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
        # The last bucket (the first layers) completes with the very last gradient of the backward pass: its collective is the one
        # nothing hides.  Cut it so that only ~tail_mb wait for the end; the rest of it leaves a few layers earlier.
        tcap = int(tail_mb * (1 << 20) / 4)
        if tcap > 0 and self.buckets:
            lo, hi, idx = self.buckets[-1]
            asc = sorted(idx)
            k = 0
            while k < len(asc) - 1 and flat.offsets[asc[k + 1]] - lo <= tcap:
                k += 1
            if k + 1 < len(asc):
                cut = flat.offsets[asc[k + 1]]
                self.buckets[-1] = (cut, hi, tuple(i for i in idx if i > asc[k]))
                self.buckets.append((lo, cut, tuple(i for i in reversed(asc[:k + 1]))))
        self.bucket_of = {}
        for b, (_, _, idx) in enumerate(self.buckets):
            for i in idx:
                self.bucket_of[i] = b
        self._pending = [0] * len(self.buckets)
        self._works = []
        self._hooks = []
        self._expected = None                      # per-parameter signal counts of one step, learned in the first step
        self._count = [0] * len(flat.params)
        self._late = None
        self.launch_order = []                     # bucket indices in the order they were launched this step (tests)
        self.last_launch_order, self.last_early = [], 0
        # force: run the collectives even in a one-rank group (tests: the RCCL path on a single GPU)
        self.enabled = self.world > 1 or (force and dist.is_initialized())
        if self.enabled and overlap:
            for i, p in enumerate(flat.params):
                h = self._make_hook(i)
                self._hooks.append(p.register_post_accumulate_grad_hook(h))
                p._xr_grad_hook = h  # direct-accumulation path bypasses AccumulateGrad: it calls the hook itself
        self.reset()

    def reset(self):
        exp = self._expected
        self._pending = [sum(1 for i in idx if exp is None or exp[i] > 0) for (_, _, idx) in self.buckets]
        self._count = [0] * len(self.flat.params)
        self._launched = [False] * len(self.buckets)
        self._works = []

    def _launch(self, b):
        self._launched[b] = True
        self.launch_order.append(b)
        lo, hi, _ = self.buckets[b]
        buf = self.flat.grad[lo:hi]
        if buf.is_cuda and self.comm_stream:
            # The bucket holds gradients written by the stream backward runs on AND by the weight-gradient side stream, which runs
            # behind it by design.  Joining the side stream into the backward stream here (once per bucket: 8 times per IR-SE-50
            # step) would stall backward until the weight gradients caught up; instead a launcher stream waits for both and the
            # collective is enqueued from there -- backward never waits, finish() makes the optimizer's stream wait for the works.
            cur = torch.cuda.current_stream(buf.device)
            if self._comm is None:
                self._comm = torch.cuda.Stream(buf.device)
            self._comm.wait_stream(cur)
            with torch.cuda.stream(self._comm):
                ops.join_side_stream()
                self._enqueue(buf)
            return
        ops.join_side_stream()   # weight gradients computed on the side stream must have landed in the bucket
        self._enqueue(buf)

    def _enqueue(self, buf):
        if self.backend == "nccl":
            self._works.append(dist.all_reduce(buf, op=dist.ReduceOp.AVG, group=self.group, async_op=True))
        else:
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
            self._works.append((w, buf))

    def _make_hook(self, i):
        def hook(_p):
            self._count[i] += 1
            exp = self._expected
            if exp is None:            # first step: learn the per-parameter signal counts, launch from finish()
                return
            b = self.bucket_of[i]
            if self._count[i] == exp[i]:
                self._pending[b] -= 1
                if self._pending[b] == 0:
                    self._launch(b)
            elif self._count[i] > exp[i] and self._launched[b]:
                self._late = i         # more use sites than learned, and the bucket already went out
        return hook

    def finish(self):
        """Launch any bucket whose parameters never produced a gradient this step, then wait for all."""
        if not self.enabled:
            return
        self.last_early = len(self.launch_order)   # buckets that went out from the gradient hooks, during backward
        for b in range(len(self.buckets)):
            if not self._launched[b]:
                self._launch(b)
        for w in self._works:
            if isinstance(w, tuple):
                w[0].wait()
                w[1].div_(self.world)
            else:
                w.wait()
        late = self._late
        if self._expected is None and self._hooks:
            self._expected = list(self._count)
        self._late = None
        self.last_launch_order, self.launch_order = self.launch_order, []
        if late is not None:
            self._expected = None     # re-learn on the next step
            self.reset()
            raise RuntimeError(f"BucketedAllReduce: parameter #{late} accumulated gradient at more sites than in the step the "
                               "bucket schedule was learned from, after its bucket had been all-reduced; this step's "
                               "averaged gradients are incomplete -- repeat the step (the schedule is being re-learned)")
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

    _STATE = ()   # names of the flat state buffers of the subclass

    def __init__(self, flat: FlatParams, defaults):
        self.flat = flat
        super().__init__(flat.params, defaults)
        assert len(self.param_groups) == 1, "fused flat optimizers take ONE parameter group (hyper-parameters are per buffer)"
        self._steps = 0
        self._nd = frozenset()

    def zero_grad(self, set_to_none: bool = False):
        self.flat.zero_grad()

    def _wd_mask(self):
        return self.flat.touched_mask(self._nd)

    def state_dict(self):
        """torch.optim.Optimizer.state_dict() plus the flat state buffers and the step count (momentum / second moments /
        bias-correction step live outside ``self.state`` -- one buffer each, not one entry per parameter)."""
        sd = super().state_dict()
        sd["xr_flat"] = {"steps": self._steps, "numel": self.flat.numel, "shapes": [tuple(p.shape) for p in self.flat.params],
                         **{k: getattr(self, k).detach().clone() for k in self._STATE}}
        return sd

    def load_state_dict(self, state_dict):
        """Everything is validated BEFORE anything is mutated (torch's own load rewrites param_groups first)."""
        state_dict = dict(state_dict)
        fl = state_dict.pop("xr_flat", None)
        if fl is None:
            raise RuntimeError("not a fused flat optimizer checkpoint (no 'xr_flat' entry): momentum would silently reset")
        if int(fl["numel"]) != self.flat.numel or ("shapes" in fl and [tuple(s) for s in fl["shapes"]] != [tuple(p.shape) for p in self.flat.params]):
            raise RuntimeError("fused flat optimizer checkpoint was written for a different parameter layout "
                               f"({fl['numel']} elements vs {self.flat.numel})")
        for k in self._STATE:
            if k not in fl or fl[k].numel() != self.flat.numel:
                raise RuntimeError(f"fused flat optimizer checkpoint: state buffer '{k}' missing or of the wrong size")
        super().load_state_dict(state_dict)
        self._steps = int(fl["steps"])
        for k in self._STATE:
            getattr(self, k).copy_(fl[k])


class FusedSGD(_FlatOptimizer):
    """torch.optim.SGD(momentum, weight_decay) semantics (DISTILLATION/train_HRN.py:75-84); ``no_decay`` lists
    parameters excluded from weight decay (the reference keeps BN parameters decay-free, :75-80)."""
    _STATE = ("mom",)

    def __init__(self, flat, lr, momentum=0.9, weight_decay=0.0, no_decay=()):
        super().__init__(flat, dict(lr=lr, momentum=momentum, weight_decay=weight_decay))
        self.mom = torch.zeros_like(flat.flat)
        self._nd = frozenset(id(p) for p in no_decay) if weight_decay != 0.0 else frozenset()

    @torch.no_grad()
    def step(self, closure=None):
        ops.join_side_stream()   # no-op unless a backward pass ended without its end-of-backward join (e.g. it raised)
        g = self.param_groups[0]
        lib.xr_sgd_step(ptr(self.flat.flat), ptr(self.flat.grad), ptr(self.mom), self.flat.numel, g["lr"], g["momentum"],
                        g["weight_decay"], ptr(self._wd_mask()), int(self._steps == 0), stream())
        self._steps += 1
        ops.invalidate_weight_cache(self.flat.params)


class FusedRMSprop(_FlatOptimizer):
    """torch.optim.RMSprop(alpha, eps, weight_decay) (Face_Hallucination_sub_Net.py:120-124)."""
    _STATE = ("sq",)

    def __init__(self, flat, lr, alpha=0.99, eps=1e-8, weight_decay=0.0):
        super().__init__(flat, dict(lr=lr, alpha=alpha, eps=eps, weight_decay=weight_decay))
        self.sq = torch.zeros_like(flat.flat)

    @torch.no_grad()
    def step(self, closure=None):
        ops.join_side_stream()   # no-op unless a backward pass ended without its end-of-backward join (e.g. it raised)
        g = self.param_groups[0]
        lib.xr_rmsprop_step(ptr(self.flat.flat), ptr(self.flat.grad), ptr(self.sq), self.flat.numel, g["lr"], g["alpha"],
                            g["eps"], g["weight_decay"], ptr(self._wd_mask()), stream())
        self._steps += 1
        ops.invalidate_weight_cache(self.flat.params)


class FusedAdam(_FlatOptimizer):
    """torch.optim.Adam(betas, eps, weight_decay) (SUPER_RESOLUTION/train_FHN.py:115-121).
    Deviation (documented, parity unpinned by any fixture): the bias corrections use ONE step count for the whole buffer;
    torch.optim.Adam keeps a per-parameter count that only advances when .grad is not None, so a parameter FIRST touched at
    step k > 1 gets slightly different corrections here (parameters touched from step 1 on -- every parameter of the
    reference's steps -- are identical)."""
    _STATE = ("m", "v")

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
                         ptr(ops._graph["tick"]) if cap else None, ops._graph["tick_ref"] if cap else 0,
                         ptr(self._wd_mask()), stream())
        ops.invalidate_weight_cache(self.flat.params)
