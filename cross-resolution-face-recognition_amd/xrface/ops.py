"""torch.autograd.Function wrappers over the C ABI (include/xrface.h).

Internal activations are explicit NHWC buffers ``[N, H, W, Cp]`` (Cp = channel pitch, a multiple of 8,
zero padded) in bf16 or fp32; the autograd graph is built on those buffers.  ``enter``/``leave``
convert at module boundaries (zero-copy for feature maps whose channel count is a multiple of 8).
Every op fails loudly when the HIP library is unavailable -- there is no CPU path here.
"""
from __future__ import annotations

import ctypes
import weakref

import torch
from torch.autograd import Function

from ._lib import ACT_NONE, ACT_PRELU, ACT_RELU, ACT_TANH, XR_BF16, XR_F32, XR_F32X2, dt, lib, ptr, stream, stream_of

EPS = 1e-5
import os as _os
_cfg = {"compute_dtype": torch.float32, "wgrad_blocks": int(_os.environ.get("XR_WGRAD_BLOCKS", "512")), "dalpha_spread": 32, "pack_plan": 1, "wgrad_stream": int(_os.environ.get("XR_WGRAD_STREAM", "1")), "fuse_prelu": 1, "fold_finalize": 1, "fuse_bn_reduce": 1, "fuse_conv_stats": 1, "direct64": int(_os.environ.get("XR_DIRECT64", "1")),
        "lockstep": int(_os.environ.get("XR_LOCKSTEP", "1")), "wgrad_rows": int(_os.environ.get("XR_WGRAD_ROWS", "2")),
        "fuse_in_reduce": int(_os.environ.get("XR_FUSE_IN_REDUCE", "1")),
        "res_trunk": int(_os.environ.get("XR_RES_TRUNK", "1")), "chain_units": int(_os.environ.get("XR_CHAIN_UNITS", "1")), "direct64_prelu": 1,
        "wgrad_rows112": int(_os.environ.get("XR_WGRAD_ROWS112", "0")), "f32_planes": 3,
        "wgrad64_stream": int(_os.environ.get("XR_WGRAD64_STREAM", "1")), "deterministic": 0,
        "block_abi": int(_os.environ.get("XR_BLOCK_ABI", "1")), "conv1x1_subsample": int(_os.environ.get("XR_CONV1X1_SUBSAMPLE", "0")), "dgrad_s2": int(_os.environ.get("XR_DGRAD_S2", "0")),
        "ir_block": int(_os.environ.get("XR_IR_BLOCK", "1")), "sub_pass": int(_os.environ.get("XR_SUB_PASS", "1"))}


if _os.environ.get("XR_DETERMINISTIC", "0") == "1":   # host half of the switch (see set_deterministic); _lib.load() sets the device half
    _cfg["deterministic"], _cfg["direct64"] = 1, 0


def set_compute_dtype(dtype):
    """Activation dtype chosen at module entry for fp32 NCHW inputs: torch.float32 (parity mode: fp32 tensors, every matrix
    operand split into three bf16 planes, six MFMAs per product, ~fp32 accuracy), "fp32x2" (fp32 tensors, TWO planes, three
    MFMAs per product: ~16 significand bits, inside the 1e-3 embedding tolerance at half the matrix work) or torch.bfloat16
    (throughput mode)."""
    if dtype == "fp32x2":
        _cfg["compute_dtype"], _cfg["f32_planes"] = torch.float32, 2
        return
    assert dtype in (torch.float32, torch.bfloat16)
    _cfg["compute_dtype"] = dtype
    if dtype == torch.float32:
        _cfg["f32_planes"] = 3


_det_saved = {}


def set_deterministic(on=True):
    """XR_DETERMINISTIC: every fp32 sum of a step is formed in a fixed order, so two runs of the same step on the same data give
    bit-identical gradients and weights (replica-divergence debugging, exact repeat tests).  Device side (xr_set_deterministic):
    one reduction block per statistics group, weight-gradient slices summed by one group, loss scalars by one block.  Host
    side (here): one statistics group per image, one partial row per convolution tile for the epilogue reductions (dalpha /
    BatchNorm sums; folded in row order by the kernels that consume them), no split-K, and the fused direct 64-channel kernels
    -- whose per-image sums meet by atomics from several workgroups -- give way to the implicit-GEMM path with standalone
    statistics passes.  Slower (the switch is a debugging tool); results differ from the default mode only by summation order."""
    on = bool(on)
    if on == bool(_cfg.get("deterministic")):
        return
    lib.xr_set_deterministic(int(on))
    if on:
        _det_saved.update(direct64=_cfg["direct64"])
        _cfg["direct64"] = 0
    else:
        _cfg["direct64"] = _det_saved.pop("direct64", 1)
    _cfg["deterministic"] = int(on)


def _tile_rows(m, default):
    """Partial rows for a convolution epilogue reduction over m GEMM rows: ``default`` spread rows shared by the tiles (atomics),
    or -- deterministic mode -- one row per tile: every tile configuration covers >= 128 rows, so ceil(m / 128) rows are never
    shared (tile i adds into row i % spread)."""
    return (m + 127) // 128 if _cfg.get("deterministic") else default


def dtc(t):
    """dtype code for the matrix kernels (xr_conv_igemm / xr_conv_wgrad): fp32 tensors run with 3 or 2 bf16 planes."""
    if t.dtype == torch.float32 and _cfg["f32_planes"] == 2:
        return XR_F32X2
    return dt(t)


def get_compute_dtype():
    return _cfg["compute_dtype"]


def r8(c: int) -> int:
    return (c + 7) // 8 * 8


def kg_of(taps: int, cp: int) -> int:
    return (taps * cp + 63) // 64 * 64


def _need_cuda(t):
    if not t.is_cuda:
        raise RuntimeError("xrface: tensors must live on a ROCm device (no CPU fallback on the product path)")


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


class _ZeroPool:
    """Small fp32 scratch buffers that must start at zero (statistic sums, reduction slabs, loss scalars) are carved
    from one pre-zeroed slab: one memset per ~64 MB instead of one fill launch per buffer.  A carved view keeps its
    slab alive; slabs are never reused after being dirtied."""

    def __init__(self, nfloats=1 << 24):
        self.cap = nfloats
        self.buf = None
        self.off = 0

    def get(self, shape, device):
        n = 1
        for d in shape:
            n *= int(d)
        n_al = (n + 3) // 4 * 4
        if n_al > self.cap // 4:
            return torch.zeros(shape, dtype=torch.float32, device=device)
        if self.buf is None or self.buf.device != device or self.off + n_al > self.cap:
            self.buf = torch.zeros(self.cap, dtype=torch.float32, device=device)
            self.off = 0
        # .data: same storage (keeps the slab alive) but its own autograd version counter -- views of one slab would
        # otherwise share a counter and trip save_for_backward's in-place-modification check
        v = self.buf[self.off:self.off + n].view(shape).data
        self.off += n_al
        return v


_zpools = {}     # one pool per stream: a slab is zeroed by a memset on the stream that carves from it

# HIP-graph capture state (xrface.graph.GraphedStep): while a step is being captured, the zero slab is re-created inside the
# capture (so every replay re-zeroes it) and dropout reads a device-side step counter that the graph itself increments
_graph = {"capturing": False, "tick": None, "tick_ref": 0, "side_ok": False}


def _one_stream():
    """A HIP-graph capture keeps everything on the capturing stream unless it was opened with side_stream=True (graph.GraphedStep):
    the weight-gradient fork / join is then captured as two branches of the graph."""
    return _graph["capturing"] and not _graph["side_ok"]



def _zpool_of(device):
    key = (device.index, stream_of(device.index if device.index is not None else torch.cuda.current_device()))
    zp = _zpools.get(key)
    if zp is None:
        zp = _zpools[key] = _ZeroPool()
    return zp


def zeros_f32(shape, device):
    return _zpool_of(device).get(tuple(shape) if not isinstance(shape, int) else (shape,), device)


# steps._pair_grads asks autograd for d loss_k / d theta_k only, but ``needs_input_grad`` of the custom Functions was fixed at
# forward time: every backward the engine traverses would still compute (and, in direct mode, ACCUMULATE) the gradients of
# parameters outside theta_k.  While a per-pair gradient is being taken this holds {id(p) for p in theta_k}: parameters
# outside it get no weight-gradient launch and no in-place accumulation.
_grad_only = [None]


def _wanted(p):
    s = _grad_only[0]
    return s is None or id(p) in s


class grad_only:
    """Context: restrict parameter-gradient work of the HIP backward kernels to ``params``."""

    def __init__(self, params):
        self.ids = {id(p) for p in params}

    def __enter__(self):
        self.prev, _grad_only[0] = _grad_only[0], self.ids
        return self

    def __exit__(self, *exc):
        _grad_only[0] = self.prev
        return False


def _direct(p):
    """Parameters re-pointed into a flat gradient buffer (parallel.FlatParams) take their gradient by in-kernel
    accumulation into ``p.grad`` -- no autograd AccumulateGrad add, no temporary.  Returns the target or None."""
    if getattr(p, "_xr_direct", False) and p.grad is not None and _wanted(p):
        p.__dict__["_xr_touched"] = True
        return p.grad
    return None


def _direct_done(p):
    hook = getattr(p, "_xr_grad_hook", None)
    if hook is not None:
        hook(p)


# Weight gradients have no consumer inside backward (direct mode: they are accumulated into p.grad and first read by the
# all-reduce / optimizer), so they can run on a side stream, concurrently with the dgrad -> norm-backward chain of the
# main stream: their MFMA work fills the ramp-up / tail bubbles and the idle CUs of the main-stream kernels and overlaps
# the HBM-bound elementwise passes.  The main stream re-joins at the end of the backward pass (autograd engine callback)
# and before any gradient bucket is handed to RCCL.
_side = {"stream": None, "dev": None, "ev": None, "seq": 0, "joined": {}, "cb": False}


def join_side_stream():
    """Make the CURRENT stream wait for everything launched on the weight-gradient side stream so far (no-op when this stream
    already joined the latest side launch).  Bookkeeping is per joining stream: with lockstep chains (steps.c4_step / kd_step)
    several streams feed the one side stream, and a gradient bucket launched from one chain's stream must wait for that chain's
    weight gradients even if another chain's stream joined in between."""
    seq = _side["seq"]
    if not seq:
        return
    cur = torch.cuda.current_stream(_side["dev"])
    key = cur.cuda_stream
    if _side["joined"].get(key) != seq:
        cur.wait_stream(_side["stream"])
        _side["joined"][key] = seq


def _join_cb():
    _side["cb"] = False
    join_side_stream()


def _side_fork(dev):
    """The side stream, made to wait for all work enqueued so far on the current stream (one reused event)."""
    if _side["stream"] is None or _side["dev"] != dev:
        _side["stream"], _side["dev"], _side["ev"] = torch.cuda.Stream(dev), dev, torch.cuda.Event()
        _side["seq"], _side["joined"] = 0, {}
    ev = _side["ev"]
    ev.record()
    _side["stream"].wait_event(ev)
    return _side["stream"]


def _side_done(side, tensors):
    """Bookkeeping after side-stream launches: the caching allocator must not hand `tensors` out again before the side
    kernels ran; the main stream re-joins at the end of the backward pass."""
    for t in tensors:
        t.record_stream(side)
    _side["seq"] += 1
    if not _side["cb"]:
        try:
            torch.autograd.Variable._execution_engine.queue_callback(_join_cb)
            _side["cb"] = True
        except RuntimeError:   # not inside a backward pass: join right away
            join_side_stream()


def _wgrad64_ok(x, dy, H, W, Cp, Ho, Wo, K, R, S, stride, pad, transposed):
    """bf16 64 -> 64 3x3 / stride 1 / pad 1 with W % 8 == 0, W <= 112: the direct weight-gradient kernel (csrc/xr_wgrad64.hip)."""
    return (_cfg["direct64"] and x.dtype == torch.bfloat16 and Cp == 64 and K == 64 and R == 3 and S == 3 and stride == 1 and pad == 1
            and not transposed and Ho == H and Wo == W and W % 8 == 0 and W <= 112 and dy.shape[-1] == 64
            and x.numel() * 2 < (1 << 31))


def _wgrad_rows_ok(x, dy, H, W, Cp, Ho, Wo, K, R, S, stride, pad, transposed):
    """bf16 3x3 / pad 1 / stride 1 or 2 with 64-multiple channel counts and an output at least 14 wide: the row-walking direct
    weight-gradient kernel (csrc/xr_wgrad_rows.hip)."""
    mode = _cfg["wgrad_rows"]        # 0 off, 1 every eligible layer, 2 only the layers where it won the in-step A/B
    if mode == 2 and not ((stride == 1 and Cp <= 128) or (stride == 2 and Cp == 64)):
        return False
    return (mode and x.dtype == torch.bfloat16 and R == 3 and S == 3 and pad == 1 and stride in (1, 2) and not transposed
            and Cp % 64 == 0 and K % 64 == 0 and dy.shape[-1] == K and H % stride == 0 and W % stride == 0 and Ho == H // stride
            and Wo == W // stride and 14 <= Wo <= (112 if stride == 1 else 64) and x.numel() * 2 < (1 << 31)
            and dy.numel() * 2 < (1 << 31))


def _wgrad(w, x, dy, N, H, W, Cp, Ho, Wo, K, R, S, stride, pad, transposed, Kp, kg, split, A1, A2, taps, B, Bp, sa1, sa2, st, sb,
           xform=None):
    """Weight gradient: sliced implicit GEMM (or, for the 64-channel 3x3 layers, the direct row-walking kernel) into partial
    slabs, then sum + convert to the parameter layout (accumulating straight into ``w.grad`` when direct mode is on).
    ``xform`` = (scale [N][64], shift [N][64], alpha [64] or None): x is transformed on load (direct kernel only)."""
    tgt = _direct(w)
    d64 = _wgrad64_ok(x, dy, H, W, Cp, Ho, Wo, K, R, S, stride, pad, transposed)
    assert xform is None or d64, "on-load transform needs the direct 64-channel weight-gradient kernel"
    rows_ok = _wgrad_rows_ok(x, dy, H, W, Cp, Ho, Wo, K, R, S, stride, pad, transposed)
    if d64 and xform is None and W > 64 and rows_ok and _cfg["wgrad_rows112"]:
        # 112-wide maps without the on-load transform: the row-walking kernel's two-rows-per-step plan is faster in isolation (966 vs
        # 822 TFLOP/s) and in the FHN step alone (C3 37.4 -> 36.4 ms), but its 145 KB workgroups cost the composed step more than
        # they save beside the assistant's backward (C4 109.0 -> 110.0 ms): off by default (XR_WGRAD_ROWS112=1)
        d64 = False
    rows = (not d64) and rows_ok
    if d64:
        split = min(256, N * H)
    elif rows:
        split = max(1, 256 // ((K // 64) * (Cp // 64)))

    def launch(slabs, sh, on=None):
        pe = _probe_begin(("wgrad", Cp, K, H, W, R, stride), on)
        if d64:
            sc, sf, al = xform if xform is not None else (None, None, None)
            ns = lib.xr_conv64_wgrad(ptr(x), ptr(dy), ptr(slabs), N, H, W, split, ptr(sc), ptr(sf), ptr(al), sh)
        elif rows:
            ns = lib.xr_conv_wgrad_rows(ptr(x), ptr(dy), ptr(slabs), N, H, W, Cp, K, stride, split, sh)
        else:
            ns = lib.xr_conv_wgrad(dtc(x), ptr(x), ptr(dy), ptr(slabs), N, H, W, Cp, Ho, Wo, K, R, S, stride, pad, transposed, Kp,
                                   kg, split, sh)
        if pe is not None:
            if on is None:
                pe.record()
            else:
                pe.record(on)
        return ns

    # (not while a HIP graph is being captured: graphs stay single-stream -- a captured fork/join brought nothing at the
    # small batch sizes graphs are for, and multi-stream graph teardown is the less-trodden path of the runtime)
    # (wgrad64_stream: the direct 64-channel weight-gradient kernel keeps one persistent 147 KB-LDS workgroup per CU; beside the
    # direct forward / input-gradient kernels of the main stream (120 KB) the two only take turns on the CUs, so in the FHN step
    # alone the side stream buys nothing for it (C3 N = 128: 36.41 vs 36.42 ms in the interleaved A/B) -- in the composed step it
    # still overlaps the IR-SE-50 chains: C4 105.2 vs 106.7 ms, so it stays on)
    if tgt is not None and _cfg["wgrad_stream"] and not _one_stream() and (not d64 or _cfg["wgrad64_stream"]):
        dev = x.device
        side = _side_fork(dev)     # side stream now waits for everything enqueued on the current stream (x, dy, zeroed grads)
        sh = side.cuda_stream      # launch on the side stream by handle: no current-stream switch on the host
        slabs = torch.empty((split, K, kg), dtype=torch.float32, device=dev)
        ns = launch(slabs, sh, side)
        lib.xr_unpack_wgrad(ptr(slabs), ptr(tgt), A1, A2, taps, B, Bp, kg, sa1, sa2, st, sb, 1, ns, sh)
        _side_done(side, (x, dy, slabs) + (tuple(t for t in xform if t is not None) if xform is not None else ()))
        _direct_done(w)
        return None
    slabs = torch.empty((split, K, kg), dtype=torch.float32, device=x.device)
    ns = launch(slabs, stream())
    if tgt is not None:
        lib.xr_unpack_wgrad(ptr(slabs), ptr(tgt), A1, A2, taps, B, Bp, kg, sa1, sa2, st, sb, 1, ns, stream())
        _direct_done(w)
        return None
    dw = torch.empty_like(w, dtype=torch.float32)
    lib.xr_unpack_wgrad(ptr(slabs), ptr(dw), A1, A2, taps, B, Bp, kg, sa1, sa2, st, sb, 0, ns, stream())
    return dw


def _emit_small(p, val):
    """Small fp32 gradient vector (bias / gamma / beta / alpha): add into p.grad directly when enabled."""
    tgt = _direct(p)
    if tgt is not None and val is not None:
        tgt.add_(val.view_as(tgt))
        _direct_done(p)
        return None
    return val


# ------------------------------------------------------------------------------------------------- launch probe
def _probe_begin(tag, on=None):
    """bench.py times convolution launches INSIDE the training step with HIP event pairs recorded on the stream the kernel is
    launched on (`on`: a torch.cuda.Stream, default the current one).  _cfg["probe"] = {"tag": t, "events": []} brackets the
    launches of ONE site family (kind, C, K, H, W, R, stride); {"all": {}} brackets every family (an untimed survey step).
    No-op (one dict lookup) otherwise.  Returns the end event (the caller records it after the launch) or None."""
    pr = _cfg.get("probe")
    if pr is None:
        return None
    every = pr.get("all")
    if every is None and pr.get("tag") != tag:
        return None
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    if on is None:
        e0.record()
    else:
        e0.record(on)
    (every.setdefault(tag, []) if every is not None else pr["events"]).append((e0, e1))
    return e1


# ------------------------------------------------------------------------------------------------- weight packs
_pack_epoch = [0]


# GraphedStep's warm-up records which parameters the step's fused optimizers update and which BatchNorm layers run in training
# mode (their running statistics move): a replay then invalidates exactly those caches instead of everything
_touch_log = [None]      # None | {"params": {id: weakref}, "bns": {id: weakref}}


def invalidate_weight_cache(params=None):
    """Fused optimizers update parameters through raw pointers (no version bump): they call this, with the parameters
    they touched (packs of other, e.g. frozen-teacher, parameters stay valid) or without (everything is stale)."""
    if params is None:
        _pack_epoch[0] += 1
    else:
        log = _touch_log[0]
        for p in params:
            p.__dict__["_xr_epoch"] = p.__dict__.get("_xr_epoch", 0) + 1
            if log is not None:
                log["params"][id(p)] = weakref.ref(p)


def _pack_tag(w):
    return (w._version, w.data_ptr(), _pack_epoch[0], w.__dict__.get("_xr_epoch", 0), tuple(w.shape))


class _PackPlan:
    """Every (parameter, pack form) seen so far; when a registered pack is found stale, ALL stale registered packs are
    refreshed by ONE launch (xr_pack_run over a device descriptor table) -- instead of ~110 per-layer pack launches after
    each optimizer step.  Tables are cached per stale set (a training loop has a few: {student}, {assistant}, ...; packs of
    a frozen teacher are never stale and never re-packed).  Pack buffers are persistent: a refresh overwrites them in
    stream order, after every kernel of the previous step that read them."""

    def __init__(self):
        self.entries = []     # [weakref(w), key, pk, kg, args]
        self.index = {}       # (id(w), key) -> entry
        self.tables = {}      # tuple(entry ids) -> (table, (n, blocks, smem), [src ptrs])
        self.builds = 0       # descriptor tables built so far (GraphedStep warms up until a step builds none)

    def add(self, w, key, pk, kg, args):
        e = [weakref.ref(w), key, pk, kg, args]
        self.entries.append(e)
        self.index[(id(w), key)] = e

    def _build(self, sel):
        import numpy as np
        ent = np.zeros((len(sel), 14), dtype=np.int64)
        for i, (w, e) in enumerate(sel):
            planes, A1, A2, taps, B, Bp, sa1, sa2, st, sb = e[4]
            ent[i] = (w.data_ptr(), e[2].data_ptr(), planes, A1, A2, taps, B, Bp, e[3], sa1, sa2, st, sb, 0)
        table = torch.empty(len(sel) * 128, dtype=torch.uint8, device=sel[0][0].device)
        smem = ctypes.c_int(0)
        blocks = lib.xr_pack_plan(ent.ctypes.data, len(sel), ptr(table), ctypes.addressof(smem), stream())
        return table, (len(sel), blocks, smem.value), [w.data_ptr() for w, _ in sel]

    def prebuild(self):
        """Build (without running) the descriptor table of the CURRENTLY stale set.  GraphedStep calls this right before it
        captures: the first convolution of the captured step then finds its table cached -- building one uploads from the
        host and waits, which a capture cannot contain."""
        stale = []
        for e in self.entries:
            w = e[0]()
            hit = None if w is None else w.__dict__.get("_xr_pack", {}).get(e[1])
            if hit is not None and hit[1] is e[2] and hit[0] != _pack_tag(w) and w.dtype == torch.float32 and w.is_contiguous():
                stale.append((w, e))
        if not stale:
            return
        key = tuple(id(e) for _, e in stale)
        ent = self.tables.get(key)
        if ent is None or ent[2] != [w.data_ptr() for w, _ in stale] or ent[0].device != stale[0][0].device:
            if len(self.tables) >= 8:
                self.tables.clear()
            self.tables[key] = self._build(stale)
            self.builds += 1

    def refresh(self):
        stale, dead = [], False
        for e in self.entries:
            w = e[0]()
            hit = None if w is None else w.__dict__.get("_xr_pack", {}).get(e[1])
            if hit is None or hit[1] is not e[2] or w.dtype != torch.float32 or not w.is_contiguous():
                dead = True          # parameter gone / re-packed elsewhere: drop the entry
                e[0] = None
            elif hit[0] != _pack_tag(w):
                stale.append((w, e))
        if dead:
            self.entries = [e for e in self.entries if e[0] is not None]
            self.index = {(id(e[0]()), e[1]): e for e in self.entries}
            self.tables.clear()
        if not stale:
            return
        key = tuple(id(e) for _, e in stale)
        ent = self.tables.get(key)
        if ent is None or ent[2] != [w.data_ptr() for w, _ in stale] or ent[0].device != stale[0][0].device:
            if _graph["capturing"]:
                # a stale set never seen before shows up inside a HIP-graph capture: building its descriptor table uploads
                # from the host and waits (not capturable) -- refresh these packs one launch each instead
                for w, e in stale:
                    planes, A1, A2, taps, B, Bp, sa1, sa2, st, sb = e[4]
                    lib.xr_pack_weight(ptr(w.detach()), ptr(e[2]), planes, A1, A2, taps, B, Bp, e[3], sa1, sa2, st, sb, stream())
                    w.__dict__["_xr_pack"][e[1]] = (_pack_tag(w), e[2], e[3])
                return
            if len(self.tables) >= 8:
                self.tables.clear()
            ent = self.tables[key] = self._build(stale)
            self.builds += 1
        n, blocks, smem = ent[1]
        lib.xr_pack_run(ptr(ent[0]), n, blocks, smem, stream())
        for w, e in stale:
            w.__dict__["_xr_pack"][e[1]] = (_pack_tag(w), e[2], e[3])


_pack_plan = _PackPlan()


def _packed(w, kind, dtype, A1, A2, taps, B, Bp, sa1, sa2, st, sb):
    """Pack an fp32 parameter for the implicit-GEMM kernels.  The pack is cached ON the parameter object
    (so it dies with it) and is refreshed whenever the parameter's version / storage / epoch changes."""
    cache = w.__dict__.setdefault("_xr_pack", {})
    planes = _cfg["f32_planes"] if dtype == torch.float32 else 1
    key = (kind, dtype, Bp, planes)
    tag = _pack_tag(w)
    hit = cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1], hit[2]
    if hit is not None and _cfg["pack_plan"] and (id(w), key) in _pack_plan.index and w.device.type == "cuda":
        _pack_plan.refresh()      # stale registered pack: refresh ALL registered packs in one launch
        hit = cache.get(key)
        if hit is not None and hit[0] == _pack_tag(w):
            return hit[1], hit[2]
    kg = kg_of(taps, Bp)
    rows = A1 * A2
    pk = torch.empty((planes, rows, kg), dtype=torch.bfloat16, device=w.device)
    wd = w.detach()
    plain = wd.dtype == torch.float32 and wd.is_contiguous()
    if not plain:
        wd = wd.float().contiguous()
    lib.xr_pack_weight(ptr(wd), ptr(pk), planes, A1, A2, taps, B, Bp, kg, sa1, sa2, st, sb, stream())
    cache[key] = (tag, pk, kg)
    if plain and _cfg["pack_plan"] and w.device.type == "cuda" and not (taps == 1 and sa1 == 1 and A1 <= 64 and A2 > 1):
        # (the Linear input-gradient form keeps its own LDS-tiled kernel: its tile does not fit the batch kernel's LDS budget)
        _pack_plan.add(w, key, pk, kg, (planes, A1, A2, taps, B, Bp, sa1, sa2, st, sb))
    return pk, kg


def _dgrad_s2_ok(x_shape, w, stride, pad, Ho, Wo):
    """3x3 / stride 2 / pad 1 with even input extent: the dense 2x2-window input gradient (xr_conv_dgrad_s2) applies.  OFF by default
    (XR_DGRAD_S2=1): isolated at batch 256 it ties the class-mode strided gather on all four IR-SE-50 stage openings (64 ch @112: 287
    vs 257 us; 128 @56: 189 vs 196; 256 @28: 159 vs 158; 512 @14: 134 vs 142) -- 16/9 of the MACs on the 4-wave kernel cost what the
    dense gather saves -- and the C2 step did not move (19.43 vs 19.30 ms); it would need the 8-wave kernel's rate to pay."""
    N, H, W, Cp = x_shape
    K, C, R, S = w.shape
    return (_cfg["dgrad_s2"] and R == 3 and S == 3 and stride == 2 and pad == 1 and H % 2 == 0 and W % 2 == 0 and Ho * 2 == H
            and Wo * 2 == W and C % 8 == 0 and K % 8 == 0)


def _packed_s2(w, dtype):
    """[planes][4 * C][4 * K] pack of a [K][C][3][3] parameter for xr_conv_dgrad_s2 (cached on the parameter like _packed; not part of
    the batched refresh: four small launches per network and step)."""
    cache = w.__dict__.setdefault("_xr_pack", {})
    planes = _cfg["f32_planes"] if dtype == torch.float32 else 1
    key = ("s2dgrad", dtype, planes)
    tag = _pack_tag(w)
    hit = cache.get(key)
    if hit is not None and hit[0] == tag:
        return hit[1], hit[2]
    K, C = w.shape[0], w.shape[1]
    kg = kg_of(4, K)
    pk = hit[1] if hit is not None else torch.empty((planes, 4 * C, kg), dtype=torch.bfloat16, device=w.device)
    wd = w.detach()
    if wd.dtype != torch.float32 or not wd.is_contiguous():
        wd = wd.float().contiguous()
    lib.xr_pack_dgrad_s2(ptr(wd), ptr(pk), planes, K, C, K, C, kg, stream())
    cache[key] = (tag, pk, kg)
    return pk, kg


def _wgrad_split(M, K, kg):
    tiles = ((K + 127) // 128) * ((kg + 127) // 128) if (K > 64 or kg <= 128) else ((kg + 255) // 256)
    steps = (M + 63) // 64
    # a one-tile problem (the 3 -> 64 stems: 32 KB slabs) takes two workgroups per CU: 253 -> 144 us at 112 x 112, batch 256
    cap = 512 if (tiles == 1 and K * kg <= 16384) else 256
    return max(1, min(steps, _cfg["wgrad_blocks"] // max(tiles, 1), cap))


# ------------------------------------------------------------------------------------------------- layout
class _Enter(Function):
    """NCHW fp32 (contiguous) -> NHWC compute dtype, channels zero-padded to a multiple of 8."""

    @staticmethod
    def forward(ctx, x, dtype):
        _need_cuda(x)
        N, C, H, W = x.shape
        ctx.shape = (N, C, H, W)
        xs = _c(x.detach().float())
        out = torch.empty((N, H, W, r8(C)), dtype=dtype, device=x.device)
        lib.xr_nchw_to_nhwc(dt(out), ptr(xs), ptr(out), N, C, H, W, r8(C), stream())
        return out

    @staticmethod
    def backward(ctx, g):
        N, C, H, W = ctx.shape
        g = _c(g)
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=g.device)
        lib.xr_nhwc_to_nchw(dt(g), ptr(g), ptr(out), N, C, H, W, g.shape[3], stream())
        return out, None


class _Leave(Function):
    """NHWC buffer -> plain NCHW fp32 (used for tensors whose channel count is not a multiple of 8)."""

    @staticmethod
    def forward(ctx, buf, C):
        N, H, W, Cp = buf.shape
        ctx.meta = (C, Cp, buf.dtype)
        out = torch.empty((N, C, H, W), dtype=torch.float32, device=buf.device)
        lib.xr_nhwc_to_nchw(dt(buf), ptr(buf), ptr(out), N, C, H, W, Cp, stream())
        return out

    @staticmethod
    def backward(ctx, g):
        C, Cp, dtype = ctx.meta
        N, _, H, W = g.shape
        gs = _c(g.float())
        out = torch.empty((N, H, W, Cp), dtype=dtype, device=g.device)
        lib.xr_nchw_to_nhwc(dt(out), ptr(gs), ptr(out), N, C, H, W, Cp, stream())
        return out, None


def enter(x, dtype=None):
    """User tensor (N,C,H,W) or (N,F) -> internal NHWC buffer [N,H,W,Cp].  Zero-copy for channels_last
    feature maps with C % 8 == 0 already in a compute dtype."""
    _need_cuda(x)
    if x.dim() == 2:
        x = x.reshape(x.shape[0], x.shape[1], 1, 1)
    if x.dim() != 4:
        raise RuntimeError(f"xrface: expected a 4-D NCHW tensor, got shape {tuple(x.shape)}")
    C = x.shape[1]
    if x.dtype in (torch.float32, torch.bfloat16) and C % 8 == 0 and (dtype is None or dtype == x.dtype):
        v = x.permute(0, 2, 3, 1)
        if v.is_contiguous() and v.data_ptr() % 16 == 0:
            return v
    return _Enter.apply(x, dtype or _cfg["compute_dtype"])


def leave(buf, C=None):
    """Internal NHWC buffer -> user-facing tensor of logical shape (N,C,H,W): a zero-copy channels_last view
    when no channel padding is involved, otherwise a plain NCHW fp32 tensor."""
    Cp = buf.shape[3]
    C = Cp if C is None else C
    if C == Cp:
        return buf.permute(0, 3, 1, 2)
    return _Leave.apply(buf, C)


def leave2d(buf):
    """[N,1,1,C] -> (N, C)."""
    return buf.reshape(buf.shape[0], buf.shape[3])


# ------------------------------------------------------------------------------------------------- convolution
def _bias_grad(dy, K):
    N, Ho, Wo, Kp = dy.shape
    # One group over all N * Ho * Wo rows sends every block's 2 * Kp atomics to the same few cache lines: 784 blocks on a
    # [32][56][56][64] gradient took 31 us (12.8 MB: 5 us of streaming) -- the SR-variant bottlenecks have three biased
    # convolutions each.  Per-image sums (the InstanceNorm geometry: blocks of one image meet in their own lines) + a fold of N
    # rows: ~11 us.  Also the deterministic form: one block per image, the images summed in a fixed order.
    if N > 1 and (_cfg.get("deterministic") or N * Ho * Wo >= 16384):
        sums = zeros_f32((2, N, Kp), dy.device)
        lib.xr_group_stats(dt(dy), ptr(dy), ptr(sums), N, Ho * Wo, Kp, stream())
        out = torch.empty((Kp,), dtype=torch.float32, device=dy.device)
        lib.xr_reduce_groups(ptr(sums), ptr(out), 1, N, Kp, 0, stream())
        return out[:K]
    sums = zeros_f32((2, 1, Kp), dy.device)
    lib.xr_group_stats(dt(dy), ptr(dy), ptr(sums), 1, N * Ho * Wo, Kp, stream())
    return sums[0, 0, :K]


_IDENT = {}


def _ident_coefs(n, device):
    """[n][64] ones / zeros (the identity transform for the direct kernel's backward-reduction epilogue), cached per batch size."""
    key = (n, device.index)
    if key not in _IDENT:
        _IDENT[key] = (torch.ones((n, 64), dtype=torch.float32, device=device), torch.zeros((n, 64), dtype=torch.float32, device=device))
    return _IDENT[key]


class _Conv2d(Function):
    """aten::conv2d replacement (weight [K][C][R][S] fp32 Parameter; x NHWC buffer)."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, prelu_alpha=None, bn_link=None, stats_link=None, passthrough=False):
        _need_cuda(x)
        ctx.set_materialize_grads(False)   # the non-differentiable PReLU output must not cost a zero-filled gradient tensor
        x = _c(x)
        N, H, W, Cp = x.shape
        K, C, R, S = w.shape
        assert r8(C) == Cp, f"conv: input pitch {Cp} does not match weight C={C}"
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
        pk, kg = _packed(w, "fwd", x.dtype, K, 1, R * S, C, Cp, C * R * S, 0, 1, R * S)
        Kp = r8(K)
        y = torch.empty((N, Ho, Wo, Kp), dtype=x.dtype, device=x.device)
        bf = None if b is None else _c(b.detach().float())
        ctx.direct64 = direct64_ok(x, w, stride, pad) and prelu_alpha is None and bn_link is None
        if ctx.direct64:
            # 64 -> 64 3x3: weights-stationary direct kernel; a following training-mode BatchNorm gets its statistics from the
            # per-image sums of the epilogue (StatsLink folds them over the batch)
            sred = None
            if stats_link is not None and _cfg["fuse_conv_stats"]:
                sred = zeros_f32((2, N, Kp), x.device)
            _conv64(x, pk, y, bias=bf, stats=sred, tag=("fwd", Cp, K, H, W, R, stride))
            if sred is not None:
                stats_link.deliver(y, sred)
            ctx.save_for_backward(x, w)
            ctx.geom = (stride, pad, b is not None)
            ctx.bias_ref = b
            ctx.bn_link = None
            ctx.passthrough = passthrough
            if ctx.passthrough:
                return y, x.view_as(x)
            return y
        # prelu_alpha: also emit p = prelu(y, alpha) from the epilogue (second, non-differentiable output): the activation
        # pass of conv -> PReLU -> conv disappears; its backward still rides in the next conv's dgrad epilogue
        p2 = al = None
        if prelu_alpha is not None and K % 8 == 0 and _cfg["fuse_prelu"]:
            al = _c(prelu_alpha.detach().float())
            p2 = torch.empty_like(y)
        # stats_link: the BatchNorm that follows wants sum / sum of squares of y per channel: taken in the epilogue
        sred, ssp = None, 1
        if stats_link is not None and p2 is None and K % 8 == 0 and _cfg["fuse_conv_stats"]:
            ssp = _tile_rows(N * Ho * Wo, StatsLink.SPREAD)
            sred = zeros_f32((3, ssp, Kp), x.device)
        pe = _probe_begin(("fwd", Cp, K, H, W, R, stride))
        if p2 is not None and bf is None and sred is None and _cfg["direct64_prelu"] and direct64_ok(x, w, stride, pad):
            # conv1 of the 64-channel IR units: the direct kernel with the PReLU second output (its backward keeps the
            # implicit-GEMM / bwdred paths below: ctx.direct64 stays False)
            lib.xr_conv64_direct_prelu(ptr(x), ptr(pk), ptr(y), ptr(p2), ptr(al), N, H, W, stream())
        else:
            lib.xr_conv_igemm(dtc(x), ptr(x), ptr(pk), ptr(bf), ptr(y), N, H, W, Cp, Ho, Wo, K, R, S, stride, pad, 0,
                              kg, Kp, None, 0, None, ptr(al), None, ssp, ptr(p2), ptr(sred), None, stream())
        if sred is not None:
            stats_link.deliver(y, sred)
        if pe is not None:
            pe.record()
        ctx.save_for_backward(x, w)
        ctx.geom = (stride, pad, b is not None)
        ctx.bias_ref = b
        ctx.bn_link = bn_link
        ctx.passthrough = passthrough and prelu_alpha is None
        if ctx.passthrough:   # second output aliases the input: its gradient is summed into dx by the dgrad epilogue (ep_add)
            return y, x.view_as(x)
        if prelu_alpha is not None:
            if p2 is None:
                p2 = torch.empty_like(y)
                al = _c(prelu_alpha.detach().float())
                lib.xr_affine_act(dt(y), ptr(y), None, None, None, ptr(al), ACT_PRELU, ptr(p2), 1, N * Ho * Wo, Kp, 0, stream())
            ctx.mark_non_differentiable(p2)
            return y, p2
        return y

    @staticmethod
    def backward(ctx, dy, _dp2=None):
        if dy is None:
            return (_dp2 if ctx.passthrough else None), None, None, None, None, None, None, None, None
        x, w = ctx.saved_tensors
        stride, pad, has_b = ctx.geom
        dy = _c(dy)
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        N, H, W, Cp = x.shape
        K, C, R, S = w.shape
        _, Ho, Wo, Kp = dy.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0] and getattr(ctx, "direct64", False):
            pk, kg = _packed(w, "dgrad", x.dtype, C, 1, R * S, K, Kp, R * S, 0, 1, C * R * S)
            dpass = _dp2 if ctx.passthrough else None
            if dpass is not None:
                dpass = _c(dpass)
                if dpass.dtype != x.dtype:
                    dpass = dpass.to(x.dtype)
            dx = _conv64(dy, pk, torch.empty_like(x), transposed=1, add=dpass, tag=("dgrad", Cp, K, H, W, R, stride))
        elif ctx.needs_input_grad[0]:
            pk, kg = _packed(w, "dgrad", x.dtype, C, 1, R * S, K, Kp, R * S, 0, 1, C * R * S)
            dx = torch.empty_like(x)
            link, red, sp_ = ctx.bn_link, None, 1
            if link is not None and link.x is not None and _cfg["fuse_bn_reduce"] and C % 8 == 0 and link.x.shape == x.shape \
                    and link.x.dtype == x.dtype:
                sp_ = _tile_rows(N * H * W, BnLink.SPREAD)
                red = zeros_f32((3, sp_, C), x.device)
            dpass = _dp2 if ctx.passthrough else None
            if dpass is not None:
                dpass = _c(dpass)
                if dpass.dtype != x.dtype:
                    dpass = dpass.to(x.dtype)
            add_in_kernel = dpass is not None and red is None and C % 8 == 0
            pe = _probe_begin(("dgrad", Cp, K, H, W, R, stride))
            if _dgrad_s2_ok(x.shape, w, stride, pad, Ho, Wo) and K == Kp and C == Cp:
                # 3x3 / stride 2: one dense 2x2-window GEMM with a depth-to-space epilogue (xr_conv_dgrad_s2); the BatchNorm-backward
                # partial sums then come per (spread row, sub-pixel class) -- folded like any other partial rows
                pk2, kg2 = _packed_s2(w, x.dtype)
                if red is not None:
                    red = zeros_f32((3, sp_ * 4, C), x.device)
                lib.xr_conv_dgrad_s2(dtc(x), ptr(dy), ptr(pk2), ptr(dx), N, Ho, Wo, Kp, Cp, kg2, ptr(link.x) if red is not None else None,
                                     None, None, sp_, ptr(red), ptr(dpass) if add_in_kernel else None, stream())
            elif red is not None and dpass is None and direct64_ok(x, w, stride, pad) and Kp == 64:
                # 64 -> 64 3x3 (conv1 of the IR stage-1 blocks, model_irse.py:56-59): the direct kernel with the BatchNorm-backward
                # sums in its epilogue -- sum d and sum d * x per image (identity transform: scale 1, shift 0, no slope), folded over
                # the batch by xr_norm_bwd_coeffs like the spread partials of the implicit GEMM
                red = zeros_f32((3, N, C), x.device)
                one, zero = _ident_coefs(N, x.device)
                lib.xr_conv64_direct_bwdred(ptr(dy), ptr(pk), ptr(dx), N, H, W, 1, ptr(link.x), ptr(one), ptr(zero), None, ptr(red),
                                            stream())
            else:
                lib.xr_conv_igemm(dtc(x), ptr(dy), ptr(pk), None, ptr(dx), N, Ho, Wo, Kp, H, W, C, R, S, stride, pad, 1,
                                  kg, Cp, None, 0, ptr(link.x) if red is not None else None, None, None, sp_, None, ptr(red),
                                  ptr(dpass) if add_in_kernel else None, stream())
            if pe is not None:
                pe.record()
            if dpass is not None and not add_in_kernel:
                dx = dx + dpass
            if red is not None:
                link.deliver(dx, red)
        if dx is None and ctx.passthrough and _dp2 is not None:
            dx = _dp2
        if ctx.needs_input_grad[1] and _wanted(w):
            kg = kg_of(R * S, Cp)
            dw = _wgrad(w, x, dy, N, H, W, Cp, Ho, Wo, K, R, S, stride, pad, 0, Kp, kg, _wgrad_split(N * Ho * Wo, K, kg),
                        K, 1, R * S, C, Cp, C * R * S, 0, 1, R * S)
        if has_b and ctx.needs_input_grad[2] and _wanted(ctx.bias_ref):
            db = _emit_small(ctx.bias_ref, _bias_grad(dy, K))
        return dx, dw, db, None, None, None, None, None, None


class _ConvTranspose2d(Function):
    """aten::conv_transpose2d replacement (weight [Cin][Cout][R][S]); model/FSRnet.py:436."""

    @staticmethod
    def forward(ctx, x, w, b, stride, pad, out_pad):
        x = _c(x)
        N, H, W, Cp = x.shape
        Cin, Cout, R, S = w.shape
        assert r8(Cin) == Cp
        Ho, Wo = (H - 1) * stride - 2 * pad + R + out_pad, (W - 1) * stride - 2 * pad + S + out_pad
        assert (Ho + 2 * pad - R) // stride + 1 == H and (Wo + 2 * pad - S) // stride + 1 == W, \
            "deconv geometry not expressible as the transposed gather"
        pk, kg = _packed(w, "tfwd", x.dtype, Cout, 1, R * S, Cin, Cp, R * S, 0, 1, Cout * R * S)
        Kp = r8(Cout)
        y = torch.empty((N, Ho, Wo, Kp), dtype=x.dtype, device=x.device)
        bf = None if b is None else _c(b.detach().float())
        lib.xr_conv_igemm(dtc(x), ptr(x), ptr(pk), ptr(bf), ptr(y), N, H, W, Cp, Ho, Wo, Cout, R, S, stride, pad, 1,
                          kg, Kp, None, 0, None, None, None, 1, None, None, None, stream())
        ctx.save_for_backward(x, w)
        ctx.geom = (stride, pad, b is not None)
        ctx.bias_ref = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        stride, pad, has_b = ctx.geom
        dy = _c(dy)
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        N, H, W, Cp = x.shape
        Cin, Cout, R, S = w.shape
        _, Ho, Wo, Kp = dy.shape
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # dx[n,hi,wi,ci] = sum dy[n, hi*s - p + r, ., co] w[ci][co][r][s]: an ordinary strided conv over dy
            pk, kg = _packed(w, "tdgrad", x.dtype, Cin, 1, R * S, Cout, Kp, Cout * R * S, 0, 1, R * S)
            dx = torch.empty_like(x)
            lib.xr_conv_igemm(dtc(x), ptr(dy), ptr(pk), None, ptr(dx), N, Ho, Wo, Kp, H, W, Cin, R, S, stride, pad, 0,
                              kg, Cp, None, 0, None, None, None, 1, None, None, None, stream())
        if ctx.needs_input_grad[1] and _wanted(w):
            # dw[ci][co][r][s] = sum_{n,hi,wi} x[n,hi,wi,ci] * dy[n, hi*s - p + r, wi*s - p + s', co]: the weight gradient of an
            # ordinary strided convolution whose input is dy and whose output gradient is x -- rows = ci, cols = (tap, co),
            # forward gather (the fast incremental-cursor path) instead of a transposed gather of x
            kg = kg_of(R * S, Kp)
            dw = _wgrad(w, dy, x, N, Ho, Wo, Kp, H, W, Cin, R, S, stride, pad, 0, Cp, kg,
                        _wgrad_split(N * H * W, Cin, kg), Cin, 1, R * S, Cout, Kp, Cout * R * S, 0, 1, R * S)
        if has_b and ctx.needs_input_grad[2] and _wanted(ctx.bias_ref):
            db = _emit_small(ctx.bias_ref, _bias_grad(dy, Cout))
        return dx, dw, db, None, None, None


class _LinearNHWC(Function):
    """Flatten(C,H,W order) + nn.Linear on an NHWC buffer == a full-extent H x W convolution
    (model_irse.py:146-147, model/resnet.py:221-222).  x: [N,H,W,C] with C % 8 == 0; returns [N,1,1,Kp]."""

    @staticmethod
    def forward(ctx, x, w, b):
        x = _c(x)
        N, H, W, C = x.shape
        K, F = w.shape
        assert F == C * H * W, f"linear: in_features {F} != {C}*{H}*{W}"
        HW = H * W
        pk, kg = _packed(w, "lin_fwd", x.dtype, K, 1, HW, C, C, C * HW, 0, 1, HW)
        Kp = r8(K)
        y = torch.empty((N, 1, 1, Kp), dtype=x.dtype, device=x.device)
        bf = None if b is None else _c(b.detach().float())
        tiles = ((N + 127) // 128) * ((Kp + 127) // 128)
        split = min(kg // 64, max(1, 512 // tiles)) if (kg >= 4096 and not _cfg.get("deterministic")) else 1
        if split > 1:  # long reduction, few output tiles: split-K with an fp32 workspace
            ws = zeros_f32((N, Kp), x.device)
            lib.xr_conv_igemm(dtc(x), ptr(x), ptr(pk), None, None, N, H, W, C, 1, 1, K, H, W, 1, 0, 0, kg, Kp, ptr(ws), split, None, None, None, 1, None, None, None,
                              stream())
            lib.xr_bias_cast(dt(y), ptr(ws), ptr(bf), ptr(y), N, K, Kp, stream())
        else:
            lib.xr_conv_igemm(dtc(x), ptr(x), ptr(pk), ptr(bf), ptr(y), N, H, W, C, 1, 1, K, H, W, 1, 0, 0, kg, Kp, None, 0, None, None, None, 1, None, None, None,
                              stream())
        ctx.save_for_backward(x, w)
        ctx.has_b = b is not None
        ctx.bias_ref = b
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy = _c(dy)
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        N, H, W, C = x.shape
        K, F = w.shape
        HW, Kp = H * W, dy.shape[3]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # plain GEMM dx[n][p*C + c] = sum_k dy[n][k] w[k][c*HW + p]: a 1x1 "conv" with HW*C output channels
            pk, kg = _packed(w, "lin_dgrad", x.dtype, HW, C, 1, K, Kp, 1, HW, 0, C * HW)
            dx = torch.empty_like(x)
            lib.xr_conv_igemm(dtc(x), ptr(dy), ptr(pk), None, ptr(dx), N, 1, 1, Kp, 1, 1, HW * C, 1, 1, 1, 0, 0, kg,
                              HW * C, None, 0, None, None, None, 1, None, None, None, stream())
        if ctx.needs_input_grad[1] and _wanted(w):
            kg = kg_of(HW, C)
            dw = _wgrad(w, x, dy, N, H, W, C, 1, 1, K, H, W, 1, 0, 0, Kp, kg, 1, K, 1, HW, C, C, C * HW, 0, 1, HW)
        if ctx.has_b and ctx.needs_input_grad[2] and _wanted(ctx.bias_ref):
            db = _emit_small(ctx.bias_ref, _bias_grad(dy, K))
        return dx, dw, db


class _PreluConv2d(Function):
    """y2 = conv2d(prelu(y1, alpha), w)  (bias-free; model_irse.py:59-60).  Forward is the plain two-kernel sequence;
    backward fuses the PReLU backward into the input-gradient epilogue (no d_prelu round trip through HBM)."""

    @staticmethod
    def forward(ctx, y1, alpha, w, stride, pad, p1=None):
        y1 = _c(y1)
        N, H, W, Cp = y1.shape
        K, C, R, S = w.shape
        assert C == Cp and C % 8 == 0
        al = _c(alpha.detach().float())
        if p1 is None:   # (else: prelu(y1) already came out of the producing convolution's epilogue)
            p1 = torch.empty_like(y1)
            lib.xr_affine_act(dt(y1), ptr(y1), None, None, None, ptr(al), ACT_PRELU, ptr(p1), 1, N * H * W, C, 0, stream())
        Ho, Wo = (H + 2 * pad - R) // stride + 1, (W + 2 * pad - S) // stride + 1
        pk, kg = _packed(w, "fwd", y1.dtype, K, 1, R * S, C, Cp, C * R * S, 0, 1, R * S)
        Kp = r8(K)
        y2 = torch.empty((N, Ho, Wo, Kp), dtype=y1.dtype, device=y1.device)
        if _cfg["direct64_prelu"] and direct64_ok(p1, w, stride, pad):
            _conv64(p1, pk, y2, tag=("fwd", Cp, K, H, W, R, stride))      # 64 -> 64 stride 1: the direct kernel
        else:
            pe = _probe_begin(("fwd", Cp, K, H, W, R, stride))
            lib.xr_conv_igemm(dtc(y1), ptr(p1), ptr(pk), None, ptr(y2), N, H, W, Cp, Ho, Wo, K, R, S, stride, pad, 0, kg, Kp, None, 0,
                              None, None, None, 1, None, None, None, stream())
            if pe is not None:
                pe.record()
        ctx.save_for_backward(y1, p1, w, al)
        ctx.geom = (stride, pad)
        ctx.alpha_ref = alpha
        return y2

    @staticmethod
    def backward(ctx, dy):
        y1, p1, w, al = ctx.saved_tensors
        stride, pad = ctx.geom
        alpha = ctx.alpha_ref
        dy = _c(dy)
        if dy.dtype != y1.dtype:
            dy = dy.to(y1.dtype)
        N, H, W, Cp = y1.shape
        K, C, R, S = w.shape
        _, Ho, Wo, Kp = dy.shape
        dev = y1.device
        dy1 = dalpha = dw = None
        if ctx.needs_input_grad[0] or ctx.needs_input_grad[1]:
            t_a = _direct(alpha)
            # dalpha partial sums are spread over `sp` rows (row tile i -> row i % sp) and folded afterwards: thousands of
            # tiles adding into the same C floats serialise at the memory side (measured: +60 us on a 100 us kernel)
            sp = _cfg["dalpha_spread"] if N * H * W >= 32 * 1024 else 1
            if _cfg.get("deterministic"):
                sp = _tile_rows(N * H * W, sp)
            if sp > 1:
                dal_s = zeros_f32((sp, C), dev)
                dal = t_a if t_a is not None else torch.empty((C,), dtype=torch.float32, device=dev)
            else:
                dal_s = dal = t_a if t_a is not None else zeros_f32((C,), dev)
            dy1 = torch.empty_like(y1)
            pe = _probe_begin(("dgrad", Cp, K, H, W, R, stride))
            if _dgrad_s2_ok(y1.shape, w, stride, pad, Ho, Wo) and K == Kp:
                # stage-opening unit: the strided input gradient as one dense 2x2-window GEMM with a depth-to-space epilogue; dalpha
                # partials come per (spread row, sub-pixel class) and are folded over both
                pk, kg = _packed_s2(w, y1.dtype)
                dal_s = zeros_f32((sp, 4 * C), dev)
                if t_a is None:
                    dal = torch.empty((C,), dtype=torch.float32, device=dev)
                else:
                    dal = t_a
                lib.xr_conv_dgrad_s2(dtc(y1), ptr(dy), ptr(pk), ptr(dy1), N, Ho, Wo, Kp, Cp, kg, ptr(y1), ptr(al.repeat(4)), ptr(dal_s), sp,
                                     None, None, stream())
                sp *= 4
            else:
                pk, kg = _packed(w, "dgrad", y1.dtype, C, 1, R * S, K, Kp, R * S, 0, 1, C * R * S)
                lib.xr_conv_igemm(dtc(y1), ptr(dy), ptr(pk), None, ptr(dy1), N, Ho, Wo, Kp, H, W, C, R, S, stride, pad, 1, kg, Cp, None,
                                  0, ptr(y1), ptr(al), ptr(dal_s), sp, None, None, None, stream())
            if pe is not None:
                pe.record()
            if sp > 1:
                lib.xr_reduce_groups(ptr(dal_s), ptr(dal), 1, sp, C, 1 if t_a is not None else 0, stream())
            if t_a is not None:
                _direct_done(alpha)
            else:
                dalpha = dal
        if ctx.needs_input_grad[2] and _wanted(w):
            kg = kg_of(R * S, Cp)
            dw = _wgrad(w, p1, dy, N, H, W, Cp, Ho, Wo, K, R, S, stride, pad, 0, Kp, kg, _wgrad_split(N * Ho * Wo, K, kg),
                        K, 1, R * S, C, Cp, C * R * S, 0, 1, R * S)
        return dy1, dalpha, dw, None, None, None


def prelu_conv2d(y1, alpha, w, stride=1, pad=0, p1=None):
    return _PreluConv2d.apply(y1, alpha, w, stride, pad, p1)


def conv2d_prelu(x, w, alpha, b=None, stride=1, pad=0, bn_link=None):
    """(y, prelu(y, alpha)) with the activation produced by the convolution's epilogue; pass both to prelu_conv2d.
    bn_link: the BnLink of the BatchNorm whose output `x` is (and which nothing else consumes)."""
    return _Conv2d.apply(x, w, b, stride, pad, alpha, bn_link)


class StatsLink:
    """Couples a convolution with the training-mode BatchNorm that directly follows it: the convolution's epilogue takes the
    per-channel sum / sum of squares of its output (ep_red with ep_src = NULL) and the norm skips its statistics pass."""

    SPREAD = 32

    def __init__(self):
        self.sums = None
        self.key = None
        self.pivot = None     # [fold][C]: the partial sums are relative to it (xr_affine_act_stats_pivot), or None

    def deliver(self, y, sums, pivot=None):
        self.sums, self.pivot, self.key = sums, pivot, (y.data_ptr(), y._version, tuple(y.shape))

    def take(self, x):
        """(sums, pivot) delivered for exactly this tensor, or None."""
        sums, pivot, key = self.sums, self.pivot, self.key
        self.sums = self.key = self.pivot = None
        if sums is None or key != (x.data_ptr(), x._version, tuple(x.shape)):
            return None
        return sums, pivot


def conv2d(x, w, b=None, stride=1, pad=0, stats_link=None):
    if stride > 1 and w.shape[2] == 1 and w.shape[3] == 1 and pad == 0 and _cfg["conv1x1_subsample"]:
        # 1x1 / stride s (the shortcut convolutions of the stage transitions, model_irse.py:54, model/resnet.py:196): sub-sample first,
        # then a plain stride-1 1x1 GEMM -- its input gradient is a dense GEMM + a zero-filling scatter and its weight gradient the
        # forward-gather fast path.  OFF by default (XR_CONV1X1_SUBSAMPLE=1): the strided forms look terrible in the in-step per-site
        # table (18-140 TFLOP/s, 0.45 ms per C2 step) but that time is CU sharing with the side stream -- the interleaved A/B of the
        # whole step showed no difference (19.43 vs 19.44 ms)
        x, stride = subsample(x, stride), 1
    return _Conv2d.apply(x, w, b, stride, pad, None, None, stats_link)


def conv2d_pass(x, w, b=None, stride=1, pad=0, stats_link=None):
    """(conv(x), x'): x' aliases x; route the identity / residual branch through x' and its gradient is added to the
    convolution's input gradient inside the dgrad epilogue instead of by a separate elementwise pass."""
    return _Conv2d.apply(x, w, b, stride, pad, None, None, stats_link, True)


def conv_transpose2d(x, w, b=None, stride=1, pad=0, out_pad=0):
    return _ConvTranspose2d.apply(x, w, b, stride, pad, out_pad)


def linear_nhwc(x, w, b=None):
    return _LinearNHWC.apply(x, w, b)


# ------------------------------------------------------------------------------------------------- norm + act
_ACT = {None: ACT_NONE, "none": ACT_NONE, "prelu": ACT_PRELU, "relu": ACT_RELU, "tanh": ACT_TANH}


class _NormAct(Function):
    """y = act(norm(x) + res).  mode: 'in' (InstanceNorm2d), 'bn' (BatchNorm, batch statistics when
    ``training``), 'none' (activation / residual add only)."""

    @staticmethod
    def forward(ctx, x, gamma, beta, rmean, rvar, res, alpha, mode, act, training, momentum, eps, passthrough=False, link=None,
                slink=None, eval_coef=None, tail=None, offer=None):
        x = _c(x)
        N, H, W, C = x.shape
        ctx.passthrough = passthrough
        ctx.link = link   # see BnLink: the consumer convolution's dgrad may deliver this norm's backward reductions
        ctx.tail = tail   # see TailLink: this norm's backward apply may take the backward sums of the unit that produced x
        if link is not None:
            link.x = x    # the exact buffer the statistics were taken over
        if res is not None:
            res = _c(res)
            assert res.shape == x.shape and res.dtype == x.dtype
        dev = x.device
        stats = mode == "in" or (mode == "bn" and training)
        G, rows = (N, H * W) if mode == "in" else (1, N * H * W)
        f32 = dict(dtype=torch.float32, device=dev)
        gm = None if gamma is None else _c(gamma.detach().float())
        bt = None if beta is None else _c(beta.detach().float())
        al = None if alpha is None else _c(alpha.detach().float())
        mean = invstd = scale = shift = None
        # BatchNorm sums are taken per image (few blocks per atomic address) and folded over the batch afterwards
        per_img = mode != "in" and N > 1 and H * W >= 16
        pg = 1
        if per_img:  # pseudo-groups: the largest divisor of N that is <= 32 (deterministic mode: one group per image)
            pg = N if _cfg.get("deterministic") else max(d for d in range(1, min(N, 32) + 1) if N % d == 0)
            per_img = pg > 1
        pre = slink.take(x) if (slink is not None and stats and mode == "bn") else None
        if pre is not None:   # the producer already summed x and x^2 per channel (StatsLink): a convolution epilogue, or the
            pre, piv = pre    # elementwise pass that wrote x (then relative to a pivot per partial)
            mean, invstd, scale, shift = torch.empty((4, G, C), **f32).unbind(0)
            upd = rmean is not None
            lib.xr_norm_finalize_pivot(ptr(pre), ptr(piv), ptr(gm), ptr(bt), ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                                       ptr(rmean) if upd else None, ptr(rvar) if upd else None, G, rows, C, eps, momentum,
                                       pre.shape[1], stream())
        elif stats:
            # statistics relative to a pivot (the group's first element): E[x^2] - mean^2 on raw fp32 sums cancels for
            # |mean| >> std (aten::batch_norm / instance_norm use a shifted / Welford form, model_irse.py:56-60)
            if per_img and _cfg["fold_finalize"]:
                sums = zeros_f32((2, pg, C), dev)      # folded inside xr_norm_finalize_pivot
                piv = torch.empty((pg, C), **f32)
                lib.xr_group_stats_pivot(dt(x), ptr(x), ptr(sums), ptr(piv), pg, (N // pg) * H * W, C, stream())
            else:
                sums = zeros_f32((2, G, C), dev)
                piv = torch.empty((G, C), **f32)
                lib.xr_group_stats_pivot(dt(x), ptr(x), ptr(sums), ptr(piv), G, rows, C, stream())
            mean, invstd, scale, shift = torch.empty((4, G, C), **f32).unbind(0)
            upd = mode == "bn" and rmean is not None
            lib.xr_norm_finalize_pivot(ptr(sums), ptr(piv), ptr(gm), ptr(bt), ptr(mean), ptr(invstd), ptr(scale), ptr(shift),
                                       ptr(rmean) if upd else None, ptr(rvar) if upd else None, G, rows, C, eps, momentum,
                                       pg if (per_img and _cfg["fold_finalize"]) else 1, stream())
        elif mode == "bn":
            if eval_coef is not None:   # frozen network: (scale, shift) cached by the layer (nn._BNMixin)
                scale, shift = eval_coef
            else:
                scale, shift = torch.empty((1, C), **f32), torch.empty((1, C), **f32)
                lib.xr_bn_eval_coeffs(ptr(gm), ptr(bt), ptr(rmean), ptr(rvar), ptr(scale), ptr(shift), C, eps, stream())
        y = torch.empty_like(x)
        a = _ACT[act]
        if offer is not None and mode == "bn" and N > 1:
            # the output opens a residual unit whose BatchNorm wants its statistics: one group per image, shared coefficients
            osum = zeros_f32((2, N, C), dev)
            opiv = torch.empty((N, C), **f32)
            lib.xr_affine_act_stats_pivot(dt(x), ptr(x), ptr(scale), ptr(shift), ptr(res), ptr(al), a, ptr(y), ptr(osum), ptr(opiv), N,
                                          H * W, C, 0, stream())
            offer.slink = StatsLink()
            offer.slink.deliver(y, osum, opiv)
        else:
            lib.xr_affine_act(dt(x), ptr(x), ptr(scale), ptr(shift), ptr(res), ptr(al), a, ptr(y), G, rows, C, 1, stream())
        ctx.save_for_backward(x, res, scale, shift, mean, invstd, gm, al, rmean, rvar)
        ctx.meta = (mode, a, stats, G, rows, C, eps, gamma is not None, beta is not None, alpha is not None)
        ctx.prefs = (gamma, beta, alpha)
        ctx.per_img = (per_img, pg, (N // pg) * H * W)
        if passthrough is True:  # second output aliases the input: its gradient is folded into dx by the apply kernel
            return y, x.view_as(x)
        if passthrough:          # stride s >= 2: second output = the sub-sampled input (MaxPool2d(1, s), model_irse.py:53); its
            s_ = int(passthrough)   # COMPACT gradient is added at the strided pixels by the apply kernel (xr_affine_act_bwd_apply_sub)
            xs = torch.empty((N, (H + s_ - 1) // s_, (W + s_ - 1) // s_, C), dtype=x.dtype, device=dev)
            lib.xr_subsample(dt(x), ptr(x), ptr(xs), N, H, W, C, s_, stream())
            return y, xs
        return y

    @staticmethod
    def backward(ctx, dy, dpass=None):
        x, res, scale, shift, mean, invstd, gm, al, rmean, rvar = ctx.saved_tensors
        mode, a, stats, G, rows, C, eps, has_g, has_b, has_a = ctx.meta
        if dy is None:
            dy = torch.zeros_like(x)
        dy = _c(dy)
        if dy.dtype != x.dtype:
            dy = dy.to(x.dtype)
        if dpass is not None:
            dpass = _c(dpass)
            if dpass.dtype != x.dtype:
                dpass = dpass.to(x.dtype)
        f32 = dict(dtype=torch.float32, device=x.device)
        per_img, N_, HW_ = ctx.per_img
        link_red = ctx.link.take(dy) if ctx.link is not None else None
        if link_red is not None:
            # the convolution that consumed this BatchNorm's output already reduced (dy, dy*x) in its dgrad epilogue
            red, per_img, N_ = link_red, True, link_red.shape[1]
            if not stats:   # eval-mode norm: the generic path below wants folded sums
                red = torch.empty((3, 1, C), **f32)
                lib.xr_reduce_groups(ptr(link_red), ptr(red), 3, N_, C, 0, stream())
        elif per_img:
            red_n = zeros_f32((3, N_, C), x.device)
            lib.xr_affine_act_bwd_reduce(dt(x), ptr(x), ptr(scale), ptr(shift), ptr(res), ptr(al), a, ptr(dy), ptr(red_n), N_, HW_,
                                         C, 0, stream())
            red = red_n
            if not stats:   # (with batch statistics the fold happens inside xr_norm_bwd_coeffs)
                red = torch.empty((3, 1, C), **f32)
                lib.xr_reduce_groups(ptr(red_n), ptr(red), 3, N_, C, 0, stream())
        else:
            red = zeros_f32((3, G, C), x.device)
            lib.xr_affine_act_bwd_reduce(dt(x), ptr(x), ptr(scale), ptr(shift), ptr(res), ptr(al), a, ptr(dy), ptr(red), G, rows,
                                         C, 1, stream())
        dgamma = dbeta = dalpha = None
        coef = None
        p_g, p_b, p_a = ctx.prefs
        if stats:
            coef = torch.empty((3, G, C), **f32)
            # xr_norm_bwd_coeffs ACCUMULATES into its dgamma/dbeta/dalpha pointers: aim it at .grad when direct
            t_g, t_b, t_a = (_direct(p_g) if has_g else None), (_direct(p_b) if has_b else None), (_direct(p_a) if has_a else None)
            dgamma = (t_g if t_g is not None else zeros_f32((C,), x.device)) if has_g else None
            dbeta = (t_b if t_b is not None else zeros_f32((C,), x.device)) if has_b else None
            dalpha = (t_a if t_a is not None else zeros_f32((C,), x.device)) if has_a else None
            lib.xr_norm_bwd_coeffs(ptr(red), ptr(gm), ptr(mean), ptr(invstd), ptr(coef), ptr(dgamma), ptr(dbeta), ptr(dalpha),
                                   G, rows, C, N_ if per_img else 1, stream())
            if t_g is not None:
                dgamma = None
                _direct_done(p_g)
            if t_b is not None:
                dbeta = None
                _direct_done(p_b)
            if t_a is not None:
                dalpha = None
                _direct_done(p_a)
        else:
            if has_a:
                t_a = _direct(p_a)
                if t_a is not None:
                    lib.xr_reduce_groups(ptr(red[2]), ptr(t_a), 1, G, C, 1, stream())
                    _direct_done(p_a)
                else:
                    dalpha = torch.empty(C, **f32)
                    lib.xr_reduce_groups(ptr(red[2]), ptr(dalpha), 1, G, C, 0, stream())
            if mode == "bn":
                if has_b:
                    dbeta = _emit_small(p_b, red[0, 0])
                if has_g:
                    dgamma = _emit_small(p_g, (red[1, 0] - rmean * red[0, 0]) * torch.rsqrt(rvar + eps))
        dx = torch.empty_like(x) if ctx.needs_input_grad[0] else None
        dres = torch.empty_like(x) if (res is not None and ctx.needs_input_grad[5]) else None
        tail = ctx.tail
        sub = ctx.passthrough if (ctx.passthrough is not True and ctx.passthrough) else 0
        chain = (tail is not None and dx is not None and dres is None and mode == "bn" and G == 1 and _cfg["chain_units"]
                 and tail.y is not None and tail.y.shape == x.shape and tail.y.dtype == x.dtype)
        if sub and dpass is not None and dx is not None:
            # the identity branch was sub-sampled: its compact gradient goes in at the strided pixels (no zero-filled full-size tensor)
            n_img, h_, w_ = x.shape[0], x.shape[1], x.shape[2]
            red2 = zeros_f32((2, n_img, C), x.device) if chain else None
            lib.xr_affine_act_bwd_apply_sub(dt(x), ptr(x), ptr(scale), ptr(shift), ptr(res), ptr(al), a, ptr(dy), ptr(coef), ptr(dx), n_img,
                                            h_, w_, C, ptr(dpass), int(sub), ptr(tail.y) if chain else None, ptr(red2), stream())
            if chain:
                tail.deliver(dx, red2)
            return dx, dgamma, dbeta, None, None, dres, dalpha, None, None, None, None, None, None, None, None, None, None, None
        if chain:
            # dx IS the gradient entering the tail of the unit before this norm: take that tail's per-image sums here
            n_img = x.shape[0]
            assert not sub or dpass is None
            red2 = zeros_f32((2, n_img, C), x.device)
            lib.xr_affine_act_bwd_apply_red(dt(x), ptr(x), ptr(scale), ptr(shift), ptr(res), ptr(al), a, ptr(dy), ptr(coef), ptr(dx),
                                            None, n_img, rows // n_img, C, ptr(dpass), ptr(tail.y), ptr(red2), stream())
            tail.deliver(dx, red2)
        elif dx is not None or dres is not None:
            lib.xr_affine_act_bwd_apply(dt(x), ptr(x), ptr(scale), ptr(shift), ptr(res), ptr(al), a, ptr(dy), ptr(coef), ptr(dx),
                                        ptr(dres), G, rows, C, 1, ptr(dpass), stream())
        return dx, dgamma, dbeta, None, None, dres, dalpha, None, None, None, None, None, None, None, None, None, None, None


class BnLink:
    """Couples a training-mode BatchNorm (no activation, no residual) with the ONE convolution that consumes its output: the
    convolution's input-gradient kernel then also accumulates sum(d) and sum(d * x) per channel in its epilogue (ep_red), and
    the BatchNorm's backward skips its own reduction pass over (d, x).  The norm's backward only trusts a delivery made for
    exactly the gradient tensor it receives."""

    SPREAD = 32

    def __init__(self):
        self.x = None         # the BatchNorm's input buffer (same layout as its output), set by the norm's forward
        self.red = None
        self.key = None

    def deliver(self, dgrad_out, red):
        self.red, self.key = red, (dgrad_out.data_ptr(), dgrad_out._version)

    def take(self, dy):
        red, key = self.red, self.key
        self.red = self.key = None
        if red is None or dy is None or key != (dy.data_ptr(), dy._version):
            return None
        return red


class TailLink:
    """Couples the fused tail of one IR-SE unit (out = SE(BN(y)) + shortcut, _BnSeAdd) with the BatchNorm that opens the next
    unit.  Forward: the tail's elementwise pass also sums out / out^2 (``slink``), so the next norm skips its statistics pass.
    Backward: the next norm's apply pass writes dx = the gradient entering the tail, and sums (dx, dx * y) per image while it does
    (xr_affine_act_bwd_apply_red), so the tail skips its reduction pass.  Both sides only trust a delivery keyed to exactly the
    tensor they receive: an extra consumer of the unit output (a feature tap) makes autograd sum gradients into a new tensor
    and the tail falls back to its own pass."""

    def __init__(self):
        self.y = None         # the tail's input buffer (set by its forward)
        self.slink = None     # StatsLink carrying the statistics of the unit output
        self.red = None
        self.key = None

    def deliver(self, dx, red):
        self.red, self.key = red, (dx.data_ptr(), dx._version, tuple(dx.shape))

    def take(self, dout):
        red, key = self.red, self.key
        self.red = self.key = None
        if red is None or dout is None or key != (dout.data_ptr(), dout._version, tuple(dout.shape)):
            return None
        return red


def chain_of(x):
    """The TailLink a unit tail attached to its output (None when x is anything else)."""
    return getattr(x, "_xr_tail", None) if _cfg["chain_units"] else None


def norm_act_pass(x, gamma=None, beta=None, rmean=None, rvar=None, mode="bn", act=None, training=True, momentum=0.1, eps=EPS,
                  link=None, tail=None, sub=1):
    """(norm(x), x'): x' aliases x; route identity branches (block shortcuts) through x' and the two gradients of x are
    summed inside the norm's backward apply kernel instead of by a separate elementwise pass.  sub = s >= 2: x' is x sub-sampled
    by s (the MaxPool2d(1, s) shortcut / the input of a 1x1 stride-s shortcut convolution) and comes back as a compact gradient."""
    slink = tail.slink if tail is not None else None
    pt = True if sub <= 1 else int(sub)
    return _NormAct.apply(x, gamma, beta, rmean, rvar, None, None, mode, act, training, momentum, eps, pt, link, slink, None,
                          tail if training else None)


def norm_act(x, gamma=None, beta=None, rmean=None, rvar=None, res=None, alpha=None, mode="none", act=None, training=True,
             momentum=0.1, eps=EPS, slink=None, eval_coef=None, offer_stats=False):
    """offer_stats: the output feeds a residual unit that opens with a training-mode BatchNorm (the IR input layer): its
    statistics are taken in this pass and travel with the output (TailLink without a backward half)."""
    offer = TailLink() if (offer_stats and mode == "bn" and training and _cfg["chain_units"]) else None
    y = _NormAct.apply(x, gamma, beta, rmean, rvar, res, alpha, mode, act, training, momentum, eps, False, None, slink, eval_coef,
                       None, offer)
    if offer is not None and offer.slink is not None:
        y._xr_tail = offer
    return y


def bn_eval_coeffs(gamma, beta, rmean, rvar, eps):
    """(scale, shift) [1][C] fp32 of an eval-mode BatchNorm."""
    C = rmean.numel()
    f32 = dict(dtype=torch.float32, device=rmean.device)
    gm = None if gamma is None else _c(gamma.detach().float())
    bt = None if beta is None else _c(beta.detach().float())
    scale, shift = torch.empty((1, C), **f32), torch.empty((1, C), **f32)
    lib.xr_bn_eval_coeffs(ptr(gm), ptr(bt), ptr(rmean), ptr(rvar), ptr(scale), ptr(shift), C, eps, stream())
    return scale, shift


# ------------------------------------------------------------------------------------------------- direct 64-channel path
def direct64_ok(x, w, stride, pad):
    """bf16, 64 -> 64 channels, 3x3, stride 1, pad 1: the weights-stationary direct kernel (csrc/xr_conv64.hip) applies."""
    return (_cfg["direct64"] and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[3] == 64 and tuple(w.shape) == (64, 64, 3, 3)
            and stride == 1 and pad == 1 and x.numel() * 2 < (1 << 31))


def _conv64(x, pk, out, transposed=0, bias=None, scale=None, shift=None, alpha=None, stats=None, add=None, tag=None):
    N, H, W, _ = x.shape
    pe = _probe_begin(tag) if tag is not None else None
    lib.xr_conv64_direct(ptr(x), ptr(pk), ptr(bias), ptr(out), N, H, W, transposed, ptr(scale), ptr(shift), ptr(alpha), ptr(stats),
                         ptr(add), stream())
    if pe is not None:
        pe.record()
    return out


def _resblock_fields():
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    ptrs1 = ("w1_fwd", "w2_fwd", "w1_dgrad", "w2_dgrad", "g1", "b1", "a1", "g2", "b2", "ao", "x", "c1", "c2", "out", "mean1", "invstd1",
             "scale1", "shift1", "mean2", "invstd2", "scale2", "shift2", "ws_fwd", "dout", "tail_red", "dx", "dc2", "dy1", "dc1", "dres",
             "ws_bwd", "dg1", "db1", "da1", "dg2", "db2", "dao", "slabs", "dw1", "dw2")
    ptrs2 = ("prev_c2", "prev_x", "prev_scale2", "prev_shift2", "prev_ao", "prev_tail_red", "side_stream", "fork_event")
    return ([("N", I), ("H", I), ("W", I), ("eps", F)] + [(k, P) for k in ptrs1] + [("dw_accumulate", I), ("reserved", I)]
            + [(k, P) for k in ptrs2])


class _ResblockDesc(ctypes.Structure):
    """include/xrface.h: xr_resblock_desc (field for field; the size is checked against the library once)."""
    _fields_ = _resblock_fields()


_desc_checked = [False]


def _resblock_desc(**kw):
    if not _desc_checked[0]:
        if lib.xr_resblock_desc_size() != ctypes.sizeof(_ResblockDesc):
            raise RuntimeError("xrface: xr_resblock_desc layout mismatch between include/xrface.h and xrface.ops")
        _desc_checked[0] = True
    d = _ResblockDesc()
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return d


def _resblock_abi_ok(x):
    """The block-level entry points (xr_resblock_fwd / xr_resblock_bwd) cover what the direct weight-gradient kernel covers."""
    return _cfg["block_abi"] and x.shape[2] % 8 == 0 and x.shape[2] <= 112 and not _cfg.get("probe")


def _side_for_block(dev):
    """(side stream handle, fork event handle) for xr_resblock_bwd, or (None, None) when the weight gradients stay on the main
    stream; the C side records the event and makes the side stream wait, _side_done() does the bookkeeping afterwards."""
    if not (_cfg["wgrad_stream"] and _cfg["wgrad64_stream"]) or _one_stream():
        return None, None, None
    if _side["stream"] is None or _side["dev"] != dev:
        _side_fork(dev)            # creates stream + event (and records the event once: a HIP event exists after its first record)
    return _side["stream"], _side["stream"].cuda_stream, _side["ev"].cuda_event


class _ResBlock64(Function):
    """FSRNet residual block (model/FSRnet.py:75-98) on 64 channels in bf16 as ONE op over the direct convolution kernel:
        c1 = conv1(x)                      + per-image sum / sum-of-squares of c1 in the epilogue      (no statistics pass)
        c2 = conv2(prelu(IN1(c1)))         with IN1 + PReLU applied on load                            (y1 never exists in HBM)
                                           + statistics of c2 in the epilogue
        out = prelu_out(IN2(c2) + x)       one elementwise pass
    7 tensor passes instead of 11; backward: both input gradients on the direct kernel (the residual-branch gradient is summed
    in conv1's dgrad epilogue), y1 is recomputed once for conv2's weight gradient."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, a1, w2, g2, b2, ao, eps):
        x = _c(x)
        N, H, W, C = x.shape
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        pk1, _ = _packed(w1, "fwd", x.dtype, 64, 1, 9, 64, 64, 576, 0, 1, 9)
        pk2, _ = _packed(w2, "fwd", x.dtype, 64, 1, 9, 64, 64, 576, 0, 1, 9)
        fl = lambda t: None if t is None else _c(t.detach().float())
        g1f, b1f, a1f, g2f, b2f, aof = fl(g1), fl(b1), fl(a1), fl(g2), fl(b2), fl(ao)
        HW = H * W
        tag = ("fwd", 64, 64, H, W, 3, 1)

        def finalize(sums, gm, bt):
            mean, invstd = torch.empty((N, C), **f32), torch.empty((N, C), **f32)
            scale, shift = torch.empty((N, C), **f32), torch.empty((N, C), **f32)
            lib.xr_norm_finalize(ptr(sums), ptr(gm), ptr(bt), ptr(mean), ptr(invstd), ptr(scale), ptr(shift), None, None, N, HW, C, eps,
                                 0.0, 1, stream())
            return mean, invstd, scale, shift

        st1 = zeros_f32((2, N, C), dev)
        c1 = _conv64(x, pk1, torch.empty_like(x), stats=st1, tag=tag)
        mean1, invstd1, scale1, shift1 = finalize(st1, g1f, b1f)
        st2 = zeros_f32((2, N, C), dev)
        c2 = _conv64(c1, pk2, torch.empty_like(x), scale=scale1, shift=shift1, alpha=a1f, stats=st2, tag=tag)
        mean2, invstd2, scale2, shift2 = finalize(st2, g2f, b2f)
        out = torch.empty_like(x)
        lib.xr_affine_act(dt(x), ptr(c2), ptr(scale2), ptr(shift2), ptr(x), ptr(aof), ACT_PRELU, ptr(out), N, HW, C, 1, stream())
        ctx.save_for_backward(x, c1, c2, w1, w2, mean1, invstd1, scale1, shift1, mean2, invstd2, scale2, shift2, g1f, a1f, g2f, aof)
        ctx.prefs = (g1, b1, a1, g2, b2, ao)
        return out

    @staticmethod
    def backward(ctx, dout):
        x, c1, c2, w1, w2, mean1, invstd1, scale1, shift1, mean2, invstd2, scale2, shift2, g1f, a1f, g2f, aof = ctx.saved_tensors
        g1, b1, a1, g2, b2, ao = ctx.prefs
        N, H, W, C = x.shape
        HW = H * W
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        dout = _c(dout)
        if dout.dtype != x.dtype:
            dout = dout.to(x.dtype)
        d = dt(x)

        def small(p_, has):
            t_ = _direct(p_) if has else None
            return t_, ((t_ if t_ is not None else zeros_f32((C,), dev)) if has else None)

        def norm_bwd(xin, scale, shift, res, alf, gmf, mean, invstd, dy, p_g, p_b, p_a, want_res, red=None):
            if red is None:     # (else: the three sums came out of the epilogue of the convolution that produced dy)
                red = zeros_f32((3, N, C), dev)
                lib.xr_affine_act_bwd_reduce(d, ptr(xin), ptr(scale), ptr(shift), ptr(res), ptr(alf), ACT_PRELU, ptr(dy), ptr(red), N, HW,
                                             C, 1, stream())
            coef = torch.empty((3, N, C), **f32)
            (t_g, dg), (t_b, db), (t_a, da) = small(p_g, p_g is not None), small(p_b, p_b is not None), small(p_a, True)
            lib.xr_norm_bwd_coeffs(ptr(red), ptr(gmf), ptr(mean), ptr(invstd), ptr(coef), ptr(dg), ptr(db), ptr(da), N, HW, C, 1, stream())
            outs = []
            for p_, t_, v in ((p_g, t_g, dg), (p_b, t_b, db), (p_a, t_a, da)):
                if t_ is not None:
                    _direct_done(p_)
                    outs.append(None)
                else:
                    outs.append(v)
            dx = torch.empty_like(xin)
            dres = torch.empty_like(xin) if want_res else None
            lib.xr_affine_act_bwd_apply(d, ptr(xin), ptr(scale), ptr(shift), ptr(res), ptr(alf), ACT_PRELU, ptr(dy), ptr(coef), ptr(dx),
                                        ptr(dres), N, HW, C, 1, None, stream())
            return dx, dres, outs

        need_x = ctx.needs_input_grad[0]
        # out = prelu_out(IN2(c2) + x)
        dc2, dres, (dg2, db2, dao) = norm_bwd(c2, scale2, shift2, x, aof, g2f, mean2, invstd2, dout, g2, b2, ao, need_x)
        # c2 = conv2(y1), y1 = prelu(IN1(c1))
        tagd = ("dgrad", 64, 64, H, W, 3, 1)
        pkd2, _ = _packed(w2, "dgrad", x.dtype, 64, 1, 9, 64, 64, 9, 0, 1, 576)
        red1 = None
        if _cfg["fuse_in_reduce"]:
            # conv2's input gradient with the reductions of IN1 + PReLU's backward taken in its epilogue (it reads c1 beside the
            # output it streams out): xr_affine_act_bwd_reduce over (c1, dy1) -- two full-tensor reads -- disappears
            red1 = zeros_f32((3, N, C), dev)
            dy1 = torch.empty_like(x)
            pe = _probe_begin(tagd)
            lib.xr_conv64_direct_bwdred(ptr(dc2), ptr(pkd2), ptr(dy1), N, H, W, 1, ptr(c1), ptr(scale1), ptr(shift1), ptr(a1f), ptr(red1),
                                        stream())
            if pe is not None:
                pe.record()
        else:
            dy1 = _conv64(dc2, pkd2, torch.empty_like(x), transposed=1, tag=tagd)
        dc1, _, (dg1, db1, da1) = norm_bwd(c1, scale1, shift1, None, a1f, g1f, mean1, invstd1, dy1, g1, b1, a1, False, red=red1)
        dx = None
        if need_x:
            pkd1, _ = _packed(w1, "dgrad", x.dtype, 64, 1, 9, 64, 64, 9, 0, 1, 576)
            dx = _conv64(dc1, pkd1, torch.empty_like(x), transposed=1, add=dres, tag=tagd)   # + the residual-branch gradient
        # Both weight gradients go to the side stream after conv1's input gradient is on the main stream.  The persistent direct
        # kernels of the two streams cannot share a CU (120 KB + 87 KB of LDS): wherever the weight gradients are launched they
        # displace main-stream convolutions for as long as they run -- launching conv2's between the two input gradients or here
        # measured the same step time (C3 38.5 ms either way), so the simpler order is kept
        dw2 = None
        if ctx.needs_input_grad[5] and _wanted(w2):
            if _wgrad64_ok(c1, dc2, H, W, 64, H, W, 64, 3, 3, 1, 1, 0):
                # y1 = prelu(IN1(c1)) is rebuilt ON LOAD by the direct weight-gradient kernel (the forward never wrote it either)
                dw2 = _wgrad(w2, c1, dc2, N, H, W, 64, H, W, 64, 3, 3, 1, 1, 0, 64, 576, 0, 64, 1, 9, 64, 64, 576, 0, 1, 9,
                             xform=(scale1, shift1, a1f))
            else:
                y1 = torch.empty_like(x)   # recomputed for the weight gradient only
                lib.xr_affine_act(d, ptr(c1), ptr(scale1), ptr(shift1), None, ptr(a1f), ACT_PRELU, ptr(y1), N, HW, C, 1, stream())
                dw2 = _wgrad(w2, y1, dc2, N, H, W, 64, H, W, 64, 3, 3, 1, 1, 0, 64, 576, _wgrad_split(N * HW, 64, 576), 64, 1, 9, 64,
                             64, 576, 0, 1, 9)
        dw1 = None
        if ctx.needs_input_grad[1] and _wanted(w1):
            dw1 = _wgrad(w1, x, dc1, N, H, W, 64, H, W, 64, 3, 3, 1, 1, 0, 64, 576, _wgrad_split(N * HW, 64, 576), 64, 1, 9, 64, 64, 576,
                         0, 1, 9)
        return dx, dw1, dg1, db1, da1, dw2, dg2, db2, dao, None


def resblock64(x, conv1, in1, relu, conv2, in2, relu_out):
    """conv1 / conv2: xrface.nn.Conv2d (64 -> 64, 3x3, bias-free), in1 / in2: affine InstanceNorm2d, relu / relu_out: PReLU(64)."""
    if _cfg["res_trunk"] and _resblock_abi_ok(x) and x.dtype == torch.bfloat16 and x.numel() * 2 < (1 << 31):
        # a single application of the chained-trunk op: forward / backward are one xr_resblock_fwd / xr_resblock_bwd call each
        return _ResTrunk64.apply(x, 1, 1, in1.eps, conv1.weight, in1.weight, in1.bias, relu.weight, conv2.weight, in2.weight,
                                 in2.bias, relu_out.weight)
    return _ResBlock64.apply(x, conv1.weight, in1.weight, in1.bias, relu.weight, conv2.weight, in2.weight, in2.bias,
                             relu_out.weight, in1.eps)


class _ResTrunk64(Function):
    """``times`` x [block_0 .. block_{nb-1}] of FSRNet residual blocks (model/FSRnet.py:331-333: the same three blocks applied
    three times) on 64 channels in bf16 as ONE autograd op.  Forward = the _ResBlock64 forward per application.  Backward chains
    consecutive applications: conv1's input-gradient kernel of application i + 1 does not store the gradient dout_i of the
    previous application's output but dz_i = dout_i * prelu'(tail_i) -- the gradient of that tail's pre-activation -- plus the three
    InstanceNorm-backward sums over it (xr_conv64_direct_tailred).  dz_i is at once application i's residual-branch gradient and
    the input of IN2's backward, so application i runs no reduce pass for its tail and a two-input apply pass (dz, c2 -> dc2)
    instead of the three-input, two-output one: 18 tensor passes per application instead of 21.  Only the last application
    (whose dout comes from autograd) takes the standard path.  Block parameters: (w1, g1, b1, a1, w2, g2, b2, ao) per block."""

    @staticmethod
    def forward(ctx, x, times, nb, eps, *params):
        x = _c(x)
        N, H, W, C = x.shape
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        HW = H * W
        tag = ("fwd", 64, 64, H, W, 3, 1)
        fl = lambda t: _c(t.detach().float())
        blocks = [params[8 * b:8 * b + 8] for b in range(nb)]
        pk = [(_packed(p_[0], "fwd", x.dtype, 64, 1, 9, 64, 64, 576, 0, 1, 9)[0], _packed(p_[4], "fwd", x.dtype, 64, 1, 9, 64, 64, 576, 0, 1, 9)[0])
              for p_ in blocks]
        small = [[fl(t) for t in (p_[1], p_[2], p_[3], p_[5], p_[6], p_[7])] for p_ in blocks]     # g1 b1 a1 g2 b2 ao

        def finalize(sums, gm, bt):
            mean, invstd = torch.empty((N, C), **f32), torch.empty((N, C), **f32)
            scale, shift = torch.empty((N, C), **f32), torch.empty((N, C), **f32)
            lib.xr_norm_finalize(ptr(sums), ptr(gm), ptr(bt), ptr(mean), ptr(invstd), ptr(scale), ptr(shift), None, None, N, HW, C, eps,
                                 0.0, 1, stream())
            return mean, invstd, scale, shift

        saved = [x]
        cur = x
        abi = _resblock_abi_ok(x)
        for i in range(times * nb):
            b = i % nb
            g1f, b1f, a1f, g2f, b2f, aof = small[b]
            if abi:   # one C call enqueues the five launches of the application (xr_resblock_fwd)
                stat = torch.empty((8, N, C), **f32)
                c1, c2, out = torch.empty_like(cur), torch.empty_like(cur), torch.empty_like(cur)
                d_ = _resblock_desc(N=N, H=H, W=W, eps=eps, w1_fwd=pk[b][0], w2_fwd=pk[b][1], g1=g1f, b1=b1f, a1=a1f, g2=g2f, b2=b2f,
                                    ao=aof, x=cur, c1=c1, c2=c2, out=out, mean1=stat[0], invstd1=stat[1], scale1=stat[2],
                                    shift1=stat[3], mean2=stat[4], invstd2=stat[5], scale2=stat[6], shift2=stat[7],
                                    ws_fwd=zeros_f32((4, N, C), dev))
                lib.xr_resblock_fwd(ctypes.addressof(d_), stream())
                saved += [c1, c2, *stat.unbind(0), out]
                cur = out
                continue
            st1 = zeros_f32((2, N, C), dev)
            c1 = _conv64(cur, pk[b][0], torch.empty_like(cur), stats=st1, tag=tag)
            mean1, invstd1, scale1, shift1 = finalize(st1, g1f, b1f)
            st2 = zeros_f32((2, N, C), dev)
            c2 = _conv64(c1, pk[b][1], torch.empty_like(cur), scale=scale1, shift=shift1, alpha=a1f, stats=st2, tag=tag)
            mean2, invstd2, scale2, shift2 = finalize(st2, g2f, b2f)
            out = torch.empty_like(cur)
            lib.xr_affine_act(dt(cur), ptr(c2), ptr(scale2), ptr(shift2), ptr(cur), ptr(aof), ACT_PRELU, ptr(out), N, HW, C, 1, stream())
            saved += [c1, c2, mean1, invstd1, scale1, shift1, mean2, invstd2, scale2, shift2, out]
            cur = out
        for sm in small:
            saved += sm
        ctx.save_for_backward(*saved)
        ctx.prefs = blocks
        ctx.shape = (times, nb)
        return cur

    @staticmethod
    def backward(ctx, dout):
        sv = ctx.saved_tensors
        times, nb = ctx.shape
        A = times * nb
        blocks = ctx.prefs
        x0 = sv[0]
        per = [sv[1 + 11 * i: 1 + 11 * (i + 1)] for i in range(A)]     # c1 c2 mean1 invstd1 scale1 shift1 mean2 invstd2 scale2 shift2 out
        small = [sv[1 + 11 * A + 6 * b: 1 + 11 * A + 6 * (b + 1)] for b in range(nb)]
        N, H, W, C = x0.shape
        HW = H * W
        dev = x0.device
        f32 = dict(dtype=torch.float32, device=dev)
        d = dt(x0)
        g = _c(dout)
        if g.dtype != x0.dtype:
            g = g.to(x0.dtype)
        tagd = ("dgrad", 64, 64, H, W, 3, 1)
        need_x = ctx.needs_input_grad[0]
        # small-gradient accumulators, one per parameter, shared by the applications of a block (the coefficient kernel adds)
        acc = {}

        def small_acc(p_):
            if id(p_) not in acc:
                t_ = _direct(p_)
                acc[id(p_)] = (t_, t_ if t_ is not None else zeros_f32((C,), dev))
            return acc[id(p_)]

        dws = {}

        def add_dw(p_, v):
            if v is not None:
                dws[id(p_)] = v if id(p_) not in dws else dws[id(p_)] + v

        red_tail = None     # sums of the CURRENT application's tail, when they came out of the previous kernel's epilogue
        abi = _resblock_abi_ok(x0)
        for i in range(A - 1, -1, -1):
            b = i % nb
            w1, g1, b1, a1, w2, g2, b2, ao = blocks[b]
            g1f, b1f, a1f, g2f, b2f, aof = small[b]
            c1, c2, mean1, invstd1, scale1, shift1, mean2, invstd2, scale2, shift2, _ = per[i]
            xin = x0 if i == 0 else per[i - 1][10]
            if abi:
                # ---- one C call for the whole application (xr_resblock_bwd): tail, both input gradients with their fused sums, both
                # weight gradients (forked onto the side stream inside the call)
                (t_g2, dg2), (t_b2, db2), (t_ao, dao) = small_acc(g2), small_acc(b2), small_acc(ao)
                (t_g1, dg1), (t_b1, db1), (t_a1, da1) = small_acc(g1), small_acc(b1), small_acc(a1)
                want1, want2 = _wanted(w1), _wanted(w2)
                tg1, tg2 = (_direct(w1) if want1 else None), (_direct(w2) if want2 else None)
                accum = tg1 is not None or tg2 is not None     # one flag for both targets: a fresh target then starts from zero
                fresh = torch.zeros_like if accum else torch.empty_like
                dw1 = (tg1 if tg1 is not None else fresh(w1, dtype=torch.float32)) if want1 else None
                dw2 = (tg2 if tg2 is not None else fresh(w2, dtype=torch.float32)) if want2 else None
                on_side = (dw1 is None or tg1 is not None) and (dw2 is None or tg2 is not None) and (want1 or want2)
                side, side_h, ev_h = _side_for_block(dev) if on_side else (None, None, None)
                split = min(256, N * H)
                slabs = torch.empty((2 * split * 64 * 576,), **f32) if (want1 or want2) else None
                dc2, dy1, dc1 = torch.empty_like(x0), torch.empty_like(x0), torch.empty_like(x0)
                dres = torch.empty_like(x0) if red_tail is None else None
                chain = i > 0
                nxt = torch.empty_like(x0) if (chain or need_x) else None
                nred = zeros_f32((3, N, C), dev) if chain else None
                pi = per[i - 1] if chain else None
                d_ = _resblock_desc(
                    N=N, H=H, W=W, eps=0.0,
                    w1_dgrad=_packed(w1, "dgrad", x0.dtype, 64, 1, 9, 64, 64, 9, 0, 1, 576)[0] if nxt is not None else None,
                    w2_dgrad=_packed(w2, "dgrad", x0.dtype, 64, 1, 9, 64, 64, 9, 0, 1, 576)[0],
                    g1=g1f, b1=b1f, a1=a1f, g2=g2f, b2=b2f, ao=aof, x=xin, c1=c1, c2=c2, mean1=mean1, invstd1=invstd1, scale1=scale1,
                    shift1=shift1, mean2=mean2, invstd2=invstd2, scale2=scale2, shift2=shift2, dout=g, tail_red=red_tail, dx=nxt,
                    dc2=dc2, dy1=dy1, dc1=dc1, dres=dres, ws_bwd=zeros_f32((12, N, C), dev), dg1=dg1, db1=db1, da1=da1, dg2=dg2,
                    db2=db2, dao=dao, slabs=slabs, dw1=dw1, dw2=dw2, dw_accumulate=int(accum),
                    prev_c2=pi[1] if chain else None, prev_x=(x0 if i == 1 else per[i - 2][10]) if chain else None,
                    prev_scale2=pi[8] if chain else None, prev_shift2=pi[9] if chain else None,
                    prev_ao=small[(i - 1) % nb][5] if chain else None, prev_tail_red=nred, side_stream=side_h, fork_event=ev_h)
                lib.xr_resblock_bwd(ctypes.addressof(d_), stream())
                if side is not None:
                    _side_done(side, (xin, c1, dc2, dc1, slabs, scale1, shift1, a1f))
                for p_ in (g2, b2, ao, g1, b1, a1):
                    if acc[id(p_)][0] is not None:
                        _direct_done(p_)
                for p_, t_, v in ((w2, tg2, dw2), (w1, tg1, dw1)):
                    if t_ is not None:
                        _direct_done(p_)
                    elif v is not None:
                        add_dw(p_, v)
                g, red_tail = nxt, nred
                continue
            # ---- tail: out = prelu_out(IN2(c2) + xin)
            (t_g, dg), (t_b, db), (t_a, da) = small_acc(g2), small_acc(b2), small_acc(ao)
            coef = torch.empty((3, N, C), **f32)
            dc2 = torch.empty_like(x0)
            if red_tail is None:
                red = zeros_f32((3, N, C), dev)
                lib.xr_affine_act_bwd_reduce(d, ptr(c2), ptr(scale2), ptr(shift2), ptr(xin), ptr(aof), ACT_PRELU, ptr(g), ptr(red), N, HW,
                                             C, 1, stream())
                lib.xr_norm_bwd_coeffs(ptr(red), ptr(g2f), ptr(mean2), ptr(invstd2), ptr(coef), ptr(dg), ptr(db), ptr(da), N, HW, C, 1,
                                       stream())
                dz = torch.empty_like(x0)
                lib.xr_affine_act_bwd_apply(d, ptr(c2), ptr(scale2), ptr(shift2), ptr(xin), ptr(aof), ACT_PRELU, ptr(g), ptr(coef), ptr(dc2),
                                            ptr(dz), N, HW, C, 1, None, stream())
            else:
                # g already IS dz (the epilogue of the next application's conv1 input gradient applied prelu' and took the sums)
                lib.xr_norm_bwd_coeffs(ptr(red_tail), ptr(g2f), ptr(mean2), ptr(invstd2), ptr(coef), ptr(dg), ptr(db), ptr(da), N, HW, C, 1,
                                       stream())
                dz = g
                lib.xr_affine_act_bwd_apply(d, ptr(c2), None, None, None, None, ACT_NONE, ptr(dz), ptr(coef), ptr(dc2), None, N, HW, C, 1,
                                            None, stream())
            for p_ in (g2, b2, ao):
                if acc[id(p_)][0] is not None:
                    _direct_done(p_)
            # ---- conv2's input gradient + the sums of IN1 / PReLU's backward
            pkd2, _ = _packed(w2, "dgrad", x0.dtype, 64, 1, 9, 64, 64, 9, 0, 1, 576)
            red1 = zeros_f32((3, N, C), dev)
            dy1 = torch.empty_like(x0)
            pe = _probe_begin(tagd)
            lib.xr_conv64_direct_bwdred(ptr(dc2), ptr(pkd2), ptr(dy1), N, H, W, 1, ptr(c1), ptr(scale1), ptr(shift1), ptr(a1f), ptr(red1),
                                        stream())
            if pe is not None:
                pe.record()
            (t_g1, dg1), (t_b1, db1), (t_a1, da1) = small_acc(g1), small_acc(b1), small_acc(a1)
            coef1 = torch.empty((3, N, C), **f32)
            lib.xr_norm_bwd_coeffs(ptr(red1), ptr(g1f), ptr(mean1), ptr(invstd1), ptr(coef1), ptr(dg1), ptr(db1), ptr(da1), N, HW, C, 1,
                                   stream())
            for p_ in (g1, b1, a1):
                if acc[id(p_)][0] is not None:
                    _direct_done(p_)
            dc1 = torch.empty_like(x0)
            lib.xr_affine_act_bwd_apply(d, ptr(c1), ptr(scale1), ptr(shift1), None, ptr(a1f), ACT_PRELU, ptr(dy1), ptr(coef1), ptr(dc1), None,
                                        N, HW, C, 1, None, stream())
            # ---- conv1's input gradient (+ the residual-branch gradient dz); chained into the previous application's tail
            pkd1, _ = _packed(w1, "dgrad", x0.dtype, 64, 1, 9, 64, 64, 9, 0, 1, 576)
            if i > 0:
                pc2, pscale2, pshift2 = per[i - 1][1], per[i - 1][8], per[i - 1][9]
                pxin = x0 if i == 1 else per[i - 2][10]
                paof = small[(i - 1) % nb][5]
                red_tail = zeros_f32((3, N, C), dev)
                nxt = torch.empty_like(x0)
                pe = _probe_begin(tagd)
                lib.xr_conv64_direct_tailred(ptr(dc1), ptr(pkd1), ptr(nxt), N, H, W, 1, ptr(dz), ptr(pc2), ptr(pxin), ptr(pscale2),
                                             ptr(pshift2), ptr(paof), ptr(red_tail), stream())
                if pe is not None:
                    pe.record()
                g = nxt
            elif need_x:
                g = _conv64(dc1, pkd1, torch.empty_like(x0), transposed=1, add=dz, tag=tagd)
            else:
                g = None
            # ---- weight gradients (side stream when accumulating in place)
            if _wanted(w2):
                add_dw(w2, _wgrad(w2, c1, dc2, N, H, W, 64, H, W, 64, 3, 3, 1, 1, 0, 64, 576, 0, 64, 1, 9, 64, 64, 576, 0, 1, 9,
                                  xform=(scale1, shift1, a1f)))
            if _wanted(w1):
                add_dw(w1, _wgrad(w1, xin, dc1, N, H, W, 64, H, W, 64, 3, 3, 1, 1, 0, 64, 576, 0, 64, 1, 9, 64, 64, 576, 0, 1, 9))
        grads = []
        for b in range(nb):
            w1, g1, b1, a1, w2, g2, b2, ao = blocks[b]
            sm = lambda p_: None if (id(p_) not in acc or acc[id(p_)][0] is not None) else acc[id(p_)][1]
            grads += [dws.get(id(w1)), sm(g1), sm(b1), sm(a1), dws.get(id(w2)), sm(g2), sm(b2), sm(ao)]
        return (g, None, None, None, *grads)


def res_trunk64_ok(x, blocks):
    """bf16 64-channel NHWC input with W % 8 == 0 and W <= 112 and every block a 64 -> 64 FSRNet residual block: the chained trunk op."""
    if not (_cfg["direct64"] and _cfg["res_trunk"] and x.dtype == torch.bfloat16 and x.dim() == 4 and x.shape[3] == 64
            and x.shape[2] % 8 == 0 and x.shape[2] <= 112 and x.numel() * 2 < (1 << 31)):
        return False
    return all(tuple(b.conv1.weight.shape) == (64, 64, 3, 3) and tuple(b.conv2.weight.shape) == (64, 64, 3, 3) for b in blocks)


def res_trunk64(x, blocks, times):
    """``times`` passes over ``blocks`` (xrface.model.FSRnet._Residual_Block modules) as one chained op."""
    params = []
    for b in blocks:
        params += [b.conv1.weight, b.in1.weight, b.in1.bias, b.relu.weight, b.conv2.weight, b.in2.weight, b.in2.bias, b.relu_out.weight]
    return _ResTrunk64.apply(x, times, len(blocks), blocks[0].in1.eps, *params)


# ------------------------------------------------------------------------------------------------- SE
class _SEScaleAdd(Function):
    """out = r * sigmoid(fc2(relu(fc1(avgpool(r))))) + shortcut   (SEModule + residual add,
    model_irse.py:38-46,88-91).  w1: [Cr][C][1][1], w2: [C][Cr][1][1]."""

    @staticmethod
    def forward(ctx, r, w1, w2, shortcut):
        r = _c(r)
        N, H, W, C = r.shape
        Cr = w1.shape[0]
        f32 = dict(dtype=torch.float32, device=r.device)
        sums = zeros_f32((2, N, C), r.device)
        lib.xr_group_stats(dt(r), ptr(r), ptr(sums), N, H * W, C, stream())
        w1f, w2f = _c(w1.detach().float()), _c(w2.detach().float())
        hidden, s = torch.empty((N, Cr), **f32), torch.empty((N, C), **f32)
        inv_hw = 1.0 / (H * W)
        lib.xr_se_excite_fwd(ptr(sums[0]), ptr(w1f), ptr(w2f), ptr(hidden), ptr(s), N, C, Cr, inv_hw, stream())
        sc = None if shortcut is None else _c(shortcut)
        y = torch.empty_like(r)
        lib.xr_affine_act(dt(r), ptr(r), ptr(s), None, ptr(sc), None, ACT_NONE, ptr(y), N, H * W, C, 1, stream())
        ctx.save_for_backward(r, w1f, w2f, hidden, s, sums)
        ctx.inv_hw = inv_hw
        ctx.has_sc = shortcut is not None
        ctx.wrefs = (w1, w2)
        return y

    @staticmethod
    def backward(ctx, dy):
        r, w1f, w2f, hidden, s, sums = ctx.saved_tensors
        dy = _c(dy)
        if dy.dtype != r.dtype:
            dy = dy.to(r.dtype)
        N, H, W, C = r.shape
        Cr = hidden.shape[1]
        f32 = dict(dtype=torch.float32, device=r.device)
        red = zeros_f32((3, N, C), r.device)
        # act none, no shift/res: dz = dy, red[1] = sum dy * r = ds
        lib.xr_affine_act_bwd_reduce(dt(r), ptr(r), None, None, None, None, ACT_NONE, ptr(dy), ptr(red), N, H * W, C, 1, stream())
        dpre2, dhid = torch.empty((N, C), **f32), torch.empty((N, Cr), **f32)
        coef = zeros_f32((3, N, C), r.device)
        coef[0].copy_(s)
        lib.xr_se_excite_bwd(ptr(w1f), ptr(w2f), ptr(hidden), ptr(s), ptr(red[1]), ptr(dpre2), ptr(dhid), ptr(coef[2]), N, C, Cr,
                             ctx.inv_hw, stream())
        dr = torch.empty_like(r)
        lib.xr_affine_act_bwd_apply(dt(r), ptr(r), None, None, None, None, ACT_NONE, ptr(dy), ptr(coef), ptr(dr), None, N, H * W, C,
                                    1, None, stream())
        w1, w2 = ctx.wrefs
        t1, t2 = _direct(w1), _direct(w2)
        dw1 = t1 if t1 is not None else zeros_f32((Cr, C, 1, 1), r.device)
        dw2 = t2 if t2 is not None else zeros_f32((C, Cr, 1, 1), r.device)
        lib.xr_small_atb(ptr(dhid), ptr(sums[0]), ptr(dw1), N, Cr, C, ctx.inv_hw, 1, stream())
        lib.xr_small_atb(ptr(dpre2), ptr(hidden), ptr(dw2), N, C, Cr, 1.0, 1, stream())
        if t1 is not None:
            dw1 = None
            _direct_done(w1)
        if t2 is not None:
            dw2 = None
            _direct_done(w2)
        return dr, dw1, dw2, (dy if ctx.has_sc else None)


def se_scale_add(r, w1, w2, shortcut=None):
    return _SEScaleAdd.apply(r, w1, w2, shortcut)


class _BnSeAdd(Function):
    """out = SE(BatchNorm(y)) + shortcut with BatchNorm(y) never written to HBM (tail of bottleneck_IR_SE,
    model_irse.py:76-91): one per-image statistics pass over y serves both the batch statistics and the SE squeeze,
    then a single elementwise pass out = y*(a*s) + (b*s) + shortcut.  Backward is one reduction pass + one apply pass."""

    @staticmethod
    def forward(ctx, y, gamma, beta, rmean, rvar, w1, w2, shortcut, training, momentum, eps, tail=None):
        y = _c(y)
        N, H, W, C = y.shape
        HW, Cr = H * W, w1.shape[0]
        ctx.tail = tail
        dev = y.device
        f32 = dict(dtype=torch.float32, device=dev)
        sums_n = zeros_f32((2, N, C), dev)
        lib.xr_group_stats(dt(y), ptr(y), ptr(sums_n), N, HW, C, stream())
        gm, bt = _c(gamma.detach().float()), _c(beta.detach().float())
        mean, invstd, a, b = torch.empty((4, 1, C), **f32).unbind(0)
        if training:
            if _cfg["fold_finalize"]:   # per-image sums folded over the batch in the same launch
                lib.xr_norm_finalize(ptr(sums_n), ptr(gm), ptr(bt), ptr(mean), ptr(invstd), ptr(a), ptr(b), ptr(rmean), ptr(rvar),
                                     1, N * HW, C, eps, momentum, N, stream())
            else:
                sums = torch.empty((2, 1, C), **f32)
                lib.xr_reduce_groups(ptr(sums_n), ptr(sums), 2, N, C, 0, stream())
                lib.xr_norm_finalize(ptr(sums), ptr(gm), ptr(bt), ptr(mean), ptr(invstd), ptr(a), ptr(b), ptr(rmean), ptr(rvar), 1,
                                     N * HW, C, eps, momentum, 1, stream())
        else:
            lib.xr_bn_eval_coeffs(ptr(gm), ptr(bt), ptr(rmean), ptr(rvar), ptr(a), ptr(b), C, eps, stream())
            mean = rmean.detach().float().reshape(1, C)
            invstd = torch.rsqrt(rvar.detach().float() + eps).reshape(1, C)
        w1f, w2f = _c(w1.detach().float()), _c(w2.detach().float())
        pooled, s, cA, cB = torch.empty((4, N, C), **f32).unbind(0)
        hidden = torch.empty((N, Cr), **f32)
        lib.xr_bnse_fwd(ptr(sums_n[0]), ptr(a), ptr(b), ptr(w1f), ptr(w2f), ptr(pooled), ptr(hidden), ptr(s), ptr(cA), ptr(cB), N, C,
                        Cr, HW, stream())
        sc = None if shortcut is None else _c(shortcut)
        out = torch.empty_like(y)
        if tail is not None:   # the statistics of `out` ride along for the BatchNorm that opens the next unit
            tail.y = y
            osum = zeros_f32((2, N, C), dev)
            opiv = torch.empty((N, C), **f32)
            lib.xr_affine_act_stats_pivot(dt(y), ptr(y), ptr(cA), ptr(cB), ptr(sc), None, ACT_NONE, ptr(out), ptr(osum), ptr(opiv), N, HW,
                                          C, 1, stream())
            tail.slink = StatsLink()
            tail.slink.deliver(out, osum, opiv)
        else:
            lib.xr_affine_act(dt(y), ptr(y), ptr(cA), ptr(cB), ptr(sc), None, ACT_NONE, ptr(out), N, HW, C, 1, stream())
        ctx.save_for_backward(y, sums_n, a, b, mean, invstd, gm, w1f, w2f, pooled, hidden, s)
        ctx.meta = (training, shortcut is not None)
        ctx.prefs = (gamma, beta, w1, w2)
        return out

    @staticmethod
    def backward(ctx, dout):
        y, sums_n, a, b, mean, invstd, gm, w1f, w2f, pooled, hidden, s = ctx.saved_tensors
        training, has_sc = ctx.meta
        p_g, p_b, w1, w2 = ctx.prefs
        dout = _c(dout)
        if dout.dtype != y.dtype:
            dout = dout.to(y.dtype)
        N, H, W, C = y.shape
        HW, Cr = H * W, hidden.shape[1]
        dev = y.device
        f32 = dict(dtype=torch.float32, device=dev)
        red = ctx.tail.take(dout) if ctx.tail is not None else None   # delivered by the next unit's BatchNorm backward
        if red is None:
            red = zeros_f32((3, N, C), dev)  # red[0] = S1 = sum dout, red[1] = S2 = sum dout*y   (per image)
            lib.xr_affine_act_bwd_reduce(dt(y), ptr(y), None, None, None, None, ACT_NONE, ptr(dout), ptr(red), N, HW, C, 1, stream())
        dhid = torch.empty((N, Cr), **f32)
        big = torch.empty((5, N, C), **f32)      # dpre2, dp, coef[3]: one allocation
        dpre2, dp, coef = big[0], big[1], big[2:]
        t_g, t_b = _direct(p_g), _direct(p_b)
        need_g, need_b = ctx.needs_input_grad[1], ctx.needs_input_grad[2]
        dgamma = (t_g if t_g is not None else zeros_f32((C,), dev)) if need_g else None
        dbeta = (t_b if t_b is not None else zeros_f32((C,), dev)) if need_b else None
        lib.xr_bnse_bwd(ptr(red[0]), ptr(red[1]), ptr(sums_n[0]), ptr(a), ptr(b), ptr(w1f), ptr(w2f), ptr(hidden), ptr(s), ptr(gm),
                        ptr(mean), ptr(invstd), ptr(dpre2), ptr(dhid), ptr(dp), ptr(coef), ptr(dgamma), ptr(dbeta), N, C, Cr, HW,
                        int(training), stream())
        if t_g is not None and need_g:
            dgamma = None
            _direct_done(p_g)
        if t_b is not None and need_b:
            dbeta = None
            _direct_done(p_b)
        dy = None
        if ctx.needs_input_grad[0]:
            dy = torch.empty_like(y)
            lib.xr_affine_act_bwd_apply(dt(y), ptr(y), None, None, None, None, ACT_NONE, ptr(dout), ptr(coef), ptr(dy), None, N, HW,
                                        C, 1, None, stream())
        dw1 = dw2 = None
        if ctx.needs_input_grad[5] and _wanted(w1):
            t1, t2 = _direct(w1), _direct(w2)
            dw1 = t1 if t1 is not None else zeros_f32((Cr, C, 1, 1), dev)
            dw2 = t2 if t2 is not None else zeros_f32((C, Cr, 1, 1), dev)

            on_side = t1 is not None and t2 is not None and _cfg["wgrad_stream"] and not _one_stream()
            side = _side_fork(dev) if on_side else None      # parameter gradients: off the critical path
            sh = side.cuda_stream if on_side else stream()
            lib.xr_small_atb(ptr(dhid), ptr(pooled), ptr(dw1), N, Cr, C, 1.0 / HW, 1, sh)
            lib.xr_small_atb(ptr(dpre2), ptr(hidden), ptr(dw2), N, C, Cr, 1.0, 1, sh)
            if on_side:
                _side_done(side, (dhid, pooled, dpre2, hidden))
            if t1 is not None:
                dw1 = None
                _direct_done(w1)
            if t2 is not None:
                dw2 = None
                _direct_done(w2)
        return dy, dgamma, dbeta, None, None, dw1, dw2, (dout if has_sc else None), None, None, None, None


def bn_se_add(y, bn, se, shortcut):
    """bn: an xrface.nn.BatchNorm2d holder, se: a model_irse.SEModule holder."""
    training = bn.training or not bn.track_running_stats
    bn._count()
    mom = bn._momentum()
    # a training unit offers its output to the next unit's opening BatchNorm through a TailLink (picked up by chain_of)
    tail = TailLink() if (training and _cfg["chain_units"] and torch.is_grad_enabled() and y.requires_grad) else None
    out = _BnSeAdd.apply(y, bn.weight, bn.bias, bn.running_mean, bn.running_var, se.fc1.weight, se.fc2.weight, shortcut, training,
                         mom, bn.eps, tail)
    if tail is not None:
        out._xr_tail = tail
    return out


# ------------------------------------------------------------------------------------------------- IR-SE unit, block level
def _ir_block_fields():
    P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
    ints = ("wgrad_rows", "fold_in", "pg", "ep_spread", "da_spread", "split", "dw_accumulate", "dalpha_accumulate", "reserved")
    ptrs = ("g1", "b1", "rmean1", "rvar1", "w1_fwd", "w2_fwd", "w1_dgrad", "w2_dgrad", "alpha", "g2", "b2", "rmean2", "rvar2", "se1", "se2",
            "x", "bn1", "y1", "p1", "y2", "out", "stats_in", "pivot_in", "stats_own", "pivot_own", "mean1", "invstd1", "scale1", "shift1",
            "sums2", "mean2", "invstd2", "a2", "b2c", "pooled", "hidden", "s", "cA", "cB", "stats_out", "pivot_out",
            "dout", "tail_red", "red_tail", "dpre2", "dhid", "dp", "coef2", "dg2", "db2", "dy2", "dy1", "db1t", "dx", "dal_s", "dalpha",
            "red1", "coef1", "dg1", "db1g", "prev_y2", "prev_red2", "slabs1", "slabs2", "dw1", "dw2", "dse1", "dse2", "side_stream",
            "fork_event")
    return ([(k, I) for k in ("N", "H", "W", "C", "Cr")] + [("eps", F), ("momentum", F)] + [(k, I) for k in ints] + [(k, P) for k in ptrs])


class _IrBlockDesc(ctypes.Structure):
    """include/xrface.h: xr_ir_block_desc (field for field; the size is checked against the library once)."""
    _fields_ = _ir_block_fields()


_ir_desc_checked = [False]


def _ir_block_desc(**kw):
    if not _ir_desc_checked[0]:
        if lib.xr_ir_block_desc_size() != ctypes.sizeof(_IrBlockDesc):
            raise RuntimeError("xrface: xr_ir_block_desc layout mismatch between include/xrface.h and xrface.ops")
        _ir_desc_checked[0] = True
    d = _IrBlockDesc()
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if isinstance(v, torch.Tensor) else v)
    return d


class _IrSeUnit(Function):
    """bottleneck_IR_SE with identity shortcut (model_irse.py:69-91) as ONE autograd node over the block-level entry points
    xr_ir_block_fwd / xr_ir_block_bwd: the same launches, in the same order, as the op-level composition (BN1 with BnLink, conv ->
    PReLU -> conv with the activation out of the epilogue, BatchNorm + SE + shortcut tail with TailLink), from one C call each way.
    ``links``: (tail_prev, tail) -- the TailLink the previous unit attached to x (or None) and the one this unit offers."""

    @staticmethod
    def forward(ctx, x, g1, b1, w1, alpha, w2, g2, b2, se1, se2, bn1, bn2, links):
        x = _c(x)
        N, H, W, C = x.shape
        Cr = se1.shape[0]
        HW = H * W
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        tail_prev, tail = links
        pk1, _ = _packed(w1, "fwd", x.dtype, C, 1, 9, C, C, C * 9, 0, 1, 9)
        pk2, _ = _packed(w2, "fwd", x.dtype, C, 1, 9, C, C, C * 9, 0, 1, 9)
        fl = lambda t: _c(t.detach().float())
        g1f, b1f, alf, g2f, b2f, se1f, se2f = fl(g1), fl(b1), fl(alpha), fl(g2), fl(b2), fl(se1), fl(se2)
        pre = tail_prev.slink.take(x) if (tail_prev is not None and tail_prev.slink is not None) else None
        pg = 0
        sown = pown = None
        if pre is None:     # no delivered statistics: BN1 takes its own over pseudo-groups, like _NormAct
            pg = N if _cfg.get("deterministic") else max(d for d in range(1, min(N, 32) + 1) if N % d == 0)
            sown, pown = zeros_f32((2, pg, C), dev), torch.empty((pg, C), **f32)
        c8 = torch.empty((8, C), **f32)          # mean1 invstd1 scale1 shift1 mean2 invstd2 a2 b2c
        nc4 = torch.empty((4, N, C), **f32)      # pooled s cA cB
        hidden = torch.empty((N, Cr), **f32)
        sums2 = zeros_f32((2, N, C), dev)
        bn1t, y1, p1, y2, out = (torch.empty_like(x) for _ in range(5))
        osum = opiv = None
        if tail is not None:
            osum, opiv = zeros_f32((2, N, C), dev), torch.empty((N, C), **f32)
        d = _ir_block_desc(N=N, H=H, W=W, C=C, Cr=Cr, eps=bn1.eps, momentum=bn1.momentum, fold_in=0 if pre is None else pre[0].shape[1], pg=pg,
                           g1=g1f, b1=b1f, rmean1=bn1.running_mean, rvar1=bn1.running_var, w1_fwd=pk1, w2_fwd=pk2, alpha=alf, g2=g2f, b2=b2f,
                           rmean2=bn2.running_mean, rvar2=bn2.running_var, se1=se1f, se2=se2f, x=x, bn1=bn1t, y1=y1, p1=p1, y2=y2, out=out,
                           stats_in=None if pre is None else pre[0], pivot_in=None if pre is None else pre[1], stats_own=sown,
                           pivot_own=pown, mean1=c8[0], invstd1=c8[1], scale1=c8[2], shift1=c8[3], sums2=sums2, mean2=c8[4],
                           invstd2=c8[5], a2=c8[6], b2c=c8[7], pooled=nc4[0], hidden=hidden, s=nc4[1], cA=nc4[2], cB=nc4[3],
                           stats_out=osum, pivot_out=opiv)
        lib.xr_ir_block_fwd(ctypes.addressof(d), stream())
        if tail is not None:
            tail.y = y2
            tail.slink = StatsLink()
            tail.slink.deliver(out, osum, opiv)
        ctx.save_for_backward(x, bn1t, y1, p1, y2, c8, nc4, hidden, sums2, g1f, alf, g2f, se1f, se2f, w1, w2)
        ctx.prefs = (g1, b1, w1, alpha, w2, g2, b2, se1, se2)
        ctx.links = links
        return out

    @staticmethod
    def backward(ctx, dout):
        x, bn1t, y1, p1, y2, c8, nc4, hidden, sums2, g1f, alf, g2f, se1f, se2f, w1, w2 = ctx.saved_tensors
        g1, b1, _w1, alpha, _w2, g2, b2, se1, se2 = ctx.prefs
        tail_prev, tail = ctx.links
        N, H, W, C = x.shape
        Cr = hidden.shape[1]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        dout = _c(dout)
        if dout.dtype != x.dtype:
            dout = dout.to(x.dtype)
        red_in = tail.take(dout) if tail is not None else None          # [2][N][C] from the next unit's BN1 backward
        want = _wanted(w1)

        def small(p_):
            t_ = _direct(p_) if want else None
            return t_, (t_ if t_ is not None else zeros_f32((C,), dev))
        (t_g1, dg1), (t_b1, db1), (t_al, dal), (t_g2, dg2), (t_b2, db2) = small(g1), small(b1), small(alpha), small(g2), small(b2)
        tw1, tw2 = (_direct(w1) if want else None), (_direct(w2) if want else None)
        ts1, ts2 = (_direct(se1) if want else None), (_direct(se2) if want else None)
        accum = tw1 is not None or tw2 is not None
        fresh = torch.zeros_like if accum else torch.empty_like
        dw1 = (tw1 if tw1 is not None else fresh(w1, dtype=torch.float32)) if want else None
        dw2 = (tw2 if tw2 is not None else fresh(w2, dtype=torch.float32)) if want else None
        dse1 = (ts1 if ts1 is not None else zeros_f32(tuple(se1.shape), dev)) if want else None
        dse2 = (ts2 if ts2 is not None else zeros_f32(tuple(se2.shape), dev)) if want else None
        all_direct = want and all(t is not None for t in (tw1, tw2, ts1, ts2))
        side, side_h, ev_h = (None, None, None)
        if all_direct and _cfg["wgrad_stream"] and not _one_stream():
            if _side["stream"] is None or _side["dev"] != dev:
                _side_fork(dev)
            side, side_h, ev_h = _side["stream"], _side["stream"].cuda_stream, _side["ev"].cuda_event
        kg = 9 * C
        rows = bool(_cfg["wgrad_rows"] == 1 or (_cfg["wgrad_rows"] == 2 and C <= 128)) and 14 <= W <= 112
        split = max(1, 256 // ((C // 64) * (C // 64))) if rows else _wgrad_split(N * H * W, C, kg)
        slabs = torch.empty((2, split, C, kg), **f32) if want else None
        da_sp = _cfg["dalpha_spread"] if N * H * W >= 32 * 1024 else 1
        need_x = ctx.needs_input_grad[0]
        chain = (tail_prev is not None and need_x and _cfg["chain_units"] and tail_prev.y is not None and tail_prev.y.shape == x.shape
                 and tail_prev.y.dtype == x.dtype)
        dy2, dy1, db1t = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        dx = torch.empty_like(x) if need_x else None
        red2 = zeros_f32((2, N, C), dev) if chain else None
        big = torch.empty((5, N, C), **f32)          # dpre2 dp coef2[3]
        dhid = torch.empty((N, Cr), **f32)           # (read by the SE weight gradients on the side stream: must outlive this call)
        d = _ir_block_desc(N=N, H=H, W=W, C=C, Cr=Cr, eps=0.0, momentum=0.0, wgrad_rows=int(rows), ep_spread=BnLink.SPREAD, da_spread=da_sp,
                           split=split, dw_accumulate=int(accum), dalpha_accumulate=int(t_al is not None),
                           g1=g1f, w1_dgrad=_packed(w1, "dgrad", x.dtype, C, 1, 9, C, C, 9, 0, 1, C * 9)[0],
                           w2_dgrad=_packed(w2, "dgrad", x.dtype, C, 1, 9, C, C, 9, 0, 1, C * 9)[0], alpha=alf, g2=g2f,
                           b1=g1f, b2=g2f,     # (the betas are not read by the backward pass: any non-null pointer)
                           se1=se1f, se2=se2f, x=x, bn1=bn1t, y1=y1, p1=p1, y2=y2, mean1=c8[0], invstd1=c8[1], scale1=c8[2], shift1=c8[3],
                           sums2=sums2, mean2=c8[4], invstd2=c8[5], a2=c8[6], b2c=c8[7], pooled=nc4[0], hidden=hidden, s=nc4[1], cA=nc4[2],
                           cB=nc4[3], dout=dout, tail_red=red_in, red_tail=None if red_in is not None else zeros_f32((3, N, C), dev),
                           dpre2=big[0], dhid=dhid, dp=big[1], coef2=big[2:], dg2=dg2, db2=db2, dy2=dy2, dy1=dy1,
                           db1t=db1t, dx=dx, dal_s=zeros_f32((da_sp, C), dev), dalpha=dal, red1=zeros_f32((3, BnLink.SPREAD, C), dev),
                           coef1=torch.empty((3, 1, C), **f32), dg1=dg1, db1g=db1, prev_y2=tail_prev.y if chain else None, prev_red2=red2,
                           slabs1=slabs[0] if want else None, slabs2=slabs[1] if want else None, dw1=dw1, dw2=dw2, dse1=dse1, dse2=dse2,
                           side_stream=side_h, fork_event=ev_h)
        lib.xr_ir_block_bwd(ctypes.addressof(d), stream())
        if side is not None:
            _side_done(side, (x, bn1t, p1, dy2, dy1, slabs, nc4, hidden, big, dhid, c8, alf))
        if chain:
            tail_prev.deliver(dx, red2)
        outs = []
        for p_, t_, v in ((g1, t_g1, dg1), (b1, t_b1, db1), (w1, tw1, dw1), (alpha, t_al, dal), (w2, tw2, dw2), (g2, t_g2, dg2),
                          (b2, t_b2, db2), (se1, ts1, dse1), (se2, ts2, dse2)):
            if t_ is not None:
                _direct_done(p_)
                outs.append(None)
            else:
                outs.append(v if want else None)
        return (dx, *outs, None, None, None)


def ir_se_unit_ok(x, unit):
    """The block-level path covers the training-mode identity-shortcut bottleneck_IR_SE units on bf16 tensors with C >= 128."""
    if not (_cfg["ir_block"] and x.dtype == torch.bfloat16 and x.dim() == 4 and not _cfg.get("probe") and not _cfg.get("deterministic")
            and torch.is_grad_enabled()):
        return False
    rl = unit.res_layer
    C = x.shape[3]
    if not (unit.training and C >= 128 and C % 64 == 0 and tuple(rl[1].weight.shape) == (C, C, 3, 3) and tuple(rl[3].weight.shape) == (C, C, 3, 3)
            and rl[3].stride[0] == 1 and rl[0].track_running_stats and rl[4].track_running_stats and rl[0].momentum is not None
            and rl[4].momentum is not None and x.numel() * 2 < (1 << 31)):
        return False
    if rl[0].momentum != rl[4].momentum or rl[0].eps != rl[4].eps:
        return False
    return all(p.requires_grad for p in unit.parameters()) and x.shape[2] >= 7


def ir_se_unit(x, unit):
    """bottleneck_IR_SE.f for the units ir_se_unit_ok accepts (identity shortcut: MaxPool2d(1, 1))."""
    rl = unit.res_layer
    rl[0]._count()
    rl[4]._count()
    tail_prev = chain_of(x)
    tail = TailLink() if (_cfg["chain_units"] and x.requires_grad) else None
    out = _IrSeUnit.apply(x, rl[0].weight, rl[0].bias, rl[1].weight, rl[2].weight, rl[3].weight, rl[4].weight, rl[4].bias, rl[5].fc1.weight,
                          rl[5].fc2.weight, rl[0], rl[4], (tail_prev, tail))
    if tail is not None:
        out._xr_tail = tail
    return out


# ------------------------------------------------------------------------------------------------- resampling
class _Subsample(Function):
    @staticmethod
    def forward(ctx, x, stride):
        x = _c(x)
        N, H, W, C = x.shape
        Ho, Wo = (H + stride - 1) // stride, (W + stride - 1) // stride
        y = torch.empty((N, Ho, Wo, C), dtype=x.dtype, device=x.device)
        lib.xr_subsample(dt(x), ptr(x), ptr(y), N, H, W, C, stride, stream())
        ctx.meta = (N, H, W, C, stride)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C, stride = ctx.meta
        dy = _c(dy)
        dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
        lib.xr_subsample_bwd(dt(dy), ptr(dy), ptr(dx), N, H, W, C, stride, stream())
        return dx, None


class _MaxPool2(Function):
    @staticmethod
    def forward(ctx, x):
        x = _c(x)
        N, H, W, C = x.shape
        y = torch.empty((N, H // 2, W // 2, C), dtype=x.dtype, device=x.device)
        lib.xr_maxpool2(dt(x), ptr(x), ptr(y), N, H, W, C, stream())
        ctx.save_for_backward(x)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        N, H, W, C = x.shape
        dx = torch.empty_like(x)
        lib.xr_maxpool2_bwd(dt(x), ptr(x), ptr(dy), ptr(dx), N, H, W, C, stream())
        return dx


class _UpAdd2(Function):
    """up1 + nearest_upsample_x2(low) (model/FSRnet.py:210-211)."""

    @staticmethod
    def forward(ctx, up1, low):
        up1, low = _c(up1), _c(low)
        N, H, W, C = up1.shape
        assert low.shape == (N, H // 2, W // 2, C)
        y = torch.empty_like(up1)
        lib.xr_upadd2(dt(up1), ptr(up1), ptr(low), ptr(y), N, H, W, C, stream())
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = _c(dy)
        N, H, W, C = dy.shape
        dlow = torch.empty((N, H // 2, W // 2, C), dtype=dy.dtype, device=dy.device)
        lib.xr_upadd2_bwd(dt(dy), ptr(dy), ptr(dlow), N, H, W, C, stream())
        return dy, dlow


class _Cat2(Function):
    """torch.cat((a, b), channel) on NHWC buffers (model/FSRnet.py:505,534)."""

    @staticmethod
    def forward(ctx, a, b):
        a, b = _c(a), _c(b)
        N, H, W, Ca = a.shape
        Cb = b.shape[3]
        y = torch.empty((N, H, W, Ca + Cb), dtype=a.dtype, device=a.device)
        M = N * H * W
        lib.xr_copy_channels(dt(a), ptr(a), Ca, 0, ptr(y), Ca + Cb, 0, M, Ca, stream())
        lib.xr_copy_channels(dt(a), ptr(b), Cb, 0, ptr(y), Ca + Cb, Ca, M, Cb, stream())
        ctx.split = (Ca, Cb)
        return y

    @staticmethod
    def backward(ctx, dy):
        Ca, Cb = ctx.split
        dy = _c(dy)
        N, H, W, _ = dy.shape
        M = N * H * W
        da = torch.empty((N, H, W, Ca), dtype=dy.dtype, device=dy.device)
        db = torch.empty((N, H, W, Cb), dtype=dy.dtype, device=dy.device)
        lib.xr_copy_channels(dt(dy), ptr(dy), Ca + Cb, 0, ptr(da), Ca, 0, M, Ca, stream())
        lib.xr_copy_channels(dt(dy), ptr(dy), Ca + Cb, Ca, ptr(db), Cb, 0, M, Cb, stream())
        return da, db


class _Dropout(Function):
    @staticmethod
    def forward(ctx, x, p, mask, seed):
        x = _c(x)
        y = torch.empty_like(x)
        tick = _graph["tick"] if _graph["capturing"] else None
        lib.xr_dropout(dt(x), ptr(x), ptr(mask), ptr(y), x.numel(), p, seed, ptr(tick), stream())
        ctx.meta = (p, seed, tick)
        ctx.save_for_backward(mask)
        return y

    @staticmethod
    def backward(ctx, dy):
        p, seed, tick = ctx.meta
        (mask,) = ctx.saved_tensors
        dy = _c(dy)
        dx = torch.empty_like(dy)
        lib.xr_dropout(dt(dy), ptr(dy), ptr(mask), ptr(dx), dy.numel(), p, seed, ptr(tick), stream())
        return dx, None, None, None


class _AddSub(Function):
    @staticmethod
    def forward(ctx, a, b, sign):
        a, b = _c(a), _c(b)
        assert a.shape == b.shape and a.dtype == b.dtype
        y = torch.empty_like(a)
        (lib.xr_add if sign > 0 else lib.xr_sub)(dt(a), ptr(a), ptr(b), ptr(y), a.numel(), stream())
        ctx.sign = sign
        return y

    @staticmethod
    def backward(ctx, dy):
        return dy, (dy if ctx.sign > 0 else -dy), None


class _ReflectPad(Function):
    """nn.ReflectionPad2d(p) on an NHWC buffer (SUPER_RESOLUTION/model/FSRnet.py:255-292)."""

    @staticmethod
    def forward(ctx, x, p):
        x = _c(x)
        N, H, W, C = x.shape
        y = torch.empty((N, H + 2 * p, W + 2 * p, C), dtype=x.dtype, device=x.device)
        lib.xr_reflect_pad(dt(x), ptr(x), ptr(y), N, H, W, C, p, stream())
        ctx.meta = (N, H, W, C, p)
        return y

    @staticmethod
    def backward(ctx, dy):
        N, H, W, C, p = ctx.meta
        dy = _c(dy)
        dx = torch.empty((N, H, W, C), dtype=dy.dtype, device=dy.device)
        lib.xr_reflect_pad_bwd(dt(dy), ptr(dy), ptr(dx), N, H, W, C, p, stream())
        return dx, None


def reflect_pad(x, p):
    return _ReflectPad.apply(x, int(p))


def subsample(x, stride):
    return x if stride == 1 else _Subsample.apply(x, stride)


def maxpool2(x):
    return _MaxPool2.apply(x)


def upadd2(up1, low):
    return _UpAdd2.apply(up1, low)


def cat2(a, b):
    return _Cat2.apply(a, b)


_drop_counter = [0]


def dropout(x, p, training, mask=None, seed=None):
    """mask: optional uint8 keep-mask laid out like ``x`` (test injection); otherwise a counter-based stream."""
    if not training or p == 0.0:
        return x
    if seed is None:
        _drop_counter[0] += 1
        seed = (torch.initial_seed() * 0x9E3779B97F4A7C15 + _drop_counter[0]) & 0xFFFFFFFFFFFFFFFF
    if mask is not None:
        mask = _c(mask.to(torch.uint8))
    return _Dropout.apply(x, float(p), mask, int(seed))


class _GradScale(Function):
    """Identity in forward; the gradient is multiplied by ``s`` on the way back (steps.fhn_step_fused: one backward pass
    serves loss_k / theta_k pairs whose losses differ only by a constant factor)."""

    @staticmethod
    def forward(ctx, x, s):
        ctx.s = s
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return g * ctx.s, None


def grad_scale(x, s):
    return _GradScale.apply(x, float(s))


def sub(a, b):
    return _AddSub.apply(a, b, -1)


def add(a, b):
    return _AddSub.apply(a, b, 1)


def sub_detached(a, b):
    """(a - b).detach() for two same-layout activation tensors (the residual targets t_k - s_k of the KD step,
    distill_main.py:68-70), one launch, no autograd node."""
    with torch.no_grad():
        a2, b2 = _same_layout(a.detach(), b.detach())
        y = torch.empty_like(a2)
        lib.xr_sub(dt(a2), ptr(a2), ptr(b2), ptr(y), a2.numel(), stream())
    return y


# ------------------------------------------------------------------------------------------------- losses
def _same_layout(a, b):
    """Two same-shape tensors as flat contiguous memory in a common element order."""
    if a.dtype != b.dtype:
        a, b = a.float(), b.float()
    if a.dtype not in (torch.float32, torch.bfloat16):
        a, b = a.float(), b.float()
    if a.stride() == b.stride() and (a.is_contiguous() or (a.dim() == 4 and a.is_contiguous(memory_format=torch.channels_last))):
        return a, b
    return a.contiguous(), b.contiguous()


class _MSE(Function):
    """scale * mean((a-b)^2): MSELossFunc (scale 97) loss/loss.py:14 and nn.MSELoss (scale 1)."""

    @staticmethod
    def forward(ctx, a, b, scale):
        _need_cuda(a)
        assert a.shape == b.shape, f"mse: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}"
        a2, b2 = _same_layout(a.detach(), b.detach())
        loss = zeros_f32((1,), a.device).view(())
        n = a2.numel()
        lib.xr_loss_mse(dt(a2), ptr(a2), ptr(b2), scale, 1.0, ptr(loss), None, None, n, n, None, stream())
        ctx.save_for_backward(a2, b2)
        ctx.meta = (scale, a.dtype, b.dtype)
        return loss

    @staticmethod
    def backward(ctx, g):
        a2, b2 = ctx.saved_tensors
        scale, adt, bdt = ctx.meta
        g = _c(g.float())
        da = torch.empty_like(a2) if ctx.needs_input_grad[0] else None
        db = torch.empty_like(b2) if ctx.needs_input_grad[1] else None
        n = a2.numel()
        lib.xr_loss_mse(dt(a2), ptr(a2), ptr(b2), scale, 1.0, None, ptr(da), ptr(db), n, n, ptr(g), stream())
        if da is not None and da.dtype != adt:
            da = da.to(adt)
        if db is not None and db.dtype != bdt:
            db = db.to(bdt)
        return da, db, None


class _Landmark(Function):
    """MSELoss_Landmark (loss/loss.py:28-31); pred NCHW fp32, target (N,H,W) fp32."""

    @staticmethod
    def forward(ctx, pred, target, scale):
        _need_cuda(pred)
        p = _c(pred.detach().float())
        t = _c(target.detach().float())
        N, C, H, W = p.shape
        assert t.numel() == N * H * W
        loss = zeros_f32((1,), p.device).view(())
        lib.xr_loss_landmark(ptr(p), ptr(t), scale, 1.0, ptr(loss), None, N, C, H * W, None, stream())
        ctx.save_for_backward(p, t)
        ctx.meta = (scale, pred.dtype)
        return loss

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        scale, pdt = ctx.meta
        g = _c(g.float())
        N, C, H, W = p.shape
        dp = torch.empty_like(p)
        lib.xr_loss_landmark(ptr(p), ptr(t), scale, 1.0, None, ptr(dp), N, C, H * W, ptr(g), stream())
        return dp.to(pdt), None, None


class _CE2d(Function):
    """CrossEntropyLoss2d (loss/loss.py:61-62): NLL(log_softmax(pred, 1), squeeze(target)); NCHW fp32."""

    @staticmethod
    def forward(ctx, pred, target):
        _need_cuda(pred)
        p = _c(pred.detach().float())
        N, C, H, W = p.shape
        t = _c(target.detach().reshape(N, H * W).long())
        loss = zeros_f32((1,), p.device).view(())
        lib.xr_loss_ce_nchw(ptr(p), ptr(t), 1.0, ptr(loss), None, N, C, H * W, None, stream())
        ctx.save_for_backward(p, t)
        ctx.pdt = pred.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        p, t = ctx.saved_tensors
        g = _c(g.float())
        N, C, H, W = p.shape
        dp = torch.empty_like(p)
        lib.xr_loss_ce_nchw(ptr(p), ptr(t), 1.0, None, ptr(dp), N, C, H * W, ptr(g), stream())
        return dp.to(ctx.pdt), None


class _CERows(Function):
    """nn.CrossEntropyLoss on (M, C) logits (main.py:132; train_teacher_model.py:190)."""

    @staticmethod
    def forward(ctx, logits, target):
        _need_cuda(logits)
        x = logits.detach()
        if x.dtype not in (torch.float32, torch.bfloat16):
            x = x.float()
        x = _c(x)
        M, C = x.shape
        t = _c(target.detach().long())
        loss = zeros_f32((1,), x.device).view(())
        lib.xr_loss_softmax_ce(dt(x), ptr(x), ptr(t), 1.0, ptr(loss), None, M, C, C, None, stream())
        ctx.save_for_backward(x, t)
        ctx.ldt = logits.dtype
        return loss

    @staticmethod
    def backward(ctx, g):
        x, t = ctx.saved_tensors
        g = _c(g.float())
        M, C = x.shape
        dx = torch.empty_like(x)
        lib.xr_loss_softmax_ce(dt(x), ptr(x), ptr(t), 1.0, None, ptr(dx), M, C, C, ptr(g), stream())
        return dx.to(ctx.ldt), None


def mse_loss(a, b, scale=1.0):
    return _MSE.apply(a, b, float(scale))


def landmark_loss(pred, target, scale=97.0):
    return _Landmark.apply(pred, target, float(scale))


def cross_entropy_2d(pred, target):
    return _CE2d.apply(pred, target)


def cross_entropy(logits, target):
    return _CERows.apply(logits, target)


# ------------------------------------------------------------------------------------------------- ArcFace head
class _L2NormRows(Function):
    """y = x / ||x||_2 per row (l2_norm, model_irse.py:16-20); fp32."""

    @staticmethod
    def forward(ctx, x):
        _need_cuda(x)
        x = _c(x.float())
        M, C = x.shape
        y = torch.empty_like(x)
        inv = torch.empty(M, dtype=torch.float32, device=x.device)
        lib.xr_l2norm_rows(ptr(x), ptr(y), ptr(inv), M, C, stream())
        ctx.save_for_backward(y, inv)
        return y

    @staticmethod
    def backward(ctx, dy):
        y, inv = ctx.saved_tensors
        dy = _c(dy.float())
        dx = torch.empty_like(y)
        lib.xr_l2norm_rows_bwd(ptr(y), ptr(inv), ptr(dy), ptr(dx), y.shape[0], y.shape[1], stream())
        return dx


class _ArcMargin(Function):
    """logits = s * (phi(cos) at the target column, cos elsewhere), phi = cos(theta + m) with the usual
    cos(pi - m) fallback (build-defined ArcFace margin)."""

    @staticmethod
    def forward(ctx, cos, target, s, m):
        cos = _c(cos.float())
        M, C = cos.shape
        out = cos.clone()
        dphi = torch.ones(M, dtype=torch.float32, device=cos.device)
        t = _c(target.long())
        lib.xr_arcface_margin(ptr(out), ptr(t), ptr(dphi), M, C, s, m, stream())
        ctx.save_for_backward(dphi, t)
        ctx.s = s
        return out

    @staticmethod
    def backward(ctx, dlogits):
        dphi, t = ctx.saved_tensors
        g = dlogits.float() * ctx.s
        rows = torch.arange(g.shape[0], device=g.device)
        g[rows, t] = g[rows, t] * dphi  # host-side index fix-up on N elements (plumbing)
        return g, None, None, None


def l2norm_rows(x):
    return _L2NormRows.apply(x)


def arcface_logits(emb, weight, target, s=64.0, m=0.5):
    """emb (N, D), weight (classes, D) -> (N, classes) fp32 margin logits.  The cosine matrix is an implicit-GEMM
    launch in XR_F32 mode (3-plane split MFMA) on the row-normalised operands."""
    e = l2norm_rows(emb)
    w = l2norm_rows(weight)
    n, d = e.shape
    cos = leave2d(linear_nhwc(e.reshape(n, 1, 1, d), w, None))[:, :w.shape[0]]
    return _ArcMargin.apply(cos, target, float(s), float(m))


# ------------------------------------------------------------------------------------------------- MMD
class _MMD(Function):
    """Biased multi-bandwidth Gaussian-kernel MMD^2 between two (N, D) batches (build-defined)."""

    @staticmethod
    def forward(ctx, a, b, sigmas):
        _need_cuda(a)
        assert a.shape == b.shape and a.dim() == 2
        z = torch.cat((a.detach().float(), b.detach().float()), 0).contiguous()
        n, d = a.shape
        sg = torch.tensor(list(sigmas), dtype=torch.float32, device=a.device)
        w = torch.empty((2 * n, 2 * n), dtype=torch.float32, device=a.device)
        loss = zeros_f32((1,), a.device).view(())
        lib.xr_mmd_fwd(ptr(z), ptr(w), ptr(loss), n, d, ptr(sg), sg.numel(), stream())
        ctx.save_for_backward(z, w)
        ctx.meta = (n, d, a.dtype, b.dtype)
        return loss

    @staticmethod
    def backward(ctx, g):
        z, w = ctx.saved_tensors
        n, d, adt, bdt = ctx.meta
        g = _c(g.float())
        dz = torch.empty_like(z)
        lib.xr_mmd_bwd(ptr(z), ptr(w), ptr(dz), n, d, ptr(g), stream())
        return dz[:n].to(adt), dz[n:].to(bdt), None


def mmd(a, b, sigmas=(1.0, 2.0, 4.0, 8.0, 16.0)):
    return _MMD.apply(a, b, tuple(float(s) for s in sigmas))
