"""GPU: whole-step HIP graphs (xrface.graph.GraphedStep) reproduce the eager step."""
import copy

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _rel(a, b):
    return float((a.double() - b.double()).abs().max() / max(float(b.double().abs().max()), 1e-30))


def _faces(n, seed):
    g = torch.Generator(device=DEV).manual_seed(seed)
    return torch.rand(n, 3, 112, 112, device=DEV, generator=g) * 2 - 1


@pytest.mark.parametrize("optname", ["sgd", "adam"])
def test_graphed_resnet34_step_matches_eager(optname):
    """Same initial weights, same batches: N replays of the captured step track N eager steps.  Two eager runs of this
    tiny-batch train-mode network already differ by a few 1e-3 (fp32 atomics order x ill-conditioned BatchNorm), so the
    bar is 3e-2 on the parameters and 5 % on the losses; exactness of the optimizer under replay is test_graphed_optimizers_exact."""
    import xrface
    from xrface import parallel
    from xrface.graph import GraphedStep
    from xrface.loss.loss import CrossEntropyLoss
    from xrface.model.resnet import ResNet_34

    xrface.set_compute_dtype(torch.float32)
    torch.manual_seed(3)
    m_e = ResNet_34().to(DEV).train()
    m_g = copy.deepcopy(m_e)
    crit = CrossEntropyLoss()
    y = torch.randint(0, 512, (8,), device=DEV)
    batches = [_faces(8, s) for s in range(6)]

    def make(model):
        flat = parallel.FlatParams(model.parameters())
        if optname == "sgd":
            opt = parallel.FusedSGD(flat, lr=0.01, momentum=0.9, weight_decay=1e-4)
        else:
            opt = parallel.FusedAdam(flat, lr=1e-3, betas=(0.5, 0.999))
        loss_buf = torch.zeros((), device=DEV)

        def step(x):
            opt.zero_grad()
            out = model(x)
            loss = crit(out[0] if isinstance(out, (tuple, list)) else out, y)
            loss.backward()
            opt.step()
            loss_buf.copy_(loss.detach())
            return loss_buf
        return flat, step

    flat_e, step_e = make(m_e)
    flat_g, step_g = make(m_g)
    # GraphedStep warms up with 3 eager steps on its example input; give the eager twin the same 3 steps
    for _ in range(3):
        step_e(batches[0])
    gs = GraphedStep(step_g, [batches[0]], warmup=3)
    assert _rel(flat_g.flat, flat_e.flat) < 3e-2
    losses_e, losses_g = [], []
    for x in batches[1:]:
        losses_e.append(float(step_e(x)))
        losses_g.append(float(gs(x)))
    assert _rel(flat_g.flat, flat_e.flat) < 3e-2, (losses_e, losses_g)
    for a, b in zip(losses_e, losses_g):
        assert abs(a - b) < 5e-2 * max(1.0, abs(a))
    sd_e, sd_g = m_e.state_dict(), m_g.state_dict()
    k = next(k for k in sd_e if k.endswith("num_batches_tracked"))
    assert int(sd_e[k]) == int(sd_g[k]) == 3 + len(batches) - 1   # BatchNorm counters advance inside the graph too


@pytest.mark.parametrize("optname", ["sgd", "rmsprop", "adam"])
def test_graphed_optimizers_exact(optname):
    """A deterministic problem (gradient = p - target, no atomics): replays of the captured update equal eager updates to
    fp32 round-off -- Adam's bias corrections advance with the device-side counter, not with the frozen host step."""
    from xrface import parallel
    from xrface.graph import GraphedStep

    torch.manual_seed(5)
    target = torch.randn(4096, device=DEV)

    def make():
        p = torch.nn.Parameter(torch.zeros(4096, device=DEV))
        flat = parallel.FlatParams([p])
        opt = {"sgd": lambda: parallel.FusedSGD(flat, lr=0.1, momentum=0.9, weight_decay=1e-3),
               "rmsprop": lambda: parallel.FusedRMSprop(flat, lr=1e-2, alpha=0.9, weight_decay=1e-3),
               "adam": lambda: parallel.FusedAdam(flat, lr=0.05, betas=(0.8, 0.99))}[optname]()

        def step(t):
            flat.grad.copy_(flat.flat - t)
            opt.step()
            return flat.flat
        return flat, step

    flat_e, step_e = make()
    flat_g, step_g = make()
    for _ in range(2):
        step_e(target)
    gs = GraphedStep(step_g, [target], warmup=2)
    for _ in range(7):
        step_e(target)
        gs(target)
    torch.cuda.synchronize()
    assert _rel(flat_g.flat, flat_e.flat) < 1e-5


def test_graphed_dropout_draws_fresh_masks():
    """Dropout inside a replayed graph: the keep mask changes from replay to replay (device-side tick) and the backward of
    each replay uses the mask of its own forward."""
    import xrface
    from xrface import ops
    from xrface.graph import GraphedStep

    xrface.set_compute_dtype(torch.float32)
    x = torch.ones(8, 4, 4, 64, device=DEV)
    out = torch.zeros_like(x)
    grad = torch.zeros_like(x)

    def step(xin):
        xr = xin.clone().requires_grad_(True)
        yb = ops.dropout(xr, 0.4, True)
        yb.sum().backward()
        out.copy_(yb.detach())
        grad.copy_(xr.grad)
        return out

    gs = GraphedStep(step, [x], warmup=2)
    seen = []
    for _ in range(4):
        gs(x)
        torch.cuda.synchronize()
        assert torch.equal(out, grad)                       # d(sum)/dx = keep/(1-p) = forward output on an all-ones input
        keep = float((out != 0).float().mean())
        assert 0.5 < keep < 0.7
        seen.append(out.clone())
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2]) and not torch.equal(seen[2], seen[3])


def test_side_stream_wgrad_matches_single_stream():
    """FlatParams(direct=True): weight gradients computed on the side stream (overlapping the rest of backward) equal the
    single-stream gradients; checked over several steps with recycled activation memory (allocator record_stream path)."""
    import xrface
    from xrface import ops, parallel
    from xrface.loss.loss import CrossEntropyLoss
    from xrface.model.model_irse import IR_SE_50

    xrface.set_compute_dtype(torch.float32)
    torch.manual_seed(11)
    m1 = IR_SE_50([112, 112]).to(DEV).train()
    m2 = copy.deepcopy(m1)
    for m in (m1, m2):
        for mod in m.modules():
            if isinstance(mod, torch.nn.Dropout):
                mod.p = 0.0
    f1, f2 = parallel.FlatParams(m1.parameters()), parallel.FlatParams(m2.parameters())
    crit = CrossEntropyLoss()
    y = torch.randint(0, 512, (6,), device=DEV)
    old = ops._cfg["wgrad_stream"]
    try:
        for it in range(3):
            x = _faces(6, 40 + it)
            for flat, model, mode in ((f1, m1, 0), (f2, m2, 1)):
                ops._cfg["wgrad_stream"] = mode
                flat.zero_grad()
                crit(model(x), y).backward()
                junk = [torch.empty(1 << 22, device=DEV).normal_() for _ in range(8)]   # churn the allocator right after
                del junk
            torch.cuda.synchronize()
            # bar = the single-stream path's own run-to-run spread on this tiny-batch train-mode network (fp32 atomics order
            # amplified by BatchNorm: up to 5e-3); a race would be orders of magnitude above it
            assert _rel(f2.grad, f1.grad) < 1e-2, it
            assert float(f1.grad.abs().max()) > 0
    finally:
        ops._cfg["wgrad_stream"] = old


def test_graph_capture_with_side_stream_enabled():
    """With the weight-gradient side stream switched on (the default), a captured step stays single-stream and trains."""
    import xrface
    from xrface import ops, parallel
    from xrface.graph import GraphedStep
    from xrface.loss.loss import CrossEntropyLoss
    from xrface.model.resnet import ResNet_34

    xrface.set_compute_dtype(torch.bfloat16)
    torch.manual_seed(2)
    model = ResNet_34().to(DEV).train()
    flat = parallel.FlatParams(model.parameters())
    opt = parallel.FusedSGD(flat, lr=0.01, momentum=0.9)
    crit = CrossEntropyLoss()
    y = torch.randint(0, 512, (8,), device=DEV)
    loss_buf = torch.zeros((), device=DEV)
    old = ops._cfg["wgrad_stream"]
    ops._cfg["wgrad_stream"] = 1
    try:
        def step(x):
            opt.zero_grad()
            loss = crit(model(x)[0], y)
            loss.backward()
            opt.step()
            loss_buf.copy_(loss.detach())
            return loss_buf
        x = _faces(8, 1)
        gs = GraphedStep(step, [x], warmup=3)
        first = float(gs(x))
        for _ in range(20):
            last = float(gs(x))
        assert last == last and last < first     # still training on the fixed batch
        gs.close()
        with pytest.raises(RuntimeError, match="after close"):
            gs(x)
    finally:
        ops._cfg["wgrad_stream"] = old
        xrface.set_compute_dtype(torch.float32)


def test_graph_capture_keeps_the_side_stream_and_replays_equal_eager_steps():
    """GraphedStep(side_stream=True): the weight-gradient fork / join is captured as two branches of the graph.  In the deterministic
    mode (every fp32 sum in a fixed order) concurrency cannot change a result, so N replays must leave exactly the parameters N
    eager two-stream steps leave -- bit for bit -- and the captured graph must really contain side-stream work."""
    import xrface
    from xrface import ops, parallel
    from xrface.graph import GraphedStep
    from xrface.loss.loss import CrossEntropyLoss
    from xrface.model import model_irse

    xrface.set_compute_dtype(torch.bfloat16)
    xrface.set_deterministic(True)
    old = ops._cfg["wgrad_stream"]
    ops._cfg["wgrad_stream"] = 1
    try:
        torch.manual_seed(4)
        m_e = model_irse.IR_SE_50([112, 112]).to(DEV).train()
        m_e.output_layer[1].p = 0.0                      # dropout off: the two runs draw from different counters
        m_g = copy.deepcopy(m_e)
        crit = CrossEntropyLoss()
        y = torch.randint(0, 512, (16,), device=DEV)
        x = _faces(16, 9)

        def make(model):
            flat = parallel.FlatParams(model.parameters_in_execution_order())
            opt = parallel.FusedSGD(flat, lr=0.01, momentum=0.9, weight_decay=5e-4)
            lbuf = torch.zeros((), device=DEV)

            def step(xx):
                opt.zero_grad()
                loss = crit(model(xx), y)
                loss.backward()
                opt.step()
                lbuf.copy_(loss.detach())
                return lbuf
            return flat, step
        flat_e, step_e = make(m_e)
        flat_g, step_g = make(m_g)
        forks = [0]
        done0 = ops._side_done

        def counting_done(side, tensors):
            if ops._graph["capturing"]:
                forks[0] += 1
            return done0(side, tensors)
        ops._side_done = counting_done
        try:
            gs = GraphedStep(step_g, [x], warmup=2, side_stream=True)
        finally:
            ops._side_done = done0
        assert forks[0] > 20, f"only {forks[0]} groups of side-stream launches inside the capture"
        for _ in range(2):                                # the eager twin takes the warm-up's two steps
            step_e(x)
        for _ in range(3):
            le = float(step_e(x))
            lg = float(gs(x))
            assert le == lg, (le, lg)
        torch.cuda.synchronize()
        assert torch.equal(flat_e.flat, flat_g.flat)
        gs.close()
    finally:
        ops._cfg["wgrad_stream"] = old
        xrface.set_deterministic(False)
        xrface.set_compute_dtype(torch.float32)


def test_eval_between_replays_sees_current_weights_and_statistics():
    """train/validate loop around a captured step: replays move the parameters and the BatchNorm running statistics without
    bumping any tensor version, so the host-side caches (weight packs, eval-mode BatchNorm scale/shift) must be invalidated
    by the replay itself.  Checked exactly: after the replays the cached-path eval forward must equal the eval forward of a
    FRESH module built from the replayed model's own state_dict (no caches at all), and must differ from the eval forward
    taken before the replays."""
    import xrface
    from xrface import parallel
    from xrface.graph import GraphedStep
    from xrface.loss.loss import CrossEntropyLoss
    from xrface.model.resnet import ResNet_34

    xrface.set_compute_dtype(torch.float32)
    torch.manual_seed(5)
    model = ResNet_34().to(DEV).train()
    other = ResNet_34().to(DEV).train()        # a second model sharing the global weight-pack plan (pending refreshes)
    crit = CrossEntropyLoss()
    y = torch.randint(0, 512, (8,), device=DEV)
    x, xv = _faces(8, 3), _faces(4, 9)
    flat = parallel.FlatParams(model.parameters())
    opt = parallel.FusedSGD(flat, lr=0.05, momentum=0.9)
    flat_o = parallel.FlatParams(other.parameters())
    opt_o = parallel.FusedSGD(flat_o, lr=0.05, momentum=0.9)
    buf = torch.zeros((), device=DEV)

    def step(xin):
        opt.zero_grad()
        loss = crit(model(xin)[0], y)
        loss.backward()
        opt.step()
        buf.copy_(loss.detach())
        return buf

    def validate(m):
        m.eval()
        with torch.no_grad():
            e = m(xv)[0].float().clone()
        m.train()
        return e

    for _ in range(2):                         # leaves `other`'s packs stale when the capture below warms up
        opt_o.zero_grad()
        crit(other(x)[0], y).backward()
        opt_o.step()
    with GraphedStep(step, [x], warmup=2) as gs:
        v0 = validate(model)                   # fills the eval caches (packs + BatchNorm coefficients)
        for _ in range(5):
            gs(x)
        v1 = validate(model)
        torch.cuda.synchronize()
        fresh = ResNet_34().to(DEV)
        fresh.load_state_dict(model.state_dict())
        v1_fresh = validate(fresh)
        assert _rel(v1, v0) > 1e-2, "stale eval caches: the eval forward did not change across 5 replays"
        assert _rel(v1, v1_fresh) < 1e-4, "eval after replays must use the replayed weights / running statistics"


def test_eval_coefficients_follow_running_statistics_in_eager_mode():
    """Eager mode: a train-mode forward rewrites running_mean / running_var through raw pointers (no version bump);
    the next eval forward must not reuse scale/shift cached from the older statistics."""
    import xrface
    from xrface import nn as xnn

    xrface.set_compute_dtype(torch.float32)
    bn = xnn.BatchNorm2d(16).to(DEV)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(4, 16, 8, 8, device=DEV, generator=g) * 3 + 2
    bn.eval()
    with torch.no_grad():
        y0 = bn(x).clone()
    bn.train()
    with torch.no_grad():
        bn(x)
    bn.eval()
    with torch.no_grad():
        y1 = bn(x).clone()
    ref = torch.nn.functional.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.1, bn.eps)
    assert _rel(y1, ref) < 1e-5
    assert _rel(y1, y0) > 1e-2


def test_graphed_sr_perceptual_step_replays_and_trains():
    """SURVEY 8f-1/2 as one HIP graph: the SR-variant perceptual-loss step (train_FHN.py:251-308; per-pair gradients through
    torch.autograd.grad, three fused Adam updates) captured once and replayed -- the first replay reproduces an eager step from
    the same weights, and replays on a fixed batch reduce the generator losses."""
    import copy
    import xrface
    from xrface import parallel, steps
    from xrface.graph import GraphedStep
    from xrface.model import FSRnet_sr as M, model_irse

    xrface.set_compute_dtype(torch.bfloat16)
    try:
        torch.manual_seed(4)
        n = 2
        nets = {"coarse": M.Coarse_SR_Network().to(DEV), "encoder": M.Fine_SR_Encoder().to(DEV),
                "prior": M.Prior_Estimation_Network().to(DEV), "decoder": M.Fine_SR_Decoder().to(DEV)}
        bb = model_irse.IR_50([112, 112]).to(DEV).eval()
        for p_ in bb.parameters():
            p_.requires_grad_(False)
        flats = {"coarse": parallel.FlatParams(nets["coarse"].parameters()), "prior": parallel.FlatParams(nets["prior"].parameters()),
                 "encdec": parallel.FlatParams(list(nets["encoder"].parameters()) + list(nets["decoder"].parameters()))}
        opts = {k: parallel.FusedAdam(f, lr=2e-4, betas=(0.5, 0.999)) for k, f in flats.items()}
        hr = _faces(n, 3)
        lr = torch.nn.functional.interpolate(torch.nn.functional.avg_pool2d(hr, 7), size=(112, 112), mode="bilinear").contiguous()
        hm = torch.rand(n, 112, 112, device=DEV)
        par = torch.randint(0, 13, (n, 1, 112, 112), device=DEV)
        lbuf = torch.zeros(3, device=DEV)

        def step(lr_, hr_, hm_, par_):
            for o in opts.values():
                o.zero_grad()
            l_, _ = steps.fhn_perceptual_step(nets, bb, lr_, hr_, hm_, par_, optimizers=opts)
            lbuf.copy_(torch.stack([l_["coarse"].float(), l_["prior"].float(), l_["encdec"].float()]))
            return lbuf

        gs = GraphedStep(step, [lr, hr, hm, par], warmup=2)
        first = gs(lr, hr, hm, par).clone()
        for _ in range(8):
            last = gs(lr, hr, hm, par).clone()
        assert torch.isfinite(first).all() and torch.isfinite(last).all()
        assert float(last[0]) < float(first[0]) and float(last[2]) < float(first[2]), (first.tolist(), last.tolist())
        gs.close()
    finally:
        xrface.set_compute_dtype(torch.float32)
