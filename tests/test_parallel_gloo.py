"""CPU, world_size 2 over gloo: the data-parallel layer (flat parameter/gradient buffers, bucketed overlapped
all-reduce from post-accumulate-grad hooks, rank-0 broadcast) on a small stock-torch model."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "cross-resolution-face-recognition_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xrface import parallel
    torch.manual_seed(100 + rank)  # deliberately different initial weights per rank
    net = torch.nn.Sequential(torch.nn.Linear(37, 53), torch.nn.ReLU(), torch.nn.Linear(53, 11), torch.nn.Linear(11, 3))
    unused = torch.nn.Parameter(torch.ones(5))  # a registered-but-unused parameter (never gets a gradient)
    net.register_parameter("unused", unused)
    parallel.broadcast_module(net)
    w0 = torch.cat([p.detach().reshape(-1) for p in net.parameters()])
    flat = parallel.FlatParams(net.parameters())
    red = parallel.BucketedAllReduce(flat, bucket_mb=0.004)  # ~1000 floats per bucket -> several buckets
    assert len(red.buckets) >= 2
    torch.manual_seed(7 + rank)
    x = torch.randn(16, 37)
    flat.zero_grad()
    net(x).square().mean().backward()
    local = flat.grad.clone()
    red.finish()
    q.put((rank, w0.numpy(), local.numpy(), flat.grad.clone().numpy()))
    # second step re-uses the reducer state
    flat.zero_grad()
    net(x * 0.5).square().mean().backward()
    red.finish()
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_bucketed_allreduce_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=120) for _ in range(2)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    (_, w_a, loc_a, avg_a), (_, w_b, loc_b, avg_b) = res
    assert (w_a == w_b).all(), "broadcast must make replicas identical"
    assert not (loc_a == loc_b).all(), "ranks saw different data"
    expect = (torch.tensor(loc_a) + torch.tensor(loc_b)) / 2
    assert torch.allclose(torch.tensor(avg_a), expect, atol=1e-7)
    assert torch.allclose(torch.tensor(avg_b), expect, atol=1e-7)


# ---------------------------------------------------------------------------------------------------------------------
# The HIP backward kernels accumulate parameter gradients in place and signal the reducer once per USE SITE
# (ops._direct_done), not once per parameter.  _DirectLinear reproduces exactly that signalling pattern on the CPU.
class _DirectLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w):
        ctx.save_for_backward(x)
        ctx.w = w
        return x @ w.t()

    @staticmethod
    def backward(ctx, dy):
        from xrface import ops
        (x,) = ctx.saved_tensors
        w = ctx.w
        tgt = ops._direct(w)
        assert tgt is not None
        tgt.add_(dy.t() @ x)
        ops._direct_done(w)
        return dy @ w, None


def _shared_worker(rank, world, port, q):
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(root, "cross-resolution-face-recognition_amd"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from xrface import parallel
    torch.manual_seed(5)
    ws = [torch.nn.Parameter(torch.randn(24, 24) * 0.2) for _ in range(4)]
    flat = parallel.FlatParams(ws)
    red = parallel.BucketedAllReduce(flat, bucket_mb=0.002, overlap=True)   # ~520 floats -> one parameter per bucket
    assert len(red.buckets) == 4

    def fwd(x):
        # ws[1] is applied at THREE sites, ws[2] at two (the shared FSRNet trunk pattern, model/FSRnet.py:331-333)
        for i in (0, 1, 2, 1, 2, 1, 3):
            x = torch.tanh(_DirectLinear.apply(x, ws[i]))
        return x

    torch.manual_seed(40 + rank)
    xs = [torch.randn(8, 24) for _ in range(3)]
    out = []
    for step in range(3):
        flat.zero_grad()
        fwd(xs[step]).square().mean().backward()
        # local reference gradient by plain autograd on detached copies
        cp = [w.detach().clone().requires_grad_(True) for w in ws]
        x = xs[step]
        for i in (0, 1, 2, 1, 2, 1, 3):
            x = torch.tanh(x @ cp[i].t())
        x.square().mean().backward()
        local = torch.cat([c.grad.reshape(-1) for c in cp])
        order_before_finish = list(red.launch_order)
        red.finish()
        out.append((local.numpy(), flat.grad.clone().numpy(), order_before_finish, list(red.last_launch_order)))
    q.put((rank, out))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_shared_parameters_with_overlap_world2():
    """A parameter used at several sites signals the reducer several times per step: buckets must go out only when every
    site of every parameter has accumulated (learned in step 1), in reverse execution order, and the result must be the
    exact average of the ranks' complete local gradients."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shared_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for step in range(3):
        loc0, avg0, early0, order0 = res[0][step]
        loc1, avg1, early1, order1 = res[1][step]
        expect = (torch.tensor(loc0) + torch.tensor(loc1)) / 2
        assert torch.allclose(torch.tensor(avg0), expect, atol=1e-6), step
        assert torch.allclose(torch.tensor(avg1), expect, atol=1e-6), step
        if step == 0:
            assert early0 == [], "step 1 only learns the signal counts: nothing may be launched before finish()"
        else:
            # buckets are numbered from the LAST parameter: bucket 0 = ws[3] (finishes first in backward), 3 = ws[0] (last);
            # the shared ws[1] / ws[2] complete only at their FIRST forward site, i.e. late in backward
            assert early0 == [0, 2, 1, 3] or early0 == [0, 1, 2, 3], early0
            assert early0[0] == 0 and early0[-1] == 3 and len(early0) == 4, "all buckets overlapped with backward"
        assert sorted(order0) == [0, 1, 2, 3]
