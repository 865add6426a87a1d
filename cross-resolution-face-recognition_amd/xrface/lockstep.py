"""Lockstep execution of independent networks on their own streams.

A training step of one convolutional network on this path is a chain of MFMA-bound convolutions and HBM-bound normalisation
passes with nothing to overlap them with.  Two INDEPENDENT chains of the same kind of work -- the student and the assistant of
the residual-KD step (distill_main.py:59-74: different weights, same layers) -- advanced stage by stage in lockstep, each on its
own stream, let the convolutions of one run under the elementwise passes of the other.  Because the autograd nodes are created
alternately, the backward pass interleaves the same way (every node runs on the stream of its forward).

A network takes part by offering ``lockstep_plan(...) -> (stages, tap_after, finish)``: ``stages`` a list of callables on the
NHWC compute buffer, ``tap_after`` the stage indices whose outputs are collected, ``finish(last, tapped)`` the conversion to the
module's output tuple.  The caller's stream waits for all chains before the outputs are used; after ``backward()`` it must wait
again (``join``): a chain fed with a detached input hands nothing back through autograd.
"""
from __future__ import annotations

import torch

from . import ops
from .nn import batched_bn_counters
from .ops import enter

_STREAMS = {}


def _streams(dev, n, prio):
    key = (dev.index, n)
    if key not in _STREAMS or _STREAMS[key][0] != prio:
        _STREAMS[key] = (prio, [torch.cuda.Stream(dev, priority=prio) for _ in range(n)])
    return _STREAMS[key][1]


def plans_compatible(plans):
    return len({len(p_[0]) for p_ in plans}) == 1


def run_lockstep(nets, xs, plans):
    """[finish_j(...)] for every (net_j, x_j, plan_j); all plans must have the same number of stages."""
    assert plans_compatible(plans), "lockstep needs chains with the same number of stages"
    dev = xs[0].device
    main = torch.cuda.current_stream(dev)
    # the chains get the CALLER's stream priority: with a higher one, the chain that hands nothing back to the caller (the
    # assistant) would outrank the work that follows on the caller's stream (the FHN backward) -- measured +10 % on C4
    streams = _streams(dev, len(nets), getattr(main, "priority", 0))
    # stale weight packs are refreshed by ONE launch for every registered parameter (ops._PackPlan): it must run on the caller's
    # stream BEFORE the fork -- triggered lazily by the first convolution of one chain it would rewrite the other chain's packs
    # on a stream the other chain does not wait for
    if ops._cfg["pack_plan"]:
        ops._pack_plan.refresh()
    for s in streams:
        s.wait_stream(main)
    depth = len(plans[0][0])
    with batched_bn_counters(list(nets)):
        ys, tapped = [None] * len(nets), [[] for _ in nets]
        for j, (x, s) in enumerate(zip(xs, streams)):
            with torch.cuda.stream(s):
                ys[j] = enter(x)
        for i in range(depth):
            for j, s in enumerate(streams):
                with torch.cuda.stream(s):
                    ys[j] = plans[j][0][i](ys[j])
                    if i in plans[j][1]:
                        tapped[j].append(ys[j])
        outs = []
        for j, s in enumerate(streams):
            with torch.cuda.stream(s):
                outs.append(plans[j][2](ys[j], tapped[j]))
    for s in streams:
        main.wait_stream(s)
    return outs


def join(device, n):
    """The caller's stream waits for the lockstep streams (call after backward(), before the optimizers read the gradients)."""
    main = torch.cuda.current_stream(device)
    for s in _STREAMS.get((device.index, n), (0, ()))[1]:
        main.wait_stream(s)
