"""Whole-step HIP graphs.

The small-batch steps of this code base (FSRNet at N = 4-32, the per-pair FHN loop, the KD step at N = 64) are launch-bound:
700-3000 kernels of a few microseconds each, 10-30 ms of host time per step.  ``GraphedStep`` captures one complete step --
forward, backward, optimizer update, BatchNorm counter bumps, weight re-packs -- into a single HIP graph and replays it with
~0.3 ms of host work.  Everything the step reads from the host must be frozen or moved to the device:

* inputs live in static tensors (``__call__`` copies the new batch in);
* the zero-initialised scratch slab (statistic sums, loss scalars) is created inside the capture, so its memset is a
  graph node and every replay starts from zeros;
* dropout mixes a device-side step counter into its seed (``xr_dropout``'s ``tick``), incremented by the graph;
* FusedAdam's bias corrections follow the same device counter; hyper-parameters passed by value (learning rate) are
  frozen: re-capture after changing them.  Steps taken by replays are not reflected in the optimizer's host-side counter.

Replays change parameters and BatchNorm running statistics behind the host's back (no tensor version bump), so every
replay invalidates the host-side caches keyed on them (weight packs, eval-mode BatchNorm coefficients) -- for exactly the
parameters / layers the warm-up saw the step update: an eager evaluation between replays sees the current weights, and a
frozen network's packs are never rebuilt.  Mixing eager optimizer steps with replays of the same optimizer is not supported
(the replay's skip mask and step counter are those of the capture).

Teardown is explicit: ``close()`` (also run by ``__del__``) first waits for the device -- a replay may still be in flight
when the last reference dies, and destroying an executing graph or releasing its private memory pool under running
kernels is undefined -- then drops the tensors that live in the graph's pool and only then resets the graph.  Every
stream the capture touched is held for the lifetime of the object.
"""
from __future__ import annotations

import torch

from . import ops


class GraphedStep:
    def __init__(self, fn, example_inputs, warmup: int = 3, side_stream: bool = False):
        """fn(*tensors) -> tensor | tuple of tensors | None runs one full step; it must not synchronise with the host.
        side_stream=True keeps the weight-gradient side stream inside the capture: its fork (an event recorded on the capturing
        stream) and its join (the end-of-backward callback, the optimizers) become the two branches of the graph.  Measured
        slower than both the eager two-stream step and the single-stream graph at batch 256; kept for
        small / chunked batches where the overlap matters more than the branch bookkeeping of the graph runtime."""
        assert all(t.is_cuda for t in example_inputs), "GraphedStep: inputs must be device tensors"
        self.graph = None
        self.static_out = None
        self.static_in = [t.clone() for t in example_inputs]
        dev = self.static_in[0].device if self.static_in else torch.device("cuda", torch.cuda.current_device())
        self.device = dev
        cur = torch.cuda.current_stream(dev)
        side = torch.cuda.Stream(dev)
        self._streams = [cur, side]
        side.wait_stream(cur)
        self._touched = {"params": {}, "bns": {}}
        with torch.cuda.stream(side):   # warm-up: lazy initialisation, weight-pack tables, allocator pools
            for i in range(warmup):
                # the last warm-up step also records what the step moves behind the host's back (see __call__)
                ops._touch_log[0] = self._touched if i == warmup - 1 else None
                try:
                    fn(*self.static_in)
                finally:
                    ops._touch_log[0] = None
            # the set of stale weight packs the captured step will meet (this step's own parameters; other models' pending
            # refreshes were flushed by the warm-up) gets its descriptor table now -- a capture cannot build one
            ops._pack_plan.prebuild()
        cur.wait_stream(side)
        torch.cuda.synchronize(dev)
        if ops._graph["tick"] is None or ops._graph["tick"].device != dev:
            ops._graph["tick"] = torch.zeros(1, dtype=torch.int64, device=dev)
        self.tick = ops._graph["tick"]
        ops._graph["tick_ref"] = int(self.tick.item()) + 1   # the value the first replay sees
        ops.join_side_stream()   # nothing of the warm-up may still be pending on the weight-gradient stream
        graph = torch.cuda.CUDAGraph()
        ops._zpools.clear()
        ops._graph["capturing"] = True
        ops._graph["side_ok"] = bool(side_stream)
        # the device is idle (synchronised above): forget the warm-up's fork / join history, so that a join issued before the first
        # captured fork is a no-op instead of a dependency of the capturing stream on a stream that is not part of the capture
        ops._side["seq"], ops._side["joined"] = 0, {}
        try:
            with torch.cuda.graph(graph):
                self.tick.add_(1)
                self.static_out = fn(*self.static_in)
        except BaseException:
            # a step that raises inside the capture leaves torch's context manager half unwound (its capture_end() raises as
            # well, before the previous stream is restored): put the caller's stream back so that later work does not run on
            # the abandoned capture stream
            try:
                torch.cuda.set_stream(cur)
            except Exception:
                pass
            self.static_out = None
            raise
        finally:
            ops._graph["capturing"] = False
            ops._graph["side_ok"] = False
            ops._side["seq"], ops._side["joined"] = 0, {}
            ops._zpools.clear()   # the slab captured above belongs to the graph's memory pool
            if side_stream and ops._side["stream"] is not None:   # every stream / event the capture touched lives as long as the graph
                self._streams += [ops._side["stream"], ops._side["ev"]]
        self.graph = graph

    def __call__(self, *inputs):
        if self.graph is None:
            raise RuntimeError("GraphedStep: called after close()")
        assert len(inputs) == len(self.static_in)
        for s, t in zip(self.static_in, inputs):
            if s.data_ptr() != t.data_ptr():
                s.copy_(t, non_blocking=True)
        self.graph.replay()
        # the replay moved parameters / running statistics without bumping any version: invalidate the host-side caches keyed on
        # them -- exactly the parameters the step's fused optimizers update and the BatchNorm layers it runs in training mode
        # (recorded in the warm-up); the packs / eval coefficients of everything else (a frozen teacher) stay valid.  A step that
        # recorded no fused update (stock optimizers write through tensor ops the capture froze) invalidates everything.
        params = [r() for r in self._touched["params"].values()]
        if params:
            ops.invalidate_weight_cache([p for p in params if p is not None])
            for r in self._touched["bns"].values():
                m = r()
                if m is not None:
                    m.__dict__["_xr_stat_epoch"] = m.__dict__.get("_xr_stat_epoch", 0) + 1
        else:
            ops.invalidate_weight_cache()
        return self.static_out

    def close(self):
        """Release the captured graph and its private memory pool at a defined point (idempotent)."""
        graph, self.graph = self.graph, None
        if graph is None:
            return
        try:
            torch.cuda.synchronize(self.device)   # no replay may be executing while its graph / pool goes away
        except Exception:   # interpreter shutdown
            pass
        self.static_out = None                    # tensors allocated from the graph's pool go first
        self.static_in = []
        try:
            graph.reset()
        except Exception:
            pass
        del graph
        self._streams = []

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
