"""Parameter-holder layers: genuine ``torch.nn`` subclasses (so ``isinstance(m, nn.Conv2d)`` checks in the
reference's ``weights_init`` -- Face_Hallucination_sub_Net.py:368-380 -- and ``state_dict`` keys keep working)
whose compute runs on the HIP path.

Each layer has two entry points:
  ``f(buf, ...)``   internal: NHWC buffer in, NHWC buffer out (used by the fused model forwards);
  ``forward(x)``    user-facing: logical NCHW tensor in/out (used when a script calls a child directly, e.g.
                    ``backbone.input_layer(x)`` in SUPER_RESOLUTION/train_FHN.py:255).
"""
from __future__ import annotations

import torch
import torch.nn as nn

from . import ops
from .ops import enter, leave, leave2d


class Conv2d(nn.Conv2d):
    def f(self, buf, stats_link=None):
        """stats_link: an ops.StatsLink shared with the training-mode BatchNorm that directly consumes the output."""
        assert self.stride[0] == self.stride[1] and self.padding[0] == self.padding[1] and self.groups == 1 \
            and self.dilation == (1, 1), "xrface.Conv2d: square stride/padding, no groups/dilation"
        return ops.conv2d(buf, self.weight, self.bias, self.stride[0], self.padding[0], stats_link)

    def f_pass(self, buf, stats_link=None):
        """(conv(buf), buf'): buf' aliases buf and carries the residual-branch gradient into this conv's dgrad epilogue."""
        return ops.conv2d_pass(buf, self.weight, self.bias, self.stride[0], self.padding[0], stats_link)

    def forward(self, x):
        return leave(self.f(enter(x)), self.out_channels)


class ConvTranspose2d(nn.ConvTranspose2d):
    def f(self, buf):
        return ops.conv_transpose2d(buf, self.weight, self.bias, self.stride[0], self.padding[0], self.output_padding[0])

    def forward(self, x, output_size=None):
        return leave(self.f(enter(x)), self.out_channels)


class Linear(nn.Linear):
    def f(self, buf):
        """buf: [N,H,W,C] NHWC feature map (flattened in the reference's C,H,W order by the weight pack)."""
        return ops.linear_nhwc(buf, self.weight, self.bias)

    def forward(self, x):
        if x.dim() == 4:
            return leave2d(self.f(enter(x)))[:, :self.out_features]
        n, feat = x.shape
        buf = enter(x.reshape(n, feat, 1, 1))
        return leave2d(self.f(buf))[:, :self.out_features]


class InstanceNorm2d(nn.InstanceNorm2d):
    """eps 1e-5, no running statistics (model/FSRnet.py:81,112): identical in train and eval."""

    def f(self, buf, res=None, act=None, alpha=None):
        assert not self.track_running_stats
        return ops.norm_act(buf, self.weight, self.bias, None, None, res, alpha, "in", act, True, 0.0, self.eps)

    def forward(self, x):
        return leave(self.f(enter(x)))


def buf_is_cuda(t):
    return t is not None and t.is_cuda


def _any_requires_grad(bn):
    return (bn.weight is not None and bn.weight.requires_grad) or (bn.bias is not None and bn.bias.requires_grad)


_NBT_BATCHED = [False]  # set by batched_bn_counters(): the per-layer counter increments were issued as one foreach op


class batched_bn_counters:
    """Context: bump ``num_batches_tracked`` of every tracking BatchNorm under ``root`` with ONE multi-tensor add instead
    of one tiny kernel per layer (54 launches per IR-SE-50 forward), and make the layers skip their own increment."""

    def __init__(self, root):
        roots = list(root) if isinstance(root, (list, tuple)) else [root]      # several independent networks: one multi-tensor add
        self.mods = []
        training = False
        for r in roots:
            mods = getattr(r, "_xr_bn_mods", None)
            if mods is None:
                mods = [m for m in r.modules() if isinstance(m, _BNMixin) and m.track_running_stats
                        and m.num_batches_tracked is not None]
                r.__dict__["_xr_bn_mods"] = mods
            self.mods += mods
            training = training or r.training
        self.active = training and not _NBT_BATCHED[0] and len(self.mods) > 0

    def __enter__(self):
        if self.active:
            torch._foreach_add_([m.num_batches_tracked for m in self.mods if m.training], 1)
            _NBT_BATCHED[0] = True
        return self

    def __exit__(self, *exc):
        if self.active:
            _NBT_BATCHED[0] = False
        return False


class _BNMixin:
    def _count(self):
        if self.training and self.track_running_stats:
            # the training forward rewrites running_mean / running_var through raw pointers (xr_norm_finalize): no tensor
            # version changes, so the cached eval-mode coefficients are keyed on this counter as well
            self.__dict__["_xr_stat_epoch"] = self.__dict__.get("_xr_stat_epoch", 0) + 1
            log = ops._touch_log[0]
            if log is not None:
                import weakref
                log["bns"][id(self)] = weakref.ref(self)
            if self.num_batches_tracked is not None and not _NBT_BATCHED[0]:
                self.num_batches_tracked.add_(1)

    def _momentum(self):
        if self.momentum is None:
            # torch's cumulative moving average (factor 1 / num_batches_tracked) needs the device-side counter on the host;
            # the reference never uses it (every BatchNorm there keeps the default momentum 0.1)
            raise NotImplementedError("xrface BatchNorm: momentum=None (cumulative average) is not supported")
        return self.momentum

    def _eval_coef(self):
        """Eval mode: scale / shift depend only on the parameters and running statistics -- computed once per version of
        those four tensors (a frozen teacher or an inference backbone then launches nothing per forward for them)."""
        if not buf_is_cuda(self.running_mean):
            return None
        tensors = (self.weight, self.bias, self.running_mean, self.running_var)
        key = tuple((t.data_ptr(), t._version) if t is not None else None for t in tensors) + (ops._pack_epoch[0],
              self.weight.__dict__.get("_xr_epoch", 0) if self.weight is not None else 0,
              self.__dict__.get("_xr_stat_epoch", 0))
        hit = self.__dict__.get("_xr_eval_coef")
        if hit is None or hit[0] != key:
            hit = (key, ops.bn_eval_coeffs(self.weight, self.bias, self.running_mean, self.running_var, self.eps))
            self.__dict__["_xr_eval_coef"] = hit
        return hit[1]

    def f(self, buf, res=None, act=None, alpha=None, slink=None, offer_stats=False):
        training = self.training or not self.track_running_stats
        self._count()
        mom = self._momentum()
        coef = None if training or torch.is_grad_enabled() and _any_requires_grad(self) else self._eval_coef()
        return ops.norm_act(buf, self.weight, self.bias, self.running_mean, self.running_var, res, alpha, "bn", act, training,
                            mom, self.eps, slink if training else None, coef, offer_stats)



    def f_pass(self, buf, link=None, sub=1):
        """(bn(buf), buf'): buf' aliases buf and carries the identity-branch gradient into this norm's backward.
        link: an ops.BnLink shared with the one convolution that consumes bn(buf) (fused backward reduction).
        sub >= 2: buf' is buf sub-sampled by that stride (its gradient then comes back compact)."""
        training = self.training or not self.track_running_stats
        self._count()
        mom = self._momentum()
        return ops.norm_act_pass(buf, self.weight, self.bias, self.running_mean, self.running_var, "bn", None, training, mom,
                                 self.eps, link if training else None, ops.chain_of(buf), sub)


class BatchNorm2d(_BNMixin, nn.BatchNorm2d):
    def forward(self, x):
        return leave(self.f(enter(x)))


class BatchNorm1d(_BNMixin, nn.BatchNorm1d):
    def forward(self, x):
        assert x.dim() == 2
        return leave2d(self.f(enter(x)))


class PReLU(nn.PReLU):
    def f(self, buf, res=None):
        return ops.norm_act(buf, res=res, alpha=self.weight, mode="none", act="prelu")

    def forward(self, x):
        if x.dim() == 2:
            return leave2d(self.f(enter(x)))
        return leave(self.f(enter(x)))


class ReLU(nn.ReLU):
    def f(self, buf, res=None):
        return ops.norm_act(buf, res=res, mode="none", act="relu")

    def forward(self, x):
        if x.dim() == 2:
            return leave2d(self.f(enter(x)))
        return leave(self.f(enter(x)))


class Tanh(nn.Tanh):
    def f(self, buf):
        return ops.norm_act(buf, mode="none", act="tanh")

    def forward(self, x):
        return leave(self.f(enter(x)))


class ReflectionPad2d(nn.ReflectionPad2d):
    def f(self, buf):
        p = self.padding
        assert p[0] == p[1] == p[2] == p[3], "xrface.ReflectionPad2d: uniform padding only"
        return ops.reflect_pad(buf, p[0])

    def forward(self, x):
        return leave(self.f(enter(x)))


class Dropout(nn.Dropout):
    """Dropout(p) with a counter-based stream; ``inject_mask`` (uint8, NHWC layout of the input buffer) pins it
    for parity tests."""

    inject_mask = None

    def f(self, buf):
        return ops.dropout(buf, self.p, self.training, self.inject_mask)

    def forward(self, x):
        if x.dim() == 2:
            return leave2d(self.f(enter(x)))
        return leave(self.f(enter(x)))


class MaxPool2d(nn.MaxPool2d):
    """Only the two forms the hot path uses: MaxPool2d(1, stride) (pure sub-sampling, model_irse.py:53)
    and MaxPool2d(2, 2)."""

    def f(self, buf):
        k = self.kernel_size if isinstance(self.kernel_size, int) else self.kernel_size[0]
        s = self.stride if isinstance(self.stride, int) else self.stride[0]
        if k == 1:
            return ops.subsample(buf, s)
        if k == 2 and s == 2:
            return ops.maxpool2(buf)
        raise RuntimeError(f"xrface.MaxPool2d: unsupported kernel {k} / stride {s}")

    def forward(self, x):
        return leave(self.f(enter(x)))


class Flatten(nn.Module):
    """input.view(N, -1) in C,H,W order (model_irse.py:11-13); materialises the permutation when handed a
    channels_last view (the fused Backbone.forward never calls this -- it folds the order into the weight pack)."""

    def forward(self, x):
        return x.reshape(x.size(0), -1)


def run_seq(seq, buf):
    """Run an nn.Sequential of xrface layers on an NHWC buffer."""
    for m in seq:
        buf = m.f(buf)
    return buf


def res_trunk(seq, buf, times):
    """``times`` passes over an nn.Sequential of FSRNet residual blocks (model/FSRnet.py:331-333 applies the same three blocks three
    times).  bf16 64-channel blocks run as ONE chained op (ops._ResTrunk64: consecutive blocks hand the tail's pre-activation
    gradient to each other in the backward pass); anything else runs block by block."""
    blocks = list(seq)
    if (all(hasattr(b, a) for b in blocks for a in ("conv1", "in1", "relu", "conv2", "in2", "relu_out"))
            and all(b.in1.weight is not None and b.in2.weight is not None for b in blocks) and ops.res_trunk64_ok(buf, blocks)):
        return ops.res_trunk64(buf, blocks, times)
    for _ in range(times):
        buf = run_seq(seq, buf)
    return buf


def conv_bn(conv, bn, buf, res=None, act=None, alpha=None, pass_through=False, offer_stats=False):
    """conv -> BatchNorm(+residual, +activation) with the batch statistics taken in the convolution's epilogue.
    pass_through: also return buf' (aliasing buf) for an identity branch whose gradient the conv's dgrad epilogue sums in.
    offer_stats: the result opens a residual unit (ops.norm_act)."""
    link = ops.StatsLink() if (bn.training or not bn.track_running_stats) else None
    if pass_through:
        y, bufp = conv.f_pass(buf, link)
        return bn.f(y, res=res, act=act, alpha=alpha, slink=link), bufp
    return bn.f(conv.f(buf, link), res=res, act=act, alpha=alpha, slink=link, offer_stats=offer_stats)
