"""IR / IR-SE face-embedding backbones on the HIP path -- mirror of /root/reference
SUPER_RESOLUTION/model/model_irse.py (byte-identical to DISTILLATION/model/model_irse.py).

Keeps ``input_layer`` / ``body`` (an nn.Sequential of individually addressable blocks, iterated by
FeatureExtractor, GroupDepthConv.py:39-45) / ``output_layer`` and every state_dict key.
"""
from __future__ import annotations

from collections import namedtuple

import torch
import torch.nn as nn
from torch.nn import Module, Sequential

from .. import nn as xnn
from .. import ops
from ..ops import enter, leave, leave2d
from ..nn import batched_bn_counters

Conv2d, BatchNorm1d, BatchNorm2d, PReLU, ReLU, Dropout, MaxPool2d, Linear = (
    xnn.Conv2d, xnn.BatchNorm1d, xnn.BatchNorm2d, xnn.PReLU, xnn.ReLU, xnn.Dropout, xnn.MaxPool2d, xnn.Linear)
Flatten = xnn.Flatten


def l2_norm(input, axis=1):
    """reference model_irse.py:16-20 (host-side helper on small tensors)."""
    norm = torch.norm(input, 2, axis, True)
    return torch.div(input, norm)


class SEModule(Module):
    """reference model_irse.py:23-46.  Inside a block the squeeze/excite/scale/add is one fused op
    (ops.se_scale_add); ``forward`` keeps the standalone semantics x * sigmoid(fc2(relu(fc1(avgpool(x)))))."""

    def __init__(self, channels, reduction):
        super().__init__()
        self.avg_pool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = Conv2d(channels, channels // reduction, kernel_size=1, padding=0, bias=False)
        nn.init.xavier_uniform_(self.fc1.weight.data)
        self.relu = ReLU(inplace=True)
        self.fc2 = Conv2d(channels // reduction, channels, kernel_size=1, padding=0, bias=False)
        self.sigmoid = nn.Sigmoid()

    def f(self, buf, shortcut=None):
        return ops.se_scale_add(buf, self.fc1.weight, self.fc2.weight, shortcut)

    def forward(self, x):
        return leave(self.f(enter(x)))


class bottleneck_IR(Module):
    """reference model_irse.py:49-66."""

    def __init__(self, in_channel, depth, stride):
        super().__init__()
        if in_channel == depth:
            self.shortcut_layer = MaxPool2d(1, stride)
        else:
            self.shortcut_layer = Sequential(Conv2d(in_channel, depth, (1, 1), stride, bias=False), BatchNorm2d(depth))
        self.res_layer = Sequential(
            BatchNorm2d(in_channel),
            Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False), PReLU(depth),
            Conv2d(depth, depth, (3, 3), stride, 1, bias=False), BatchNorm2d(depth))

    def _conv_prelu_conv(self, b1, link=None):
        """(BN output) -> conv3x3 -> PReLU -> conv3x3(stride); the PReLU backward rides in the second conv's dgrad epilogue,
        the first BatchNorm's backward reductions in the first conv's (link)."""
        rl = self.res_layer
        c1, c2 = rl[1], rl[3]
        y1, p1 = ops.conv2d_prelu(b1, c1.weight, rl[2].weight, c1.bias, c1.stride[0], c1.padding[0], link)  # PReLU out of the epilogue
        return ops.prelu_conv2d(y1, rl[2].weight, c2.weight, c2.stride[0], c2.padding[0], p1)

    def _shortcut(self, x, presub=False):
        """presub: x already is the block input sub-sampled by the unit's stride (second output of the opening BatchNorm)."""
        if isinstance(self.shortcut_layer, Sequential):
            if presub:   # 1x1 / stride s on the full input == 1x1 / stride 1 on the sub-sampled input
                conv, bn = self.shortcut_layer[0], self.shortcut_layer[1]
                link = ops.StatsLink() if (bn.training or not bn.track_running_stats) else None
                return bn.f(ops.conv2d(x, conv.weight, conv.bias, 1, 0, link), slink=link)
            return xnn.conv_bn(self.shortcut_layer[0], self.shortcut_layer[1], x)   # BN statistics out of the conv epilogue
        return x if presub else self.shortcut_layer.f(x)

    def _sub_stride(self, x):
        """Stride by which the opening BatchNorm may hand the shortcut a sub-sampled input (0: it hands the input itself)."""
        sl = self.shortcut_layer
        s_ = sl[0].stride[0] if isinstance(sl, Sequential) else (sl.stride if isinstance(sl.stride, int) else sl.stride[0])
        if isinstance(sl, Sequential) and (sl[0].kernel_size != (1, 1) or sl[0].padding != (0, 0)):
            return 0
        ok = (ops._cfg["sub_pass"] and s_ >= 2 and x.shape[1] % s_ == 0 and x.shape[2] % s_ == 0 and self.training
              and torch.is_grad_enabled() and x.requires_grad)
        return s_ if ok else 0

    def f(self, x):
        # the block input feeds both the BN of the residual branch and the shortcut: route the shortcut through the
        # BN op's pass-through output so the two input gradients are summed inside its backward kernel
        link = ops.BnLink()
        sub = self._sub_stride(x)
        b1, xs = self.res_layer[0].f_pass(x, link, sub or 1)
        sc = self._shortcut(xs, presub=bool(sub))
        r = self._conv_prelu_conv(b1, link)
        return self.res_layer[4].f(r, res=sc)

    def forward(self, x):
        return leave(self.f(enter(x)))


class bottleneck_IR_SE(bottleneck_IR):
    """reference model_irse.py:69-91."""

    def __init__(self, in_channel, depth, stride):
        super().__init__(in_channel, depth, stride)
        self.res_layer = Sequential(
            BatchNorm2d(in_channel),
            Conv2d(in_channel, depth, (3, 3), (1, 1), 1, bias=False), PReLU(depth),
            Conv2d(depth, depth, (3, 3), stride, 1, bias=False), BatchNorm2d(depth),
            SEModule(depth, 16))

    def f(self, x):
        rl = self.res_layer
        if not isinstance(self.shortcut_layer, Sequential) and ops.ir_se_unit_ok(x, self) and self.shortcut_layer.stride in (1, (1, 1)):
            return ops.ir_se_unit(x, self)      # identity-shortcut unit: one block-level C call each way (xr_ir_block_fwd / bwd)
        link = ops.BnLink()
        sub = self._sub_stride(x)
        b1, xs = rl[0].f_pass(x, link, sub or 1)
        sc = self._shortcut(xs, presub=bool(sub))
        y2 = self._conv_prelu_conv(b1, link)
        return ops.bn_se_add(y2, rl[4], rl[5], sc)   # BatchNorm + SE + shortcut add in one elementwise pass


class Bottleneck(namedtuple('Block', ['in_channel', 'depth', 'stride'])):
    '''A named tuple describing a ResNet block.'''


def get_block(in_channel, depth, num_units, stride=2):
    return [Bottleneck(in_channel, depth, stride)] + [Bottleneck(depth, depth, 1) for _ in range(num_units - 1)]


def get_blocks(num_layers):
    if num_layers == 50:
        units = (3, 4, 14, 3)
    elif num_layers == 100:
        units = (3, 13, 30, 3)
    elif num_layers == 152:
        units = (3, 8, 36, 3)
    else:
        raise ValueError("num_layers should be 50, 100 or 152")
    return [get_block(64, 64, units[0]), get_block(64, 128, units[1]), get_block(128, 256, units[2]),
            get_block(256, 512, units[3])]


class Backbone(Module):
    """reference model_irse.py:129-189.  forward(x: N x 3 x 112 x 112) -> N x 512."""

    def __init__(self, input_size, num_layers, mode='ir'):
        super().__init__()
        assert input_size[0] in [112, 224], "input_size should be [112, 112] or [224, 224]"
        assert num_layers in [50, 100, 152], "num_layers should be 50, 100 or 152"
        assert mode in ['ir', 'ir_se'], "mode should be ir or ir_se"
        unit_module = bottleneck_IR if mode == 'ir' else bottleneck_IR_SE
        self.input_layer = Sequential(Conv2d(3, 64, (3, 3), 1, 1, bias=False), BatchNorm2d(64), PReLU(64))
        side = 7 if input_size[0] == 112 else 14
        self.output_layer = Sequential(BatchNorm2d(512), Dropout(), Flatten(), Linear(512 * side * side, 512),
                                       BatchNorm1d(512))
        modules = []
        for block in get_blocks(num_layers):
            for b in block:
                modules.append(unit_module(b.in_channel, b.depth, b.stride))
        self.body = Sequential(*modules)
        self._initialize_weights()

    # -- fused NHWC path ---------------------------------------------------------------------------
    def f_input(self, buf):
        il = self.input_layer
        return xnn.conv_bn(il[0], il[1], buf, act="prelu", alpha=il[2].weight, offer_stats=True)

    def f_output(self, buf):
        ol = self.output_layer
        y = ol[1].f(ol[0].f(buf))
        return ol[4].f(ol[3].f(y))

    def f(self, buf, taps=()):
        with batched_bn_counters(self):
            return self._f(buf, taps)

    def _f(self, buf, taps=()):
        y = self.f_input(buf)
        tapped = []
        for i, blk in enumerate(self.body):
            y = blk.f(y)
            if i in taps:
                tapped.append(y)
        return self.f_output(y), tapped

    def forward(self, x):
        emb, _ = self.f(enter(x))
        return leave2d(emb)

    def parameters_in_execution_order(self):
        """input_layer, body, output_layer -- the module registers output_layer BEFORE body (state_dict order of the
        reference, model_irse.py:139-166).  parallel.FlatParams laid out in this order makes every gradient bucket a run of
        layers that finish backward together, so the 51 MB Linear(25088, 512) bucket goes to RCCL at the START of backward."""
        return [*self.input_layer.parameters(), *self.body.parameters(), *self.output_layer.parameters()]

    def forward_taps(self, x, taps=(2, 6, 20, 23)):
        """(emb, tap_0, ..) -- the 5-output teacher distill_main.py:59 unpacks (taps after the body blocks
        that end the four stages)."""
        emb, tapped = self.f(enter(x), tuple(taps))
        return (leave2d(emb), *[leave(t) for t in tapped])

    def lockstep_plan(self, taps=(2, 6, 20, 23)):
        """(stages, tap positions, finish) for xrface.lockstep.run_lockstep: one stage per layer group of forward_taps."""
        stages = [self.f_input] + [blk.f for blk in self.body] + [self.f_output]
        tap_after = {i + 1 for i in taps}                       # stage index whose output is tapped (body block i = stage i + 1)
        finish = lambda emb, tapped: (leave2d(emb), *[leave(t) for t in tapped])
        return stages, tap_after, finish

    def _initialize_weights(self):
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.xavier_uniform_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()
            elif isinstance(m, (nn.BatchNorm2d, nn.BatchNorm1d)):
                m.weight.data.fill_(1)
                m.bias.data.zero_()
            elif isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight.data)
                if m.bias is not None:
                    m.bias.data.zero_()


def forward_taps_lockstep(nets, xs, taps=(2, 6, 20, 23)):
    """``[net.forward_taps(x, taps) for net, x in zip(nets, xs)]`` for independent backbones of the same depth (the student and
    the assistant of the residual-KD step), advanced block by block in lockstep on their own streams: see xrface/lockstep.py."""
    from ..lockstep import run_lockstep
    return run_lockstep(nets, xs, [n.lockstep_plan(taps) for n in nets])


def lockstep_join(device, n):
    from ..lockstep import join
    join(device, n)


class TeacherWithTaps(Module):
    """Wraps a Backbone so ``model(x)`` returns (emb, t1, t2, t3, t4) as distill_main.py:59 expects."""

    def __init__(self, backbone, taps=(2, 6, 20, 23)):
        super().__init__()
        self.backbone = backbone
        self.taps = tuple(taps)

    def forward(self, x):
        return self.backbone.forward_taps(x, self.taps)


def IR_50(input_size):
    return Backbone(input_size, 50, 'ir')


def IR_101(input_size):
    return Backbone(input_size, 100, 'ir')


def IR_152(input_size):
    return Backbone(input_size, 152, 'ir')


def IR_SE_50(input_size):
    return Backbone(input_size, 50, 'ir_se')


def IR_SE_101(input_size):
    return Backbone(input_size, 100, 'ir_se')


def IR_SE_152(input_size):
    return Backbone(input_size, 152, 'ir_se')
