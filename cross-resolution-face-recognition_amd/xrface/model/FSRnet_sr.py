"""SR-variant FSRNet generators on the HIP path -- mirror of /root/reference SUPER_RESOLUTION/model/FSRnet.py
(the variant SUPER_RESOLUTION/train_FHN.py:26,103-106 instantiates): CycleGAN-style generators built from
ReflectionPad2d + conv, stride-2 down / ConvTranspose2d(k3,s2,p1,op1) up, non-affine InstanceNorm + ReLU, Tanh
heads, and a 4 x depth-4 pre-activation-bottleneck hourglass prior net.  Same constructor signatures, forward
tuples and state_dict keys (``model.<i>.*`` / ``out.<i>.*`` Sequential indices included).
"""
from __future__ import annotations

import torch.nn as nn

from .. import nn as xnn
from .. import ops
from ..ops import enter, leave

__all__ = ["_Residual_Block", "Bottleneck", "Hourglass", "Coarse_SR_Network", "Fine_SR_Encoder",
           "Prior_Estimation_Network", "Fine_SR_Decoder"]


def run_fused(seq, buf):
    """Run an nn.Sequential on an NHWC buffer, fusing every (InstanceNorm2d, ReLU) pair into one elementwise pass."""
    mods = list(seq)
    i = 0
    while i < len(mods):
        m = mods[i]
        if isinstance(m, xnn.InstanceNorm2d) and i + 1 < len(mods) and isinstance(mods[i + 1], xnn.ReLU):
            buf = m.f(buf, act="relu")
            i += 2
        else:
            buf = m.f(buf)
            i += 1
    return buf


class _Residual_Block(nn.Module):
    """IN2(conv2(PReLU(IN1(conv1(x))))) + x -- NO trailing activation (reference SUPER_RESOLUTION/model/FSRnet.py:12-35)."""

    def __init__(self, out_channels, in_channels=64):
        super().__init__()
        self.conv1 = xnn.Conv2d(in_channels, out_channels, 3, 1, 1, bias=False)
        self.in1 = xnn.InstanceNorm2d(out_channels, affine=True)
        self.relu = xnn.PReLU(out_channels)
        self.conv2 = xnn.Conv2d(out_channels, out_channels, 3, 1, 1, bias=False)
        self.in2 = xnn.InstanceNorm2d(out_channels, affine=True)

    def f(self, x):
        y = self.in1.f(self.conv1.f(x), act="prelu", alpha=self.relu.weight)
        return self.in2.f(self.conv2.f(y), res=x)

    def forward(self, x):
        return leave(self.f(enter(x)))


class Bottleneck(nn.Module):
    """Pre-activation bottleneck, expansion 1, biased convs (reference :77-114)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.bn1 = xnn.InstanceNorm2d(inplanes)
        self.conv1 = xnn.Conv2d(inplanes, planes, kernel_size=1, bias=True)
        self.bn2 = xnn.InstanceNorm2d(planes)
        self.conv2 = xnn.Conv2d(planes, planes, kernel_size=3, stride=stride, padding=1, bias=True)
        self.bn3 = xnn.InstanceNorm2d(planes)
        self.conv3 = xnn.Conv2d(planes, planes, kernel_size=1, bias=True)
        self.relu = xnn.ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride

    def f(self, x):
        y = self.conv1.f(self.bn1.f(x, act="relu"))
        y = self.conv2.f(self.bn2.f(y, act="relu"))
        y = self.conv3.f(self.bn3.f(y, act="relu"))
        res = x if self.downsample is None else xnn.run_seq(self.downsample, x)
        return ops.add(y, res)

    def forward(self, x):
        return leave(self.f(enter(x)))


class Hourglass(nn.Module):
    """reference :117-156."""

    def __init__(self, block, num_blocks, planes, depth):
        super().__init__()
        self.depth = depth
        self.block = block
        self.hg = self._make_hour_glass(block, num_blocks, planes, depth)

    def _make_residual(self, block, num_blocks, planes):
        return nn.Sequential(*[block(planes * block.expansion, planes) for _ in range(num_blocks)])

    def _make_hour_glass(self, block, num_blocks, planes, depth):
        hg = []
        for i in range(depth):
            res = [self._make_residual(block, num_blocks, planes) for _ in range(3)]
            if i == 0:
                res.append(self._make_residual(block, num_blocks, planes))
            hg.append(nn.ModuleList(res))
        return nn.ModuleList(hg)

    def _hg(self, n, x):
        up1 = xnn.run_seq(self.hg[n - 1][0], x)
        low = xnn.run_seq(self.hg[n - 1][1], ops.maxpool2(x))
        low = self._hg(n - 1, low) if n > 1 else xnn.run_seq(self.hg[n - 1][3], low)
        low = xnn.run_seq(self.hg[n - 1][2], low)
        return ops.upadd2(up1, low)

    def f(self, x):
        return self._hg(self.depth, x)

    def forward(self, x):
        return leave(self.f(enter(x)))


def _up_tail(ngf):
    return [xnn.ConvTranspose2d(ngf * 4, ngf * 2, 3, 2, 1, 1, bias=False),
            xnn.ReflectionPad2d(1), xnn.Conv2d(ngf * 2, ngf * 2, 3, 1, 0, bias=False), xnn.InstanceNorm2d(ngf * 2), xnn.ReLU(True),
            xnn.ConvTranspose2d(ngf * 2, ngf, 3, 2, 1, 1, bias=False),
            xnn.ReflectionPad2d(1), xnn.Conv2d(ngf, ngf, 3, 1, 0, bias=False), xnn.InstanceNorm2d(ngf), xnn.ReLU(True)]


class Coarse_SR_Network(nn.Module):
    """reference :251-298.  forward(x) -> coarse_img (N,3,H,W) in [-1,1]."""

    def __init__(self, ngf=64, n_blocks=6):
        super().__init__()
        model = [xnn.ReflectionPad2d(3), xnn.Conv2d(3, ngf, 7, 1, 0, bias=False), xnn.InstanceNorm2d(ngf), xnn.ReLU(True),
                 xnn.ReflectionPad2d(1), xnn.Conv2d(ngf, ngf * 2, 3, 2, 0, bias=False),
                 xnn.ReflectionPad2d(1), xnn.Conv2d(ngf * 2, ngf * 2, 3, 1, 0, bias=False), xnn.InstanceNorm2d(ngf * 2), xnn.ReLU(True),
                 xnn.ReflectionPad2d(1), xnn.Conv2d(ngf * 2, ngf * 4, 3, 2, 0, bias=False),
                 xnn.ReflectionPad2d(1), xnn.Conv2d(ngf * 4, ngf * 4, 3, 1, 0, bias=False), xnn.InstanceNorm2d(ngf * 2), xnn.ReLU(True)]
        model += [_Residual_Block(out_channels=ngf * 4, in_channels=ngf * 4) for _ in range(n_blocks)]
        model += _up_tail(ngf)
        self.model = nn.Sequential(*model)
        self.out = nn.Sequential(xnn.ReflectionPad2d(1), xnn.Conv2d(ngf, 3, 3, 1, 0, bias=False), xnn.Tanh())

    def f(self, x):
        return run_fused(self.out, run_fused(self.model, x))

    def forward(self, x):
        return leave(self.f(enter(x)), 3)


class Fine_SR_Encoder(nn.Module):
    """reference :301-342.  forward(x) -> (N, ngf, H, W)."""

    def __init__(self, ngf=64, n_blocks=6):
        super().__init__()
        model = [xnn.ReflectionPad2d(1), xnn.Conv2d(3, ngf, 3, 1, 0, bias=False),
                 xnn.ReflectionPad2d(1), xnn.Conv2d(ngf, ngf * 2, 3, 2, 0, bias=False),
                 xnn.ReflectionPad2d(1), xnn.Conv2d(ngf * 2, ngf * 2, 3, 1, 0, bias=False), xnn.InstanceNorm2d(ngf * 2), xnn.ReLU(True),
                 xnn.ReflectionPad2d(1), xnn.Conv2d(ngf * 2, ngf * 4, 3, 2, 0, bias=False),
                 xnn.ReflectionPad2d(1), xnn.Conv2d(ngf * 4, ngf * 4, 3, 1, 0, bias=False), xnn.InstanceNorm2d(ngf * 2), xnn.ReLU(True)]
        model += [_Residual_Block(out_channels=ngf * 4, in_channels=ngf * 4) for _ in range(n_blocks)]
        model += _up_tail(ngf)
        self.model = nn.Sequential(*model)

    def f(self, x):
        return run_fused(self.model, x)

    def forward(self, x):
        return leave(self.f(enter(x)))


class Prior_Estimation_Network(nn.Module):
    """reference :345-370.  forward(x) -> (feat ngf, landmark num_landmark, parsing parsing_classes), all at H x W."""

    def __init__(self, n_hourglass=4, n_blocks=2, ngf=64, parsing_classes=13, num_landmark=68):
        super().__init__()
        self.fc = xnn.Conv2d(ngf, parsing_classes, kernel_size=1, bias=True)
        self.fc_landmark = xnn.Conv2d(ngf, num_landmark, kernel_size=1, bias=False)
        model = [xnn.ReflectionPad2d(3), xnn.Conv2d(3, ngf, 7, 1, 0, bias=False), xnn.InstanceNorm2d(ngf), xnn.ReLU(True)]
        model += [_Residual_Block(ngf) for _ in range(n_blocks)]
        model += [Hourglass(planes=ngf, depth=4, block=Bottleneck, num_blocks=3) for _ in range(n_hourglass)]
        self.model = nn.Sequential(*model)
        self._nc = (num_landmark, parsing_classes)

    def f(self, x):
        y = run_fused(self.model, x)
        return y, self.fc_landmark.f(y), self.fc.f(y)

    def forward(self, x):
        feat, lmk, par = self.f(enter(x))
        return leave(feat), leave(lmk, self._nc[0]), leave(par, self._nc[1])


class Fine_SR_Decoder(nn.Module):
    """reference :373-416.  forward(cat(prior_feat, enc_feat)) -> sr_img (N,3,H,W)."""

    def __init__(self, ngf=128, n_blocks=6):
        super().__init__()
        model = [xnn.ReflectionPad2d(1), xnn.Conv2d(ngf, ngf * 2, 3, 2, 0, bias=False),
                 xnn.ReflectionPad2d(1), xnn.Conv2d(ngf * 2, ngf * 2, 3, 1, 0, bias=False), xnn.InstanceNorm2d(ngf * 2), xnn.ReLU(True),
                 xnn.ReflectionPad2d(1), xnn.Conv2d(ngf * 2, ngf * 4, 3, 2, 0, bias=False),
                 xnn.ReflectionPad2d(1), xnn.Conv2d(ngf * 4, ngf * 4, 3, 1, 0, bias=False), xnn.InstanceNorm2d(ngf * 4), xnn.ReLU(True)]
        model += [_Residual_Block(out_channels=ngf * 4, in_channels=ngf * 4) for _ in range(n_blocks)]
        model += _up_tail(ngf)
        self.model = nn.Sequential(*model)
        self.out = nn.Sequential(xnn.ReflectionPad2d(1), xnn.Conv2d(ngf, 3, 3, 1, 0, bias=False), xnn.Tanh())

    def f(self, x):
        return run_fused(self.out, run_fused(self.model, x))

    def forward(self, x):
        return leave(self.f(enter(x)), 3)
