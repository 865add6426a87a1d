"""BN+ReLU ResNet (student / assistant of the residual-KD step) on the HIP path -- mirror of
/root/reference model/resnet.py.  forward -> (emb, x1, x2, x3, x4)."""
from __future__ import annotations

import torch.nn as nn
from torch.nn import Module, Sequential

from .. import nn as xnn
from ..ops import enter, leave, leave2d

Conv2d, BatchNorm1d, BatchNorm2d, ReLU, Dropout, MaxPool2d, Linear = (
    xnn.Conv2d, xnn.BatchNorm1d, xnn.BatchNorm2d, xnn.ReLU, xnn.Dropout, xnn.MaxPool2d, xnn.Linear)

__all__ = ['ResNet', 'ResNet_34', 'BasicBlock']


def conv3x3(in_planes, out_planes, stride=1):
    return Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return Conv2d(in_planes, out_planes, kernel_size=1, stride=stride, bias=False)


class BasicBlock(Module):
    """reference model/resnet.py:18-47: relu(bn2(conv2(relu(bn1(conv1(x))))) + shortcut)."""
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = conv3x3(inplanes, planes, stride)
        self.bn1 = BatchNorm2d(planes)
        self.relu = ReLU(inplace=True)
        self.conv2 = conv3x3(planes, planes)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride

    def f(self, x):
        # BN statistics out of the conv epilogues; an identity residual goes through conv1's pass-through output so that its
        # gradient is summed in conv1's dgrad epilogue
        if self.downsample is None:
            o, res = xnn.conv_bn(self.conv1, self.bn1, x, act="relu", pass_through=True)
        else:
            o = xnn.conv_bn(self.conv1, self.bn1, x, act="relu")
            res = xnn.conv_bn(self.downsample[0], self.downsample[1], x)
        return xnn.conv_bn(self.conv2, self.bn2, o, res=res, act="relu")

    def forward(self, x):
        return leave(self.f(enter(x)))


class ResNet(Module):
    """reference model/resnet.py:152-225 (stem conv7x7 s2, max-pool skipped :212)."""

    def __init__(self, input_size, block, layers, zero_init_residual=True):
        super().__init__()
        assert input_size[0] in [112, 224], "input_size should be [112, 112] or [224, 224]"
        self.inplanes = 64
        self.conv1 = Conv2d(3, 64, kernel_size=7, stride=2, padding=3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.bn_o1 = BatchNorm2d(512)
        self.dropout = Dropout()
        self.fc = Linear(25088, 512) if input_size[0] == 112 else Linear(2048 * 8 * 8, 512)
        self.bn_o2 = BatchNorm1d(512)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = Sequential(conv1x1(self.inplanes, planes * block.expansion, stride),
                                    BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes))
        return Sequential(*layers)

    def f(self, buf):
        with xnn.batched_bn_counters(self):   # one multi-tensor add instead of one counter kernel per BatchNorm
            return self._f(buf)

    def _f(self, buf):
        y = xnn.conv_bn(self.conv1, self.bn1, buf, act="relu")
        feats = []
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            y = xnn.run_seq(layer, y)
            feats.append(y)
        z = self.bn_o2.f(self.fc.f(self.bn_o1.f(y)))
        return z, feats

    def forward(self, x):
        z, feats = self.f(enter(x))
        return (leave2d(z), *[leave(t) for t in feats])

    def lockstep_plan(self):
        """(stages, tap positions, finish) for xrface.lockstep.run_lockstep: stem, every residual block, head; the four stage
        outputs are tapped (the 5-tuple forward returns)."""
        stages = [lambda buf: xnn.conv_bn(self.conv1, self.bn1, buf, act="relu")]
        tap_after = set()
        for layer in (self.layer1, self.layer2, self.layer3, self.layer4):
            stages += [blk.f for blk in layer]
            tap_after.add(len(stages) - 1)
        stages.append(lambda y: self.bn_o2.f(self.fc.f(self.bn_o1.f(y))))
        finish = lambda z, feats: (leave2d(z), *[leave(t) for t in feats])
        return stages, tap_after, finish


def ResNet_34(input_size=[112, 112]):
    return ResNet(input_size, BasicBlock, [3, 4, 6, 3])
