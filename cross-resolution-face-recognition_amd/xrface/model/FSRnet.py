"""FSRNet face-hallucination sub-nets on the HIP path -- mirror of /root/reference model/FSRnet.py.

Same class names, constructor signatures, attribute names (per-sub-network optimizers address them,
Face_Hallucination_sub_Net.py:120-124), forward tuples and state_dict keys, including the registered-but-
unused parameters (bn_end, dropout, residual_next, instance_norm).  Forward bodies
run fused HIP ops on NHWC buffers: conv -> (InstanceNorm + PReLU [+ residual]) in one elementwise pass.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import nn as xnn
from .. import ops
from ..ops import enter, leave, leave2d

__all__ = ["_Residual_Block", "conv3x3", "BasicBlock", "Hourglass", "Course_SR_Network", "Fine_SR_Encoder",
           "Prior_Estimation_Network", "Fine_SR_Decoder", "Discriminator", "OverallNetwork", "OverallNetwork_GAN"]


class _Residual_Block(nn.Module):
    """PReLU_out(IN2(conv2(PReLU(IN1(conv1(x))))) + x) -- reference model/FSRnet.py:75-98."""

    def __init__(self, out_channels, in_channels=64):
        super().__init__()
        self.conv1 = xnn.Conv2d(in_channels, out_channels, 3, 1, 1, bias=False)
        self.in1 = xnn.InstanceNorm2d(out_channels, affine=True)
        self.relu = xnn.PReLU(out_channels)
        self.conv2 = xnn.Conv2d(out_channels, out_channels, 3, 1, 1, bias=False)
        self.in2 = xnn.InstanceNorm2d(out_channels, affine=True)
        self.relu_out = xnn.PReLU(out_channels)

    def f(self, x):
        if ops.direct64_ok(x, self.conv1.weight, 1, 1) and tuple(self.conv2.weight.shape) == (64, 64, 3, 3):
            # bf16, 64 channels: the whole block as one op on the direct convolution kernel (statistics in the conv
            # epilogues, IN1 + PReLU applied on conv2's load)
            return ops.resblock64(x, self.conv1, self.in1, self.relu, self.conv2, self.in2, self.relu_out)
        # the block input feeds conv1 and the residual add: route the residual through conv1's pass-through output so the
        # two gradients of x meet in conv1's dgrad epilogue instead of in a separate elementwise add
        c1, xs = self.conv1.f_pass(x)
        y = self.in1.f(c1, act="prelu", alpha=self.relu.weight)
        return self.in2.f(self.conv2.f(y), res=xs, act="prelu", alpha=self.relu_out.weight)

    def forward(self, x):
        return leave(self.f(enter(x)))


def conv3x3(in_planes, out_planes, stride=1):
    return xnn.Conv2d(in_planes, out_planes, kernel_size=3, stride=stride, padding=1, bias=False)


class BasicBlock(nn.Module):
    """Hourglass block (reference model/FSRnet.py:105-135): 128->128 hard-coded first conv, non-affine
    InstanceNorm, ONE PReLU used at both activation sites."""
    expansion = 2

    def __init__(self, inplanes=128, planes=128, stride=1, downsample=None):
        super().__init__()
        self.conv1 = xnn.Conv2d(128, 128, kernel_size=3, stride=stride, padding=1, bias=False)
        self.bn1 = xnn.InstanceNorm2d(planes * 2)
        self.relu = xnn.PReLU(128)
        self.conv2 = conv3x3(planes * 2, planes * 2)
        self.bn2 = xnn.InstanceNorm2d(planes * 2)
        self.downsample = downsample
        self.stride = stride

    def f(self, x):
        if self.downsample is None:
            c1, res = self.conv1.f_pass(x)      # residual gradient summed in conv1's dgrad epilogue
        else:
            c1, res = self.conv1.f(x), self.downsample.f(x)
        y = self.bn1.f(c1, act="prelu", alpha=self.relu.weight)
        return self.bn2.f(self.conv2.f(y), res=res, act="prelu", alpha=self.relu.weight)

    def forward(self, x):
        return leave(self.f(enter(x)))


class Hourglass(nn.Module):
    """Recursive hourglass (reference model/FSRnet.py:176-215): max-pool down, nearest x2 up, add."""

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


def _make_layer(block, n, out_channel, in_channel=64):
    return nn.Sequential(*[block(out_channel, in_channels=in_channel) for _ in range(n)])


class Course_SR_Network(nn.Module):
    """reference model/FSRnet.py:308-340.  forward -> (feat64, coarse_img)."""

    def __init__(self):
        super().__init__()
        self.conv_input = xnn.Conv2d(3, 64, 3, 1, 1, bias=True)
        self.relu = xnn.PReLU(64)
        self.residual = self.make_layer(_Residual_Block, 3, out_channel=64)
        self.dropout = nn.Dropout2d(p=0.5, inplace=True)
        self.conv_mid = xnn.Conv2d(64, 3, 3, 1, 1, bias=True)
        self.bn_mid = xnn.InstanceNorm2d(64, affine=True)
        self.bn_end = xnn.InstanceNorm2d(3, affine=True)

    def make_layer(self, block, num_of_layer, out_channel):
        return _make_layer(block, num_of_layer, out_channel)

    def f(self, x):
        y = self.bn_mid.f(self.conv_input.f(x), act="prelu", alpha=self.relu.weight)
        y = xnn.res_trunk(self.residual, y, 3)   # the SAME three blocks applied three times (model/FSRnet.py:331-333)
        y = self.bn_mid.f(y)
        return y, self.conv_mid.f(y)

    def forward(self, x):
        feat, img = self.f(enter(x))
        return leave(feat), leave(img, 3)


class Fine_SR_Encoder(Course_SR_Network):
    """reference model/FSRnet.py:342-379 (inherits the unused conv_mid / bn_end / dropout)."""

    def __init__(self):
        super().__init__()
        self.conv_input = xnn.Conv2d(3, 64, kernel_size=7, stride=4, padding=3, bias=True)
        self.relu = xnn.PReLU(64)
        self.bn_mid = xnn.InstanceNorm2d(64, affine=True)
        self.residual = self.make_layer(_Residual_Block, 3, out_channel=64)
        self.conv_end = xnn.Conv2d(64, 64, 3, 1, 1, bias=True)

    def f(self, x):
        y = self.bn_mid.f(self.conv_input.f(x), act="prelu", alpha=self.relu.weight)
        y = xnn.res_trunk(self.residual, y, 3)
        return self.bn_mid.f(self.conv_end.f(y), act="prelu", alpha=self.relu.weight)

    def forward(self, x):
        return leave(self.f(enter(x)))


class Prior_Estimation_Network(nn.Module):
    """reference model/FSRnet.py:381-426.  forward -> (feat128, landmark97, parsing11)."""

    def __init__(self):
        super().__init__()
        self.conv = xnn.Conv2d(3, 128, kernel_size=7, stride=4, padding=3, bias=True)
        self.bn = xnn.InstanceNorm2d(128, affine=True)
        self.relu = xnn.PReLU(128)
        self.residual = self.make_layer(_Residual_Block, 3, out_channel=128, in_channel=128)
        self.residual_next = self.make_layer(_Residual_Block, 3, out_channel=128, in_channel=128)
        self.hg = Hourglass(planes=64, depth=2, block=BasicBlock, num_blocks=2)
        self.dropout = nn.Dropout2d(p=0.5, inplace=True)
        self.fc = xnn.Conv2d(128, 11, kernel_size=1, bias=True)
        self.fc_landmark = xnn.Conv2d(128, 97, kernel_size=1, bias=True)

    def make_layer(self, block, num_of_layer, in_channel, out_channel):
        return _make_layer(block, num_of_layer, out_channel, in_channel)

    def f(self, x):
        y = self.bn.f(self.conv.f(x), act="prelu", alpha=self.relu.weight)
        y = xnn.run_seq(self.residual, y)
        y = self.hg.f(y)
        return y, self.fc_landmark.f(y), self.fc.f(y)

    def forward(self, x):
        feat, lmk, par = self.f(enter(x))
        return leave(feat), leave(lmk, 97), leave(par, 11)


class Fine_SR_Decoder(nn.Module):
    """reference model/FSRnet.py:428-459 (192-ch concat in, deconv k7 s4 p2 op1, shared trunk x3)."""

    def __init__(self):
        super().__init__()
        self.conv_input = xnn.Conv2d(192, 64, 3, 1, 1, bias=True)
        self.relu = xnn.PReLU(64)
        self.bn_mid = xnn.InstanceNorm2d(64, affine=True)
        self.deconv = xnn.ConvTranspose2d(64, 64, kernel_size=7, stride=4, bias=True, padding=2, output_padding=1)
        self.residual = self.make_layer(_Residual_Block, 3, out_channel=64)
        self.dropout = nn.Dropout2d(p=0.5, inplace=True)
        self.conv_out = xnn.Conv2d(64, 3, 3, 1, 1, bias=True)
        self.instance_norm = xnn.InstanceNorm2d(3, affine=True)

    def make_layer(self, block, num_of_layer, out_channel):
        return _make_layer(block, num_of_layer, out_channel)

    def f(self, x):
        y = self.bn_mid.f(self.conv_input.f(x), act="prelu", alpha=self.relu.weight)
        y = self.bn_mid.f(self.deconv.f(y), act="prelu", alpha=self.relu.weight)
        y = xnn.res_trunk(self.residual, y, 3)
        return self.conv_out.f(self.bn_mid.f(y))

    def forward(self, x):
        return leave(self.f(enter(x)), 3)


class Discriminator(nn.Module):
    """reference model/FSRnet.py:461-485 (only meaningful at 224x224: Linear(64*56*56, 512))."""

    def __init__(self):
        super().__init__()
        self.conv_input = xnn.Conv2d(192, 64, 3, 1, 1, bias=True)
        self.relu = xnn.PReLU(64)
        self.bn_mid = xnn.BatchNorm2d(64, affine=True)
        self.residual = _make_layer(_Residual_Block, 3, 64, 64)
        self.fc = xnn.Linear(64 * 56 * 56, 512)
        self.bn_end = xnn.BatchNorm1d(512)

    def f(self, x):
        y = self.bn_mid.f(self.conv_input.f(x), act="prelu", alpha=self.relu.weight)
        y = self.bn_mid.f(y)
        return self.bn_end.f(self.fc.f(y))

    def forward(self, x):
        return leave2d(self.f(enter(x)))


class OverallNetwork_GAN(nn.Module):
    """reference model/FSRnet.py:510-545.  forward(lr, hr) -> (sr, coarse, landmark, parsing, emb1, emb2)."""

    def __init__(self):
        super().__init__()
        self._coarse_sr_network = Course_SR_Network()
        self._prior_estimation_network = Prior_Estimation_Network()
        self._fine_sr_encoder = Fine_SR_Encoder()
        self._fine_sr_decoder = Fine_SR_Decoder()
        self._discriminator = Discriminator()

    def _once(self, x):
        enc = self._fine_sr_encoder.f(x)
        pe, lmk, par = self._prior_estimation_network.f(x)
        cat = ops.cat2(pe, enc)
        return cat, lmk, par, self._discriminator.f(cat)

    def forward_once(self, x):
        cat, lmk, par, emb = self._once(enter(x))
        return leave(cat), leave(lmk, 97), leave(par, 11), leave2d(emb)

    def forward(self, lr, hr):
        _, coarse = self._coarse_sr_network.f(enter(lr))
        cat1, lmk1, par1, emb1 = self._once(coarse)
        _, _, _, emb2 = self._once(enter(hr))
        sr = self._fine_sr_decoder.f(cat1)
        return leave(sr, 3), leave(coarse, 3), leave(lmk1, 97), leave(par1, 11), leave2d(emb1), leave2d(emb2)


class OverallNetwork(nn.Module):
    """reference model/FSRnet.py:488-508.  The reference forward feeds the 64-channel feature map into
    3-channel-input convs and raises; this mirror keeps the constructor / attribute
    surface and composes the generators the way SUPER_RESOLUTION/train_FHN.py:274-279 does:
    coarse_img -> {prior, encoder} -> cat -> decoder.  forward(x) -> (coarse, sr, landmark, parsing)."""

    def __init__(self):
        super().__init__()
        self._coarse_sr_network = Course_SR_Network()
        self._prior_estimation_network = Prior_Estimation_Network()
        self._fine_sr_encoder = Fine_SR_Encoder()
        self._fine_sr_decoder = Fine_SR_Decoder()
        self.softmax = nn.Softmax()

    def forward(self, x):
        _, coarse = self._coarse_sr_network.f(enter(x))
        enc = self._fine_sr_encoder.f(coarse)
        pe, lmk, par = self._prior_estimation_network.f(coarse)
        sr = self._fine_sr_decoder.f(ops.cat2(pe, enc))
        return leave(coarse, 3), leave(sr, 3), leave(lmk, 97), leave(par, 11)
