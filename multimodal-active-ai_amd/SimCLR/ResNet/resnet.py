"""Drop-in for the reference's SimCLR/ResNet/resnet.py on MI355X.

Same public names, constructor arguments, attribute names and state_dict keys
as the reference (resnet.py:31-343): the modules below only HOLD
parameters and buffers in the reference layout — ``ResNet.forward`` hands the
whole stack to the HIP engine (maai_hip.engine.backbone_forward: implicit-GEMM
MFMA convolutions with BatchNorm statistics in the epilogue, fused
BN/ReLU/residual passes, hand-written backward).  Semantics kept from the
reference: 7x7 stride-1 stem over 3*crop_measures channels, max-pool and
avg-pool never applied, no fc, output = layer4 map in NCHW fp32.
"""
import os
import sys

import torch.nn as nn

try:
    import maai_hip  # noqa: F401
except ImportError:  # locate the package next to this SimCLR/ tree or via MAAI_AMD_HOME
    _h = os.path.dirname(os.path.abspath(__file__))
    for _c in (os.environ.get("MAAI_AMD_HOME", ""), os.path.join(_h, "..", ".."), os.path.join(_h, "..", "..", "multimodal-active-ai_amd")):
        if _c and os.path.isdir(os.path.join(_c, "maai_hip")):
            sys.path.insert(0, os.path.abspath(_c))
            break
    import maai_hip  # noqa: F401
from maai_hip import engine as _engine

__all__ = ['ResNet', 'resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152', 'resnext50_32x4d',
           'resnext101_32x8d', 'wide_resnet50_2', 'wide_resnet101_2']


def conv3x3(in_planes, out_planes, stride=1, groups=1, dilation=1):
    return nn.Conv2d(in_planes, out_planes, 3, stride, dilation, dilation, groups, bias=False)


def conv1x1(in_planes, out_planes, stride=1):
    return nn.Conv2d(in_planes, out_planes, 1, stride, bias=False)


class _Block(nn.Module):
    """Parameter container; ``ResNet.forward`` hands the whole stack to the engine, which walks
    .conv*/.bn*/.downsample/.stride.  Called on its own (a consumer iterating ``layerN``), a block runs as one
    engine call: NCHW fp32 in, NCHW fp32 out, differentiable."""

    def forward(self, x):
        return _engine.block_forward(self, x)


class BasicBlock(_Block):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        if groups != 1 or base_width != 64:
            raise ValueError('BasicBlock only supports groups=1 and base_width=64')
        if dilation > 1:
            raise NotImplementedError("Dilation > 1 not supported in BasicBlock")
        self.conv1, self.bn1 = conv3x3(inplanes, planes, stride), norm_layer(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2, self.bn2 = conv3x3(planes, planes), norm_layer(planes)
        self.downsample, self.stride = downsample, stride


class Bottleneck(_Block):
    expansion = 4  # v1.5: the 3x3 carries the stride

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64, dilation=1, norm_layer=None):
        super().__init__()
        norm_layer = norm_layer or nn.BatchNorm2d
        width = int(planes * (base_width / 64.)) * groups
        self.conv1, self.bn1 = conv1x1(inplanes, width), norm_layer(width)
        self.conv2, self.bn2 = conv3x3(width, width, stride, groups, dilation), norm_layer(width)
        self.conv3, self.bn3 = conv1x1(width, planes * self.expansion), norm_layer(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample, self.stride = downsample, stride


class ResNet(nn.Module):
    def __init__(self, block, layers, zero_init_residual=False, groups=1, width_per_group=64, crop_measures=4,
                 replace_stride_with_dilation=None, norm_layer=None):
        super().__init__()
        self._norm_layer = norm_layer = norm_layer or nn.BatchNorm2d
        self.inplanes, self.dilation = 64, 1
        rswd = [False, False, False] if replace_stride_with_dilation is None else replace_stride_with_dilation
        if len(rswd) != 3:
            raise ValueError("replace_stride_with_dilation should be None or a 3-element tuple, got {}".format(rswd))
        self.groups, self.base_width, self.crop_measures = groups, width_per_group, crop_measures
        self.conv1 = nn.Conv2d(3 * crop_measures, 64, kernel_size=7, stride=1, padding=3, bias=False)
        self.bn1 = norm_layer(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(kernel_size=3, stride=2, padding=1)   # present, never applied (reference quirk)
        for i, (planes, n) in enumerate(zip((64, 128, 256, 512), layers)):
            setattr(self, "layer%d" % (i + 1), self._make_layer(block, planes, n, stride=1 if i == 0 else 2,
                                                                dilate=False if i == 0 else rswd[i - 1]))
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))                        # present, never applied
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, (nn.BatchNorm2d, nn.GroupNorm)):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        if zero_init_residual:
            for m in self.modules():
                if isinstance(m, Bottleneck):
                    nn.init.constant_(m.bn3.weight, 0)
                elif isinstance(m, BasicBlock):
                    nn.init.constant_(m.bn2.weight, 0)

    def _make_layer(self, block, planes, blocks, stride=1, dilate=False):
        prev_dil = self.dilation
        if dilate:
            self.dilation *= stride
            stride = 1
        out_ch = planes * block.expansion
        ds = None
        if stride != 1 or self.inplanes != out_ch:
            ds = nn.Sequential(conv1x1(self.inplanes, out_ch, stride), self._norm_layer(out_ch))
        seq = [block(self.inplanes, planes, stride, ds, self.groups, self.base_width, prev_dil, self._norm_layer)]
        self.inplanes = out_ch
        seq += [block(out_ch, planes, groups=self.groups, base_width=self.base_width, dilation=self.dilation,
                      norm_layer=self._norm_layer) for _ in range(1, blocks)]
        return nn.Sequential(*seq)

    def _forward_impl(self, x):
        return _engine.backbone_forward(self, x)

    def forward(self, x):
        return self._forward_impl(x)


def _resnet(arch, block, layers, **kwargs):
    return ResNet(block, layers, **kwargs)


def resnet18(**kw):
    return _resnet('resnet18', BasicBlock, [2, 2, 2, 2], **kw)


def resnet34(**kw):
    return _resnet('resnet34', BasicBlock, [3, 4, 6, 3], **kw)


def resnet50(**kw):
    return _resnet('resnet50', Bottleneck, [3, 4, 6, 3], **kw)


def resnet101(**kw):
    return _resnet('resnet101', Bottleneck, [3, 4, 23, 3], **kw)


def resnet152(**kw):
    return _resnet('resnet152', Bottleneck, [3, 8, 36, 3], **kw)


def resnext50_32x4d(**kw):
    kw.update(groups=32, width_per_group=4)
    return _resnet('resnext50_32x4d', Bottleneck, [3, 4, 6, 3], **kw)


def resnext101_32x8d(**kw):
    kw.update(groups=32, width_per_group=8)
    return _resnet('resnext101_32x8d', Bottleneck, [3, 4, 23, 3], **kw)


def wide_resnet50_2(**kw):
    kw['width_per_group'] = 128
    return _resnet('wide_resnet50_2', Bottleneck, [3, 4, 6, 3], **kw)


def wide_resnet101_2(**kw):
    kw['width_per_group'] = 128
    return _resnet('wide_resnet101_2', Bottleneck, [3, 4, 23, 3], **kw)
