"""ResNet backbones with the torchvision state_dict layout, running on the MI355X kernels.

Drop-in for the reference's ``uda/model/resnet.py`` (:16-43 ResNet without avgpool/fc in forward,
``out_features``; :50-59 ``_resnet``; :62-183 constructors, incl. the ResNeXt / Wide-ResNet ones).  The reference subclasses
``torchvision.models.ResNet``; torchvision is not a dependency here, so the v1.5 block structure
(stride on the 3x3, 1x1-conv+BN downsample, bias-free convs) is built from mi355.nn layers under the
same attribute names: ``conv1, bn1, layer{1-4}.{i}.{conv1-3,bn1-3,downsample.0/1}, fc``.

``pretrained=True`` in the reference downloads ImageNet weights (resnet.py:52-55).  There is no network
here: weights are read from ``$MI355_PRETRAINED_DIR/<arch>.pth`` (a torchvision state_dict) when that
file exists, otherwise initialisation stays random and a warning is printed.
"""
import copy
import os
import warnings

import torch
import torch.nn as nn

from mi355.nn import Conv2d, BatchNorm2d, ReLU, MaxPool2d, FusedSequential, link_conv_bn

__all__ = ['ResNet', 'resnet18', 'resnet34', 'resnet50', 'resnet101', 'resnet152', 'resnext50_32x4d', 'resnext101_32x8d',
           'wide_resnet50_2', 'wide_resnet101_2']


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64):
        super().__init__()
        if groups != 1 or base_width != 64:
            raise ValueError('BasicBlock only supports groups=1 and base_width=64')
        self.conv1 = Conv2d(inplanes, planes, 3, stride, 1, bias=False)
        self.bn1 = BatchNorm2d(planes)
        self.relu = ReLU(inplace=True)
        self.conv2 = Conv2d(planes, planes, 3, 1, 1, bias=False)
        self.bn2 = BatchNorm2d(planes)
        self.downsample = downsample
        self.stride = stride
        link_conv_bn(self)

    def forward(self, x):
        out, skip = self.conv1.forward_skip(x)     # skip aliases x; its gradient is summed inside conv1's dgrad
        identity = skip if self.downsample is None else self.downsample(skip)
        out = self.bn1(out, relu=True)
        return self.bn2(self.conv2(out), residual=identity, relu=True)   # BN + add + ReLU in one kernel


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inplanes, planes, stride=1, downsample=None, groups=1, base_width=64):
        super().__init__()
        width = int(planes * (base_width / 64.)) * groups      # ResNeXt (groups = 32) / Wide ResNet (base_width = 128) bottleneck width
        self.conv1 = Conv2d(inplanes, width, 1, 1, 0, bias=False)
        self.bn1 = BatchNorm2d(width)
        self.conv2 = Conv2d(width, width, 3, stride, 1, bias=False, groups=groups)
        self.bn2 = BatchNorm2d(width)
        self.conv3 = Conv2d(width, planes * 4, 1, 1, 0, bias=False)
        self.bn3 = BatchNorm2d(planes * 4)
        self.relu = ReLU(inplace=True)
        self.downsample = downsample
        self.stride = stride
        link_conv_bn(self)

    def forward(self, x):
        out, skip = self.conv1.forward_skip(x)     # skip aliases x; its gradient is summed inside conv1's dgrad
        identity = skip if self.downsample is None else self.downsample(skip)
        out = self.bn1(out, relu=True)
        out = self.bn2(self.conv2(out), relu=True)
        return self.bn3(self.conv3(out), residual=identity, relu=True)


class ResNet(nn.Module):
    """ResNets without fully connected layer (reference resnet.py:16-43)."""

    def __init__(self, block, layers, num_classes=1000, groups=1, width_per_group=64):
        super().__init__()
        self.inplanes = 64
        self.groups, self.base_width = groups, width_per_group
        self.conv1 = Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = BatchNorm2d(64)
        self.relu = ReLU(inplace=True)
        self.maxpool = MaxPool2d(3, 2, 1)
        self.layer1 = self._make_layer(block, 64, layers[0])
        self.layer2 = self._make_layer(block, 128, layers[1], stride=2)
        self.layer3 = self._make_layer(block, 256, layers[2], stride=2)
        self.layer4 = self._make_layer(block, 512, layers[3], stride=2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))            # kept for attribute parity; never run
        self.fc = nn.Linear(512 * block.expansion, num_classes)  # kept for state_dict parity; never run
        for m in self.modules():
            if isinstance(m, Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._out_features = self.fc.in_features
        link_conv_bn(self)

    def _make_layer(self, block, planes, blocks, stride=1):
        downsample = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            downsample = FusedSequential(Conv2d(self.inplanes, planes * block.expansion, 1, stride, 0, bias=False),
                                         BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, downsample, self.groups, self.base_width)]
        self.inplanes = planes * block.expansion
        for _ in range(1, blocks):
            layers.append(block(self.inplanes, planes, groups=self.groups, base_width=self.base_width))
        return nn.Sequential(*layers)

    def forward(self, x):
        x = self.bn1.forward_relu_maxpool(self.conv1(x), self.maxpool)     # (training: one fused pass over the conv output)
        x = self.layer1(x)
        x = self.layer2(x)
        x = self.layer3(x)
        x = self.layer4(x)
        return x

    @property
    def out_features(self) -> int:
        """The dimension of output features"""
        return self._out_features

    def copy_head(self) -> nn.Module:
        return copy.deepcopy(self.fc)


def _load_local_pretrained(model, arch):
    root = os.environ.get('MI355_PRETRAINED_DIR', 'models')
    path = os.path.join(root, arch + '.pth')
    if not os.path.exists(path):
        warnings.warn('pretrained=True but %s not found (no network access): %s keeps its random initialisation. '
                      'Put a torchvision %s state_dict there or set MI355_PRETRAINED_DIR.' % (path, arch, arch))
        return
    model_dict = model.state_dict()
    pretrained_dict = torch.load(path, map_location='cpu')
    pretrained_dict = {k: v for k, v in pretrained_dict.items() if k in model_dict}
    model.load_state_dict(pretrained_dict, strict=False)


def _resnet(arch, block, layers, pretrained, progress, **kwargs):
    model = ResNet(block, layers, **kwargs)
    if pretrained:
        _load_local_pretrained(model, arch)
    return model


def resnet18(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet18', BasicBlock, [2, 2, 2, 2], pretrained, progress, **kwargs)


def resnet34(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet34', BasicBlock, [3, 4, 6, 3], pretrained, progress, **kwargs)


def resnet50(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet50', Bottleneck, [3, 4, 6, 3], pretrained, progress, **kwargs)


def resnet101(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet101', Bottleneck, [3, 4, 23, 3], pretrained, progress, **kwargs)


def resnet152(pretrained=False, progress=True, **kwargs):
    return _resnet('resnet152', Bottleneck, [3, 8, 36, 3], pretrained, progress, **kwargs)


def resnext50_32x4d(pretrained=False, progress=True, **kwargs):
    """ResNeXt-50 32x4d (reference resnet.py:124-135): 32 groups of width 4 in every bottleneck's 3x3 conv."""
    return _resnet('resnext50_32x4d', Bottleneck, [3, 4, 6, 3], pretrained, progress, **dict(kwargs, groups=32, width_per_group=4))


def resnext101_32x8d(pretrained=False, progress=True, **kwargs):
    """ResNeXt-101 32x8d (reference resnet.py:138-149)."""
    return _resnet('resnext101_32x8d', Bottleneck, [3, 4, 23, 3], pretrained, progress, **dict(kwargs, groups=32, width_per_group=8))


def wide_resnet50_2(pretrained=False, progress=True, **kwargs):
    """Wide ResNet-50-2 (reference resnet.py:152-166): bottleneck width doubled, outer 1x1 widths unchanged."""
    return _resnet('wide_resnet50_2', Bottleneck, [3, 4, 6, 3], pretrained, progress, **dict(kwargs, width_per_group=128))


def wide_resnet101_2(pretrained=False, progress=True, **kwargs):
    """Wide ResNet-101-2 (reference resnet.py:169-183)."""
    return _resnet('wide_resnet101_2', Bottleneck, [3, 4, 23, 3], pretrained, progress, **dict(kwargs, width_per_group=128))
