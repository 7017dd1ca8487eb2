"""torchvision-layout ResNet (v1.5) without avgpool/fc in forward.

Follows reference ``uda/model/resnet.py:16-43`` (forward override, out_features)
and ``:62-107`` (resnet18/34/50/101 block lists).  Block structure is the
published torchvision one (BasicBlock / Bottleneck, stride on the 3x3,
1x1-conv+BN downsample, bias-free convs, BN eps 1e-5 momentum 0.1,
kaiming-normal fan_out init).  ``fc`` is kept so state_dict keys match.
"""
import torch
import torch.nn as nn


def _c3(i, o, s=1):
    return nn.Conv2d(i, o, 3, s, 1, bias=False)


def _c1(i, o, s=1):
    return nn.Conv2d(i, o, 1, s, 0, bias=False)


class BasicBlock(nn.Module):
    expansion = 1

    def __init__(self, inp, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _c3(inp, planes, stride)
        self.bn1 = nn.BatchNorm2d(planes)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = _c3(planes, planes)
        self.bn2 = nn.BatchNorm2d(planes)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return self.relu(out + idt)


class Bottleneck(nn.Module):
    expansion = 4

    def __init__(self, inp, planes, stride=1, downsample=None):
        super().__init__()
        self.conv1 = _c1(inp, planes)
        self.bn1 = nn.BatchNorm2d(planes)
        self.conv2 = _c3(planes, planes, stride)
        self.bn2 = nn.BatchNorm2d(planes)
        self.conv3 = _c1(planes, planes * 4)
        self.bn3 = nn.BatchNorm2d(planes * 4)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = downsample

    def forward(self, x):
        idt = x if self.downsample is None else self.downsample(x)
        out = self.relu(self.bn1(self.conv1(x)))
        out = self.relu(self.bn2(self.conv2(out)))
        out = self.bn3(self.conv3(out))
        return self.relu(out + idt)


class ResNet(nn.Module):
    def __init__(self, block, layers, num_classes=1000):
        super().__init__()
        self.inplanes = 64
        self.conv1 = nn.Conv2d(3, 64, 7, 2, 3, bias=False)
        self.bn1 = nn.BatchNorm2d(64)
        self.relu = nn.ReLU(inplace=True)
        self.maxpool = nn.MaxPool2d(3, 2, 1)
        self.layer1 = self._make(block, 64, layers[0], 1)
        self.layer2 = self._make(block, 128, layers[1], 2)
        self.layer3 = self._make(block, 256, layers[2], 2)
        self.layer4 = self._make(block, 512, layers[3], 2)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(512 * block.expansion, num_classes)
        for m in self.modules():
            if isinstance(m, nn.Conv2d):
                nn.init.kaiming_normal_(m.weight, mode='fan_out', nonlinearity='relu')
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)
        self._out_features = self.fc.in_features

    def _make(self, block, planes, n, stride):
        ds = None
        if stride != 1 or self.inplanes != planes * block.expansion:
            ds = nn.Sequential(_c1(self.inplanes, planes * block.expansion, stride),
                               nn.BatchNorm2d(planes * block.expansion))
        layers = [block(self.inplanes, planes, stride, ds)]
        self.inplanes = planes * block.expansion
        for _ in range(1, n):
            layers.append(block(self.inplanes, planes))
        return nn.Sequential(*layers)

    def forward(self, x):  # resnet.py:23-38
        x = self.maxpool(self.relu(self.bn1(self.conv1(x))))
        return self.layer4(self.layer3(self.layer2(self.layer1(x))))

    @property
    def out_features(self):  # resnet.py:40-43
        return self._out_features


_CFG = {'resnet18': (BasicBlock, [2, 2, 2, 2]), 'resnet34': (BasicBlock, [3, 4, 6, 3]),
        'resnet50': (Bottleneck, [3, 4, 6, 3]), 'resnet101': (Bottleneck, [3, 4, 23, 3])}


def make_backbone(arch):
    block, layers = _CFG[arch]
    return ResNet(block, layers)
