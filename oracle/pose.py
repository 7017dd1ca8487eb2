"""Neck, heads, gradient layer and PoseResNet models (CPU oracle).

Reference: ``uda/model/pose_resnet2.py:11-56`` (Upsampling), ``:157-189``
(PoseResNet); ``uda/model/regda_7.py:4508-4581`` (make_head), ``:4583-4662``
(make_head2), ``:4861-4962`` (PoseResNetx9), ``:4964-5061`` (PoseResNetx10);
``utils/gl.py:8-69`` (GradientFunction / WarmStartGradientLayer).
"""
import math
import torch
import torch.nn as nn


class Upsampling(nn.Sequential):  # pose_resnet2.py:11-56
    def __init__(self, in_channel=2048, hidden_dims=(256, 256, 256)):
        layers = []
        for h in hidden_dims:
            layers += [nn.ConvTranspose2d(in_channel, h, 4, 2, 1, 0, bias=False),
                       nn.BatchNorm2d(h), nn.ReLU(inplace=True)]
            in_channel = h
        super().__init__(*layers)
        for m in self.modules():
            if isinstance(m, nn.ConvTranspose2d):
                nn.init.normal_(m.weight, std=0.001)
            elif isinstance(m, nn.BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


def _init_convs(seq):
    for m in seq.modules():
        if isinstance(m, nn.Conv2d):
            nn.init.normal_(m.weight, std=0.001)
            nn.init.constant_(m.bias, 0)
    return seq


def simple_head(num_layers, c, k):  # regda_7.py:4906-4929
    layers = []
    for _ in range(num_layers - 1):
        layers += [nn.Conv2d(c, c, 3, 1, 1), nn.BatchNorm2d(c), nn.ReLU()]
    layers.append(nn.Conv2d(c, k, 1, 1, 0))
    return _init_convs(nn.Sequential(*layers))


def _last_lay(c):  # regda_7.py:4544-4571 (one [BN,ReLU,conv3x3 s2,BN,ReLU] group + conv1x1,BN,ReLU)
    return _init_convs(nn.Sequential(
        nn.BatchNorm2d(c), nn.ReLU(), nn.Conv2d(c, c, 3, 2, 1), nn.BatchNorm2d(c), nn.ReLU(),
        nn.Conv2d(c, c, 1, 1, 0), nn.BatchNorm2d(c), nn.ReLU()))


class make_head(nn.Module):  # regda_7.py:4508-4581
    def __init__(self, num_layers, c, k):
        super().__init__()
        self.heatmap_conv = nn.Conv2d(21, 256, 1, 1, bias=True)
        self.feature_conv = nn.Conv2d(256, 256, 1, 1, bias=True)
        self.model = simple_head(num_layers, c, k)
        self.last_lay = _last_lay(c)

    def forward(self, feature, heatmap):
        x = self.heatmap_conv(heatmap) + self.feature_conv(feature)
        return self.model(self.last_lay(x))


class make_head2(nn.Module):  # regda_7.py:4583-4662
    def __init__(self, num_layers, c, k):
        super().__init__()
        self.heatmap_conv = nn.Conv2d(21, 256, 1, 1, bias=True)
        self.feature_conv = nn.Conv2d(256, 256, 3, 2, 1, bias=True)
        self.upsample = nn.Upsample(size=64, mode='bilinear')  # unused (regda_7.py:4590)
        self.model = simple_head(num_layers, c, k)
        self.last_lay = _last_lay(c)  # _make_head2(2,...) loops range(1) -> same structure

    def forward(self, feature, heatmap):
        x = self.heatmap_conv(heatmap)
        x = x + self.feature_conv(feature)
        return self.model(self.last_lay(x))


class GradientFunction(torch.autograd.Function):  # utils/gl.py:8-18 (+coeff, NOT a reversal)
    @staticmethod
    def forward(ctx, inp, coeff=1.):
        ctx.coeff = coeff
        return inp * 1.0

    @staticmethod
    def backward(ctx, g):
        return g * ctx.coeff, None


def gl_coeff(i, alpha=1.0, lo=0.0, hi=0.1, max_iters=1000):  # utils/gl.py:59-62
    return float(2.0 * (hi - lo) / (1.0 + math.exp(-alpha * i / max_iters)) - (hi - lo) + lo)


class WarmStartGradientLayer(nn.Module):  # utils/gl.py:21-69
    def __init__(self, alpha=1.0, lo=0.0, hi=1., max_iters=1000., auto_step=False):
        super().__init__()
        self.alpha, self.lo, self.hi = alpha, lo, hi
        self.iter_num, self.max_iters, self.auto_step = 0, max_iters, auto_step

    def forward(self, x):
        c = gl_coeff(self.iter_num, self.alpha, self.lo, self.hi, self.max_iters)
        if self.auto_step:
            self.step()
        return GradientFunction.apply(x, c)

    def step(self):
        self.iter_num += 1


class PoseResNet(nn.Module):  # pose_resnet2.py:157-189
    def __init__(self, backbone, upsampling, feature_dim, num_keypoints, finetune=False):
        super().__init__()
        self.backbone, self.upsampling = backbone, upsampling
        self.head = nn.Conv2d(feature_dim, num_keypoints, 1, 1, 0)
        self.finetune = finetune
        nn.init.normal_(self.head.weight, std=0.001)
        nn.init.constant_(self.head.bias, 0)

    def forward(self, x):
        return self.head(self.upsampling(self.backbone(x)))

    def get_parameters(self, lr=1.):
        return [{'params': self.backbone.parameters(), 'lr': 0.1 * lr if self.finetune else lr},
                {'params': self.upsampling.parameters(), 'lr': lr},
                {'params': self.head.parameters(), 'lr': lr}]


class PoseResNetx9(nn.Module):  # regda_7.py:4861-4962
    always_tuple = False

    def __init__(self, backbone, upsampling, feature_dim, num_keypoints, gl=None, finetune=True,
                 num_head_layers=2):
        super().__init__()
        self.backbone, self.upsampling = backbone, upsampling
        self.head = simple_head(num_head_layers, feature_dim, num_keypoints)
        self.head_adv = simple_head(num_head_layers, feature_dim, num_keypoints)
        self.head_adv2 = make_head(num_head_layers, feature_dim, num_keypoints)
        self.head_adv3 = make_head2(num_head_layers, feature_dim, num_keypoints)
        self.finetune = finetune
        self.gl_layer = WarmStartGradientLayer(1.0, 0.0, 0.1, 1000, False) if gl is None else gl

    def heads(self, f):
        f_adv = self.gl_layer(f)
        y = self.head(f)
        y_adv = self.head_adv(f_adv)
        y_adv2 = self.head_adv2(f_adv, y_adv)
        y_adv3 = self.head_adv3(f_adv, y_adv2)
        return y, y_adv, y_adv2, y_adv3

    def forward(self, x):
        f = self.upsampling(self.backbone(x))
        y, y_adv, y_adv2, y_adv3 = self.heads(f)
        if self.training or self.always_tuple:
            return y, y_adv, y_adv2, y_adv3, f
        return y

    def get_parameters(self, lr=1.):
        return [{'params': self.backbone.parameters(), 'lr': 0.1 * lr if self.finetune else lr}] + \
               [{'params': getattr(self, n).parameters(), 'lr': lr}
                for n in ('upsampling', 'head', 'head_adv', 'head_adv2', 'head_adv3')]

    def step(self):
        self.gl_layer.step()


class PoseResNetx10(PoseResNetx9):  # regda_7.py:4964-5061 (always returns the 5-tuple)
    always_tuple = True
