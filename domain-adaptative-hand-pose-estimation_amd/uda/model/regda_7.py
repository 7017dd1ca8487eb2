"""The live classes of the reference's ``uda/model/regda_7.py`` on the MI355X kernels:
``make_head`` (:4508-4581), ``make_head2`` (:4583-4662), ``PoseResNetx9`` (:4861-4962), ``PoseResNetx10``
(:4964-5061), ``PseudoLabelGenerator01`` (:2956-3039), ``PseudoLabelGenerator03`` (:3118-3201),
``RegressionDisparityx1`` (:3206-3268), ``x5`` (:3485-3561), ``x6`` (:3564-3632).
Same names, constructor signatures, attribute names and state_dict keys."""
from typing import Optional

import torch
import torch.nn as nn

from mi355.nn import Conv2d, BatchNorm2d, ReLU, FusedSequential, GradFanIn
from utils.gl import WarmStartGradientLayer
from uda.model.regda_4 import _GaussianLabels, PseudoLabelGenerator


def _init_head(layers):
    for m in layers.modules():
        if isinstance(m, Conv2d):
            nn.init.normal_(m.weight, std=0.001)
            nn.init.constant_(m.bias, 0)
    return layers


def _simple_head(num_layers, channel_dim, num_keypoints):
    """[conv3x3 -> BN -> ReLU] x (num_layers-1) -> conv1x1 (reference _make_head)."""
    layers = []
    for _ in range(num_layers - 1):
        layers.extend([Conv2d(channel_dim, channel_dim, 3, 1, 1), BatchNorm2d(channel_dim), ReLU()])
    layers.append(Conv2d(channel_dim, num_keypoints, 1, 1, 0))
    return _init_head(FusedSequential(*layers))


def _fusion_tail(groups, channel_dim):
    """[BN, ReLU, conv3x3 s2, BN, ReLU] x groups -> conv1x1 -> BN -> ReLU (reference _make_head2)."""
    layers = []
    for _ in range(groups):
        layers.extend([BatchNorm2d(channel_dim), ReLU(), Conv2d(channel_dim, channel_dim, 3, 2, 1),
                       BatchNorm2d(channel_dim), ReLU()])
    layers.extend([Conv2d(channel_dim, channel_dim, 1, 1, 0), BatchNorm2d(channel_dim), ReLU()])
    return _init_head(FusedSequential(*layers))


class make_head(nn.Module):
    """Multiscale-fusion head, level 1: (features 64x64, heat-map 64x64) -> heat-map 32x32."""

    def __init__(self, num_layers, channel_dim, num_keypoints):
        super().__init__()
        self.heatmap_conv = Conv2d(21, 256, 1, 1, 0, bias=True)
        self.feature_conv = Conv2d(256, 256, 1, 1, 0, bias=True)
        self.model = _simple_head(num_layers, channel_dim, num_keypoints)
        self.heatmap_conv.bn_follows = True      # its output goes straight into last_lay's BatchNorm: statistics in the epilogue
        self.last_lay = _fusion_tail(1, channel_dim)

    def forward(self, feature, heatmap):
        # heatmap_conv(heatmap) + feature_conv(feature) (reference :4575): one concat-K GEMM in training; otherwise the add is
        # fused into the 21->256 kernel's epilogue
        x = self.feature_conv.forward_cat(feature, heatmap, self.heatmap_conv)
        if x is None:
            x = self.heatmap_conv(heatmap, residual=self.feature_conv(feature))
        return self.model(self.last_lay(x))


class make_head2(nn.Module):
    """Multiscale-fusion head, level 2: (features 64x64, heat-map 32x32) -> heat-map 16x16."""

    def __init__(self, num_layers, channel_dim, num_keypoints):
        super().__init__()
        self.heatmap_conv = Conv2d(21, 256, 1, 1, 0, bias=True)
        self.feature_conv = Conv2d(256, 256, 3, 2, 1, bias=True)
        self.upsample = nn.Upsample(size=64, mode='bilinear')   # unused in the reference forward as well
        self.model = _simple_head(num_layers, channel_dim, num_keypoints)
        self.heatmap_conv.bn_follows = True      # its output goes straight into last_lay's BatchNorm: statistics in the epilogue
        self.last_lay = _fusion_tail(2 - 1, channel_dim)        # reference loops range(num_layers - 1) with 2

    def forward(self, feature, heatmap):
        x = self.feature_conv.forward_cat(feature, heatmap, self.heatmap_conv)      # (reference :4651-4654)
        if x is None:
            x = self.heatmap_conv(heatmap, residual=self.feature_conv(feature))
        return self.model(self.last_lay(x))


class PoseResNetx9(nn.Module):
    """Pose ResNet with one backbone, one upsampling neck, the main head and three cascaded adversarial heads.

    forward(x): train -> (y, y_adv, y_adv2, y_adv3, f); eval -> y.
    ``detach_features=True`` (extension) stops the backward at the neck output: used by step B of the
    training loop, whose backbone gradients the reference computes and then discards (train1.py:440)."""
    _always_tuple = False

    def __init__(self, backbone, upsampling, feature_dim, num_keypoints,
                 gl: Optional[WarmStartGradientLayer] = None, finetune: Optional[bool] = True, num_head_layers=2):
        super().__init__()
        self.backbone = backbone
        self.upsampling = upsampling
        self.head = _simple_head(num_head_layers, feature_dim, num_keypoints)
        self.head_adv = _simple_head(num_head_layers, feature_dim, num_keypoints)
        self.head_adv2 = make_head(num_head_layers, feature_dim, num_keypoints)
        self.head_adv3 = make_head2(num_head_layers, feature_dim, num_keypoints)
        self.finetune = finetune
        self.gl_layer = WarmStartGradientLayer(alpha=1.0, lo=0.0, hi=0.1, max_iters=1000, auto_step=False) \
            if gl is None else gl

    def features(self, x, detach_features=False):
        if detach_features:
            with torch.no_grad():
                return self.upsampling(self.backbone(x))
        f = self.upsampling(self.backbone(x))
        f._mi_bn_src = None      # f feeds four heads: its gradient is a sum, no single dgrad epilogue can reduce it
        f._mi_fan = GradFanIn()  # ... and that sum is formed inside the heads' dgrad epilogues, not by autograd add kernels
        return f

    def adv_heads(self, f):
        """The three cascaded adversarial heads behind the gradient layer: f -> (y_adv, y_adv2, y_adv3)."""
        f_adv = self.gl_layer(f)
        y_adv = self.head_adv(f_adv)
        y_adv2 = self.head_adv2(f_adv, y_adv)
        y_adv3 = self.head_adv3(f_adv, y_adv2)
        return y_adv, y_adv2, y_adv3

    def forward(self, x, detach_features=False):
        f = self.features(x, detach_features)
        if not (self.training or self._always_tuple):
            return self.head(f)             # eval: the reference also runs the adv heads and drops them
        y = self.head(f)
        y_adv, y_adv2, y_adv3 = self.adv_heads(f)
        return y, y_adv, y_adv2, y_adv3, f

    def get_parameters(self, lr=1.):
        return [
            {'params': self.backbone.parameters(), 'lr': 0.1 * lr if self.finetune else lr},
            {'params': self.upsampling.parameters(), 'lr': lr},
            {'params': self.head.parameters(), 'lr': lr},
            {'params': self.head_adv.parameters(), 'lr': lr},
            {'params': self.head_adv2.parameters(), 'lr': lr},
            {'params': self.head_adv3.parameters(), 'lr': lr},
        ]

    def step(self):
        """Call step() each iteration during training. Will increase lambda in GL layer."""
        self.gl_layer.step()


class PoseResNetx10(PoseResNetx9):
    """Same network; forward always returns the 5-tuple (frozen EMA copy, train1.py:102-119)."""
    _always_tuple = True


class PseudoLabelGenerator01(_GaussianLabels):
    """16x16 labels: centre = trunc(arg-max / 4), 7x7 Gaussian patch (tmp_size = 1.5*sigma)."""

    def __init__(self, num_keypoints, height=16, width=16, sigma=2):
        super().__init__(width, 4, sigma * 1.5, sigma)

    def forward(self, y):
        return self.labels(y, kind=1)


class PseudoLabelGenerator03(_GaussianLabels):
    """32x32 labels: centre = trunc(arg-max / 2), 9x9 Gaussian patch (tmp_size = 2*sigma)."""

    def __init__(self, num_keypoints, height=32, width=32, sigma=2):
        super().__init__(width, 2, sigma * 2, sigma)

    def forward(self, y):
        return self.labels(y, kind=1)


class _Disparity(nn.Module):
    kind = 1          # ground-false rule of the builder kernel
    normalise = True  # per-map division by its maximum
    guard_empty_maps = False   # extension (off = reference): leave an all-zero ground-false map at zero instead of 0/0 = NaN

    def __init__(self, pseudo_label_generator, criterion: nn.Module):
        super().__init__()
        self.criterion = criterion
        self.pseudo_label_generator = pseudo_label_generator

    def _run(self, y, y_adv, y_adv2, weight, mode, scale=1.0):
        assert mode in ['min', 'max']
        gen = self.pseudo_label_generator
        # only the label the mode needs is materialised (the reference builds both and drops one)
        gt, gf = gen.labels(y.detach(), self.kind, extra=None if y_adv2 is None else y_adv2.detach(),
                            normalise=(2 if (self.normalise and self.guard_empty_maps) else self.normalise),
                            want_gt=(mode == 'min'), want_gf=(mode == 'max'))
        self.ground_truth, self.ground_false = gt, gf
        if scale == 1.0:
            return self.criterion(y_adv, gt if mode == 'min' else gf, weight)
        return self.criterion(y_adv, gt if mode == 'min' else gf, weight, scale=scale)


class RegressionDisparityx1(_Disparity):
    """Level-2 (16x16) disparity: gf = clip(1 - 10 gt, 0, 1), no max-normalisation."""
    kind, normalise = 1, False

    def forward(self, y, y_adv, weight=None, mode='min', scale=1.0):
        return self._run(y, y_adv, None, weight, mode, scale)


class RegressionDisparityx5(_Disparity):
    """Level-1 (32x32): gf = clip(1 - 10 gt, 0, 1) [+ y_adv2 - 100 gt, clipped], divided by its per-map max."""
    kind, normalise = 1, True

    def forward(self, y, y_adv, y_adv2, weight=None, mode='min', scale=1.0):
        return self._run(y, y_adv, y_adv2, weight, mode, scale)


class RegressionDisparityx6(_Disparity):
    """Level-0 (64x64): gf = clip(clip(sum_k gt_k, 0, 1) - 10 gt, 0, 1) [+ y_adv2 - 100 gt, clipped], / per-map max."""
    kind, normalise = 2, True

    def forward(self, y, y_adv, y_adv2, weight=None, mode='min', scale=1.0):
        return self._run(y, y_adv, y_adv2, weight, mode, scale)
