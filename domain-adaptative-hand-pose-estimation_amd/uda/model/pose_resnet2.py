"""Deconvolution neck and the single-head pose net (reference ``uda/model/pose_resnet2.py:11-56`` Upsampling,
``:157-189`` PoseResNet), on the MI355X kernels.  Same class names, constructor signatures, child indices
(state_dict keys ``{0,3,6}.weight`` deconvs, ``{1,4,7}.*`` BatchNorms) and initialisation."""
import torch.nn as nn

from mi355.nn import Conv2d, ConvTranspose2d, BatchNorm2d, ReLU, FusedSequential


class Upsampling(FusedSequential):
    """3-layers deconvolution used in Simple Baseline: 3 x [ConvTranspose2d 4x4 s2 p1 -> BN -> ReLU]."""

    def __init__(self, in_channel=2048, hidden_dims=(256, 256, 256), kernel_sizes=(4, 4, 4), bias=False):
        assert len(hidden_dims) == len(kernel_sizes), 'ERROR: len(hidden_dims) is different len(kernel_sizes)'
        layers = []
        for hidden_dim, kernel_size in zip(hidden_dims, kernel_sizes):
            if kernel_size != 4:
                raise NotImplementedError('kernel_size is {} (only the 4x4 s2 p1 deconvolution is built)'.format(kernel_size))
            layers.append(ConvTranspose2d(in_channel, hidden_dim, kernel_size, stride=2, padding=1,
                                          output_padding=0, bias=bias))
            layers.append(BatchNorm2d(hidden_dim))
            layers.append(ReLU(inplace=True))
            in_channel = hidden_dim
        super().__init__(*layers)
        for m in self.modules():     # init following Simple Baseline
            if isinstance(m, ConvTranspose2d):
                nn.init.normal_(m.weight, std=0.001)
            elif isinstance(m, BatchNorm2d):
                nn.init.constant_(m.weight, 1)
                nn.init.constant_(m.bias, 0)


class PoseResNet(nn.Module):
    """Simple Baseline: backbone + upsampling + 1x1 head; used for source-only pre-training (train1.py:162)."""

    def __init__(self, backbone, upsampling, feature_dim, num_keypoints, finetune=False):
        super().__init__()
        self.backbone = backbone
        self.upsampling = upsampling
        self.head = Conv2d(feature_dim, num_keypoints, 1, 1, 0)
        self.finetune = finetune
        nn.init.normal_(self.head.weight, std=0.001)
        nn.init.constant_(self.head.bias, 0)

    def forward(self, x):
        return self.head(self.upsampling(self.backbone(x)))

    def get_parameters(self, lr=1.):
        return [
            {'params': self.backbone.parameters(), 'lr': 0.1 * lr if self.finetune else lr},
            {'params': self.upsampling.parameters(), 'lr': lr},
            {'params': self.head.parameters(), 'lr': lr},
        ]
