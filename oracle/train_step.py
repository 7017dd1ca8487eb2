"""One adversarial training iteration (steps A, B, C) and the eval pass (CPU oracle).

Reference: ``train1.py:131-154`` (criteria, 5x SGD-nesterov + LambdaLR),
``train1.py:371-458`` (step A / B / C loss algebra, optimizer order, GL and LR
stepping), ``train1.py:495-536`` (validate).
"""
import torch
import torch.nn as nn
from torch.optim import SGD
from torch.optim.lr_scheduler import LambdaLR

from .backbone import make_backbone
from .pose import Upsampling, PoseResNetx9
from .losses import (JointsKLLoss, PseudoLabelGenerator, PseudoLabelGenerator01, PseudoLabelGenerator03,
                     RegressionDisparityx1, RegressionDisparityx5, RegressionDisparityx6, accuracy)


def build_model(arch='resnet50', num_keypoints=21, num_head_layers=2):  # train1.py:123-127
    backbone = make_backbone(arch)
    upsampling = Upsampling(backbone.out_features)
    return PoseResNetx9(backbone, upsampling, 256, num_keypoints, num_head_layers=num_head_layers, finetune=True)


class DATrainer:
    def __init__(self, model, heatmap_size=64, lr=0.01, momentum=0.9, wd=1e-4, lr_gamma=1e-4, lr_decay=0.75,
                 trade_off=1.0, num_keypoints=21):
        self.model, self.trade_off = model, trade_off
        self.criterion = JointsKLLoss()  # train1.py:131
        kl = lambda: JointsKLLoss(epsilon=1e-7)
        self.rd = RegressionDisparityx6(PseudoLabelGenerator(num_keypoints, heatmap_size, heatmap_size), kl())
        self.rd2 = RegressionDisparityx5(PseudoLabelGenerator03(num_keypoints), kl())
        self.rd1 = RegressionDisparityx1(PseudoLabelGenerator01(num_keypoints), kl())
        mk = lambda ps: SGD(ps, lr=0.1, momentum=momentum, weight_decay=wd, nesterov=True)
        self.opt_f = mk([{'params': model.backbone.parameters(), 'lr': 0.1},
                         {'params': model.upsampling.parameters(), 'lr': 0.1}])  # train1.py:141-144
        self.opt_h = mk(model.head.parameters())
        self.opt_h_adv = mk(model.head_adv.parameters())
        self.opt_h_adv2 = mk(model.head_adv2.parameters())
        self.opt_h_adv3 = mk(model.head_adv3.parameters())
        self.opts = [self.opt_f, self.opt_h, self.opt_h_adv, self.opt_h_adv2, self.opt_h_adv3]
        fn = lambda x: lr * (1. + lr_gamma * float(x)) ** (-lr_decay)  # train1.py:149
        self.scheds = [LambdaLR(o, fn) for o in self.opts]

    def step(self, x_s, label_s, weight_s, x_t, weight_t):
        m, to = self.model, self.trade_off
        m.train()
        # Step A (train1.py:371-397)
        for o in self.opts:
            o.zero_grad()
        y_s, y_s_adv, y_s_adv2, y_s_adv3, _ = m(x_s)
        loss_s = 2 * self.criterion(y_s, label_s, weight_s) + \
            4 * self.rd2(y_s, y_s_adv2, None, weight_s, mode='min') + \
            4 * self.rd(y_s, y_s_adv, None, weight_s, mode='min') + \
            4 * self.rd1(y_s, y_s_adv3, weight_s, mode='min')
        loss_s.backward()
        for o in self.opts:
            o.step()
        # Step B (train1.py:400-436)
        for o in (self.opt_h_adv, self.opt_h_adv2, self.opt_h_adv3):
            o.zero_grad()
        y_t, y_t_adv, y_t_adv2, y_t_adv3, _ = m(x_t)
        loss1 = to * self.rd1(y_t, y_t_adv3, weight_t, mode='max')
        H = y_t.shape[-1]
        target = nn.Upsample(size=H, mode='bilinear')(y_t_adv3.detach())
        target1 = nn.Upsample(size=H, mode='bilinear')(y_t_adv2.detach())
        target0 = nn.Upsample(size=H // 2, mode='bilinear')(y_t_adv3.detach())
        target5 = 0.5 * target + target1
        loss2 = to * self.rd(y_t, y_t_adv, target5, weight_t, mode='max')
        loss3 = to * self.rd2(y_t, y_t_adv2, target0, weight_t, mode='max')
        loss_gf = 0.3 * loss1 + 1 * loss2 + 0.3 * loss3
        loss_gf.backward()
        self.opt_h_adv2.step()
        self.opt_h_adv.step()
        self.opt_h_adv3.step()
        # Step C (train1.py:439-450)
        self.opt_f.zero_grad()
        y_t, y_t_adv, y_t_adv2, y_t_adv3, _ = m(x_t)
        l1 = to * self.rd2(y_t, y_t_adv2, None, weight_t, mode='min')
        l2 = to * self.rd(y_t, y_t_adv, None, weight_t, mode='min')
        loss_gt = 0.3 * l1 + 1 * l2
        loss_gt.backward()
        self.opt_f.step()
        # train1.py:452-458
        m.step()
        for s in self.scheds:
            s.step()
        return {'loss_s': loss_s.detach(), 'loss_gf': loss_gf.detach(), 'loss_gt': loss_gt.detach(),
                'y_s': y_s.detach(), 'y_s_adv': y_s_adv.detach(), 'y_t': y_t.detach(), 'y_t_adv': y_t_adv.detach()}


@torch.no_grad()
def evaluate(model, criterion, x, label, weight):  # train1.py:505-524
    model.eval()
    y = model(x)
    loss = criterion(y, label, weight)
    acc, avg, cnt, pred = accuracy(y.cpu().numpy(), label.cpu().numpy())
    return y, loss, acc, avg, cnt, pred
