"""Losses, pseudo-label generators, keypoint decode and PCK (CPU oracle).

Reference: ``uda/model/loss.py:115-158`` (JointsKLLoss);
``uda/model/regda_4.py:17-86`` (PseudoLabelGenerator);
``uda/model/regda_7.py:2956-3039`` (PseudoLabelGenerator01), ``:3118-3201``
(PseudoLabelGenerator03), ``:3206-3268`` (RegressionDisparityx1), ``:3485-3561``
(x5), ``:3564-3632`` (x6); ``utils/keypoint_detection.py:7-35`` (get_max_preds),
``:38-92`` (calc_dists / dist_acc / accuracy), ``:209-239``
(compute_uv_from_heatmaps3); ``uda/dataset/util.py:9-68`` (generate_target);
``uda/dataset/keypoint_dataset.py:58-71,115-147`` (group_accuracy, hand groups).
"""
import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F


# ----------------------------------------------------------------- decode / metric
def get_max_preds(hm):  # utils/keypoint_detection.py:7-35
    assert isinstance(hm, np.ndarray) and hm.ndim == 4
    B, K, _, W = hm.shape
    flat = hm.reshape((B, K, -1))
    idx = np.argmax(flat, 2).reshape((B, K, 1))
    maxvals = np.amax(flat, 2).reshape((B, K, 1))
    preds = np.tile(idx, (1, 1, 2)).astype(np.float32)
    preds[:, :, 0] = preds[:, :, 0] % W
    preds[:, :, 1] = np.floor(preds[:, :, 1] / W)
    preds *= np.tile(np.greater(maxvals, 0.0), (1, 1, 2)).astype(np.float32)
    return preds, maxvals


def calc_dists(preds, target, normalize):  # utils/keypoint_detection.py:38-50
    preds, target = preds.astype(np.float32), target.astype(np.float32)
    dists = np.zeros((preds.shape[1], preds.shape[0]))
    for n in range(preds.shape[0]):
        for c in range(preds.shape[1]):
            if target[n, c, 0] > 1 and target[n, c, 1] > 1:
                dists[c, n] = np.linalg.norm(preds[n, c, :] / normalize[n] - target[n, c, :] / normalize[n])
            else:
                dists[c, n] = -1
    return dists


def dist_acc(dists, thr=0.5):  # utils/keypoint_detection.py:53-60
    cal = np.not_equal(dists, -1)
    n = cal.sum()
    return np.less(dists[cal], thr).sum() * 1.0 / n if n > 0 else -1


def accuracy(output, target, thr=0.5):  # utils/keypoint_detection.py:63-92
    pred, _ = get_max_preds(output)
    tgt, _ = get_max_preds(target)
    h, w = output.shape[2], output.shape[3]
    norm = np.ones((pred.shape[0], 2)) * np.array([h, w]) / 10
    dists = calc_dists(pred, tgt, norm)
    K = output.shape[1]
    acc = np.zeros(K)
    avg, cnt = 0, 0
    for i in range(K):
        acc[i] = dist_acc(dists[i], thr)
        if acc[i] >= 0:
            avg += acc[i]
            cnt += 1
    return acc, (avg / cnt if cnt else 0), cnt, pred


HAND_GROUPS = {"MCP": (1, 5, 9, 13, 17), "PIP": (2, 6, 10, 14, 18), "DIP": (3, 7, 11, 15, 19),
               "fingertip": (4, 8, 12, 16, 20), "all": tuple(range(21))}  # keypoint_dataset.py:115-147


def group_accuracy(acc, groups=HAND_GROUPS):  # keypoint_dataset.py:58-71
    return {n: sum(acc[i] for i in ks) / len(ks) for n, ks in groups.items()}


def soft_argmax(hm):  # utils/keypoint_detection.py:209-239 (compute_uv_from_heatmaps3)
    hm = hm.mul(100)
    B, K, H, W = hm.size()
    sm = F.softmax(hm.view(B, K, H * W), dim=2).view(B, K, H, W)
    xx, yy = torch.meshgrid(torch.arange(H), torch.arange(W), indexing='ij')
    ax = sm.mul(xx.float()).view(B, K, H * W).sum(2).unsqueeze(2)
    ay = sm.mul(yy.float()).view(B, K, H * W).sum(2).unsqueeze(2)
    return torch.cat([ay, ax], 2) * 4


# ----------------------------------------------------------------- labels
def _gauss_patch(tmp_size, sigma):
    size = 2 * tmp_size + 1
    x = np.arange(0, size, 1, np.float32)
    y = x[:, np.newaxis]
    x0 = y0 = size // 2
    return np.exp(- ((x - x0) ** 2 + (y - y0) ** 2) / (2 * sigma ** 2))


def generate_target(joints, joints_vis, heatmap_size, sigma, image_size):  # uda/dataset/util.py:9-68
    K = joints.shape[0]
    w = np.ones((K, 1), dtype=np.float32)
    w[:, 0] = joints_vis[:, 0]
    target = np.zeros((K, heatmap_size[1], heatmap_size[0]), dtype=np.float32)
    tmp = sigma * 3
    stride = np.array(image_size) / np.array(heatmap_size)
    g = _gauss_patch(tmp, sigma)
    for j in range(K):
        mu_x = int(joints[j][0] / stride[0] + 0.5)
        mu_y = int(joints[j][1] / stride[1] + 0.5)
        ul = [int(mu_x - tmp), int(mu_y - tmp)]
        br = [int(mu_x + tmp + 1), int(mu_y + tmp + 1)]
        if mu_x >= heatmap_size[0] or mu_y >= heatmap_size[1] or mu_x < 0 or mu_y < 0:
            w[j] = 0
            continue
        gx = max(0, -ul[0]), min(br[0], heatmap_size[0]) - ul[0]
        gy = max(0, -ul[1]), min(br[1], heatmap_size[1]) - ul[1]
        ix = max(0, ul[0]), min(br[0], heatmap_size[0])
        iy = max(0, ul[1]), min(br[1], heatmap_size[1])
        if w[j] > 0.5:
            target[j][iy[0]:iy[1], ix[0]:ix[1]] = g[gy[0]:gy[1], gx[0]:gx[1]]
    return target, w


def _table(width, height, tmp_size, sigma):  # regda_4.py:46-74 / regda_7.py:2986-3014 / 3148-3176
    hm = np.zeros((width, height, height, width), dtype=np.float32)
    g = _gauss_patch(tmp_size, sigma)
    for mx in range(width):
        for my in range(height):
            ul = [int(mx - tmp_size), int(my - tmp_size)]
            br = [int(mx + tmp_size + 1), int(my + tmp_size + 1)]
            gx = max(0, -ul[0]), min(br[0], width) - ul[0]
            gy = max(0, -ul[1]), min(br[1], height) - ul[1]
            ix = max(0, ul[0]), min(br[0], width)
            iy = max(0, ul[1]), min(br[1], height)
            hm[mx][my][iy[0]:iy[1], ix[0]:ix[1]] = g[gy[0]:gy[1], gx[0]:gx[1]]
    return hm


class PseudoLabelGenerator(nn.Module):  # regda_4.py:17-86
    def __init__(self, num_keypoints, height=64, width=64, sigma=2):
        super().__init__()
        self.height, self.width, self.sigma = height, width, sigma
        self.heatmaps = _table(width, height, sigma * 3, sigma)
        self.false_matrix = 1. - np.eye(num_keypoints, dtype=np.float32)

    def forward(self, y):
        B, K, H, W = y.shape
        preds, _ = get_max_preds(y.detach().cpu().numpy())
        preds = preds.reshape(-1, 2).astype(int)
        gt = self.heatmaps[preds[:, 0], preds[:, 1], :, :].copy().reshape(B, K, H, W).copy()
        gf = gt.reshape(B, K, -1).transpose((0, 2, 1))
        gf = gf.dot(self.false_matrix).clip(max=1., min=0.).transpose((0, 2, 1)).reshape(B, K, H, W).copy()
        return torch.from_numpy(gt), torch.from_numpy(gf)


class _PLGCoarse(nn.Module):
    size, div, tmp = None, None, None

    def __init__(self, num_keypoints, sigma=2):
        super().__init__()
        self.heatmaps = _table(self.size, self.size, self.tmp(sigma), sigma)

    def forward(self, y):
        B, K, H, W = y.shape
        preds, _ = get_max_preds(y.detach().cpu().numpy())
        preds = (preds.reshape(-1, 2) / self.div).astype(int)
        gt = self.heatmaps[preds[:, 0], preds[:, 1], :, :].copy().reshape(B, K, self.size, self.size).copy()
        gf = (np.ones_like(gt) - gt * 10).clip(max=1., min=0.)
        return torch.from_numpy(gt), torch.from_numpy(gf)


class PseudoLabelGenerator01(_PLGCoarse):  # regda_7.py:2956-3039: 16x16, tmp_size = sigma*1.5, preds/4
    size, div = 16, 4
    tmp = staticmethod(lambda s: s * 1.5)


class PseudoLabelGenerator03(_PLGCoarse):  # regda_7.py:3118-3201: 32x32, tmp_size = sigma*2, preds/2
    size, div = 32, 2
    tmp = staticmethod(lambda s: s * 2)


# ----------------------------------------------------------------- losses
class JointsKLLoss(nn.Module):  # uda/model/loss.py:115-158
    def __init__(self, reduction='mean', epsilon=0.):
        super().__init__()
        self.criterion = nn.KLDivLoss(reduction='none')
        self.reduction, self.epsilon = reduction, epsilon

    def forward(self, output, target, target_weight=None):
        B, K, _, _ = output.shape
        lp = F.log_softmax(output.reshape((B, K, -1)), dim=-1)
        t = target.reshape((B, K, -1)) + self.epsilon
        t = t / t.sum(dim=-1, keepdims=True)
        loss = self.criterion(lp, t).sum(dim=-1)
        if target_weight is not None:
            loss = loss * target_weight.view((B, K))
        return loss.mean() if self.reduction == 'mean' else loss.mean(dim=-1)


def _max_normalise(gf):  # regda_7.py:3546-3548 / 3623-3625 (per-(b,k) map divided by its max)
    b, c = gf.shape[:2]
    return torch.stack([torch.stack([gf[k][j] / torch.max(gf[k][j]) for j in range(c)]) for k in range(b)])


class RegressionDisparityx1(nn.Module):  # regda_7.py:3206-3268
    def __init__(self, plg, criterion):
        super().__init__()
        self.criterion, self.pseudo_label_generator = criterion, plg

    def forward(self, y, y_adv, weight=None, mode='min'):
        assert mode in ['min', 'max']
        gt, _ = self.pseudo_label_generator(y.detach())
        gf = (torch.ones_like(gt) - gt * 10).clip(max=1., min=0.)
        self.ground_truth, self.ground_false = gt, gf
        return self.criterion(y_adv, gt if mode == 'min' else gf, weight)


class RegressionDisparityx5(nn.Module):  # regda_7.py:3485-3561
    def __init__(self, plg, criterion):
        super().__init__()
        self.criterion, self.pseudo_label_generator = criterion, plg

    def forward(self, y, y_adv, y_adv2, weight=None, mode='min'):
        assert mode in ['min', 'max']
        gt, gf = self.pseudo_label_generator(y.detach())
        gf1 = (torch.ones_like(gt) - gt * 10).clip(max=1., min=0.)
        if y_adv2 is not None:
            gf = gf1 + y_adv2
            gf = (gf - gt * 100).clip(max=1., min=0.)
        gf = _max_normalise(gf)
        self.ground_truth, self.ground_false = gt, gf
        return self.criterion(y_adv, gt if mode == 'min' else gf, weight)


class RegressionDisparityx6(nn.Module):  # regda_7.py:3564-3632
    def __init__(self, plg, criterion):
        super().__init__()
        self.criterion, self.pseudo_label_generator = criterion, plg

    def forward(self, y, y_adv, y_adv2, weight=None, mode='min'):
        assert mode in ['min', 'max']
        gt, gf = self.pseudo_label_generator(y.detach())
        label_p = torch.sum(gt, dim=1).clip(max=1., min=0.)
        label_p = label_p.unsqueeze(1).repeat(1, 21, 1, 1)
        gf = (label_p - gt * 10).clip(max=1., min=0.)
        if y_adv2 is not None:
            gf = gf + y_adv2
            gf = (gf - gt * 100).clip(max=1., min=0.)
        gf = _max_normalise(gf)
        self.ground_truth, self.ground_false = gt, gf
        return self.criterion(y_adv, gt if mode == 'min' else gf, weight)
