"""Key-point decode and PCK (reference ``utils/keypoint_detection.py``): ``get_max_preds`` (:7-35),
``calc_dists`` / ``dist_acc`` / ``accuracy`` (:38-92), ``compute_uv_from_heatmaps3`` (:209-239).

The reference works on numpy arrays after a device->host copy of whole heat-maps.  Here the arg-max runs
on the GPU (bit-exact with numpy's first-max rule) and only the (B,K,2) coordinates travel:
numpy inputs are accepted for API parity (uploaded once), torch CUDA tensors avoid the copy."""
import numpy as np
import torch

from mi355 import ops


def _to_dev(hm):
    if isinstance(hm, np.ndarray):
        assert hm.ndim == 4, 'batch_images should be 4-ndim'
        return torch.from_numpy(np.ascontiguousarray(hm, dtype=np.float32)).cuda()
    assert torch.is_tensor(hm) and hm.dim() == 4, 'batch_heatmaps should be numpy.ndarray or a 4-d tensor'
    return hm.detach()


def get_max_preds_device(batch_heatmaps):
    """(preds (B,K,2) fp32 [x,y], maxvals (B,K,1)) as device tensors; no synchronisation."""
    _, xy, mv = ops.argmax2d(_to_dev(batch_heatmaps))
    return xy, mv


def get_max_preds(batch_heatmaps):
    """get predictions from score maps; returns numpy (preds, maxvals) like the reference."""
    xy, mv = get_max_preds_device(batch_heatmaps)
    return xy.cpu().numpy(), mv.cpu().numpy()


def calc_dists(preds, target, normalize):
    preds = preds.astype(np.float32)
    target = target.astype(np.float32)
    dists = np.zeros((preds.shape[1], preds.shape[0]))
    ok = (target[:, :, 0] > 1) & (target[:, :, 1] > 1)
    d = np.linalg.norm(preds / normalize[:, None, :] - target / normalize[:, None, :], axis=2)
    dists[:] = np.where(ok, d, -1).T
    return dists


def dist_acc(dists, thr=0.5):
    """Return percentage below threshold while ignoring values with a -1"""
    dist_cal = np.not_equal(dists, -1)
    num_dist_cal = dist_cal.sum()
    if num_dist_cal > 0:
        return np.less(dists[dist_cal], thr).sum() * 1.0 / num_dist_cal
    return -1


def accuracy(output, target, hm_type='gaussian', thr=0.5):
    """PCK on heat-maps (ground-truth heat-map arg-max as the label), reference :63-92.
    Returns (per-keypoint acc, average acc, count, pred (B,K,2))."""
    pred, _ = get_max_preds(output)
    tgt, _ = get_max_preds(target)
    h, w = output.shape[2], output.shape[3]
    norm = np.ones((pred.shape[0], 2)) * np.array([h, w]) / 10
    dists = calc_dists(pred, tgt, norm)
    K = output.shape[1]
    acc = np.zeros(K)
    avg_acc, cnt = 0, 0
    for i in range(K):
        acc[i] = dist_acc(dists[i], thr)
        if acc[i] >= 0:
            avg_acc += acc[i]
            cnt += 1
    avg_acc = avg_acc / cnt if cnt != 0 else 0
    return acc, avg_acc, cnt, pred


def compute_uv_from_heatmaps3(heatmap: torch.Tensor) -> torch.Tensor:
    """Soft-arg-max: softmax(100*hm) expectation of (column, row), times 4 (reference :209-239)."""
    return ops.softargmax(heatmap.detach(), beta=100.0, out_scale=4.0)
