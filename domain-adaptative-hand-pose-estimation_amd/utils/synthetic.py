"""Synthetic source/target batches of the reference's shapes: images ~ N(0,1) (stands for
ImageNet-normalised pixels, train1.py:55), 21 key-points ~ U[8, S-8)^2, labels = Gaussian heat-maps by the
reference's ``generate_target`` rule (uda/dataset/util.py:9-68).

Every key-point is visible by default (as in the H3D reader, hand_3d_studio.py:99): an invisible joint gets an
all-zero label map, which the reference's supervised KL term (epsilon=0, loss.py:150-151) turns into 0/0 = NaN
for the whole batch loss; ``p_visible`` < 1 reproduces that case for tests."""
import numpy as np
import torch


def generate_target(joints, joints_vis, heatmap_size, sigma, image_size):
    """Heat-map label for K joints: unnormalised Gaussian (sigma, 3*sigma radius) centred at
    int(joint / stride + 0.5), clipped at the border; weight 0 when the centre falls outside."""
    K = joints.shape[0]
    weight = np.ones((K, 1), dtype=np.float32)
    weight[:, 0] = joints_vis[:, 0]
    W, H = heatmap_size
    target = np.zeros((K, H, W), dtype=np.float32)
    r = sigma * 3
    ax = np.arange(0, 2 * r + 1, 1, np.float32)
    g = np.exp(-((ax - r) ** 2 + (ax[:, None] - r) ** 2) / (2 * sigma ** 2))
    stride = np.array(image_size) / np.array(heatmap_size)
    for j in range(K):
        mx, my = int(joints[j][0] / stride[0] + 0.5), int(joints[j][1] / stride[1] + 0.5)
        if mx >= W or my >= H or mx < 0 or my < 0:
            weight[j] = 0
            continue
        x0, x1, y0, y1 = max(0, mx - r), min(mx + r + 1, W), max(0, my - r), min(my + r + 1, H)
        if weight[j] > 0.5:
            target[j][y0:y1, x0:x1] = g[y0 - (my - r):y1 - (my - r), x0 - (mx - r):x1 - (mx - r)]
    return target, weight


def make_batch(batch_size, image_size=256, heatmap_size=64, num_keypoints=21, seed=1, device='cpu', with_target_labels=True,
               p_visible=1.0):
    """dict(x_s, label_s, w_s, x_t, w_t[, label_t]) as fp32 tensors on `device`."""
    rng = np.random.default_rng(seed)
    S, Hm, K, B = image_size, heatmap_size, num_keypoints, batch_size

    def labels():
        lab = np.zeros((B, K, Hm, Hm), np.float32)
        w = np.zeros((B, K, 1), np.float32)
        for b in range(B):
            kp = rng.uniform(8, S - 8, size=(K, 2))
            vis = (rng.random((K, 1)) < p_visible).astype(np.float32)
            lab[b], w[b] = generate_target(kp, vis, (Hm, Hm), 2, (S, S))
        return torch.from_numpy(lab), torch.from_numpy(w)

    x_s = torch.from_numpy(rng.standard_normal((B, 3, S, S), dtype=np.float32))
    x_t = torch.from_numpy(rng.standard_normal((B, 3, S, S), dtype=np.float32))
    label_s, w_s = labels()
    label_t, w_t = labels()
    out = dict(x_s=x_s, label_s=label_s, w_s=w_s, x_t=x_t, w_t=w_t)
    if with_target_labels:
        out['label_t'] = label_t
    return {k: v.to(device) for k, v in out.items()}
