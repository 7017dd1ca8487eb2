"""Seeded, torch-RNG-independent tensors shared by the golden generator and the tests.

Weights come from ``np.random.default_rng`` (stable across torch versions), scaled so
activations stay O(1) (the reference's N(0, 1e-3) head init gives degenerate outputs).
"""
import numpy as np
import torch


def fill_module_(module, seed):
    """Overwrite every parameter/buffer of ``module`` (sorted by name) with seeded values."""
    rng = np.random.default_rng(seed)
    sd = module.state_dict()
    for name in sorted(sd.keys()):
        t = sd[name]
        if name.endswith('num_batches_tracked'):
            t.zero_()
            continue
        shape = tuple(t.shape)
        r = rng.standard_normal(shape).astype(np.float32)
        if name.endswith('running_var'):
            v = 1.0 + 0.25 * np.abs(r)
        elif name.endswith('running_mean'):
            v = 0.1 * r
        elif t.ndim >= 2:
            fan = int(np.prod(shape)) // shape[0]
            v = r * np.float32(1.0 / np.sqrt(fan))
        elif name.endswith('bias'):
            v = 0.1 * r
        else:  # BN weight
            v = 1.0 + 0.1 * r
        t.copy_(torch.from_numpy(np.ascontiguousarray(v, dtype=np.float32)))
    return module


def randn(seed, *shape, scale=1.0):
    rng = np.random.default_rng(seed)
    return torch.from_numpy((rng.standard_normal(shape) * scale).astype(np.float32))


def rand(seed, *shape):
    rng = np.random.default_rng(seed)
    return torch.from_numpy(rng.random(shape).astype(np.float32))


def peaky_heatmaps(seed, B, K, H, W, border=True):
    """Smooth-ish random maps with one dominant peak per (b,k); some peaks on borders,
    one map all non-positive (argmax -> (0,0) with the <=0 mask)."""
    rng = np.random.default_rng(seed)
    hm = (rng.standard_normal((B, K, H, W)) * 0.1).astype(np.float32)
    edge = [0, 1, W - 2, W - 1]
    for b in range(B):
        for k in range(K):
            if border and k < 8:
                x, y = edge[k % 4], edge[(k // 4 + b) % 4] if k >= 4 else int(rng.integers(0, H))
            else:
                x, y = int(rng.integers(0, W)), int(rng.integers(0, H))
            hm[b, k, y, x] += 3.0
    hm[0, K - 1] = -np.abs(hm[0, K - 1])  # all non-positive map
    return torch.from_numpy(hm)


def weights_bk(seed, B, K):
    rng = np.random.default_rng(seed)
    w = (rng.random((B, K, 1)) > 0.15).astype(np.float32)
    return torch.from_numpy(w)


def g9_inputs():
    """Joint sets for the label-generation fixture: interior, on / across every border, far outside, invisible."""
    rng = np.random.default_rng(901)
    B, K = 4, 21
    kp = rng.uniform(-14, 270, size=(B, K, 2))
    kp[0, 0] = (0.0, 0.0); kp[0, 1] = (255.9, 255.9); kp[0, 2] = (3.9, 250.0); kp[0, 3] = (258.0, 10.0)
    kp[0, 4] = (-1.9, 100.0); kp[0, 5] = (-2.1, 100.0); kp[0, 6] = (253.9, 254.0); kp[0, 7] = (254.0, 1.9)
    kp[1, 0] = (128.0, 128.0); kp[1, 1] = (130.0, 126.0); kp[1, 2] = (1.99, 2.0); kp[1, 3] = (2.0, 1.99)
    kp[1, 4] = (12.0, 243.9); kp[1, 5] = (243.9, 12.0); kp[1, 6] = (500.0, 500.0); kp[1, 7] = (-300.0, 20.0)
    vis = (rng.random((B, K, 1)) < 0.8).astype(np.float32)
    vis[0, :8] = 1.0
    vis[1, 0] = 0.0
    return kp, vis
