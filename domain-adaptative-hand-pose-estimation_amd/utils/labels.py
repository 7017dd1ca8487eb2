"""Device-side heat-map label generation: the reference builds the Gaussian label maps in
the CPU loader workers (``uda/dataset/util.py:9-68`` ``generate_target``, called from every dataset ``__getitem__``)
and ships (B,21,64,64) fp32 tensors through pinned memory.  Here only the (B,21,2) key-points travel and the maps
are produced on the GPU by the same clipped-Gaussian kernel that builds the pseudo-labels."""
import torch

from mi355 import ops
from uda.model.regda_4 import gaussian_patch

_patch_cache = {}


def generate_target_device(keypoints, visible, heatmap_size=64, sigma=2, image_size=256):
    """keypoints (B,K,2) pixel coordinates in the input image, visible (B,K,1) or (B,K) in {0,1} (CUDA tensors) ->
    (target (B,K,H,H) fp32, target_weight (B,K,1) fp32) with the reference's rule: centre = int(kp / stride + 0.5),
    unnormalised Gaussian of radius 3*sigma clipped at the border, weight 0 (and an all-zero map) when the centre
    falls outside the map or the joint is invisible."""
    if not keypoints.is_cuda:
        raise RuntimeError('generate_target_device needs CUDA tensors')
    B, K, _ = keypoints.shape
    stride = float(image_size) / float(heatmap_size)
    xy = keypoints.float() / stride + 0.5                       # int() truncation happens in the kernel
    mu = xy.to(torch.int32)                                     # same truncation toward zero as Python's int()
    inside = ((mu >= 0) & (mu < heatmap_size)).all(dim=-1, keepdim=True) & (xy > -1).all(dim=-1, keepdim=True)
    vis = visible.reshape(B, K, 1).float()
    weight = vis * inside.float()
    xy = torch.where(weight > 0.5, xy, torch.full_like(xy, -1.0e4)).contiguous()   # far away centre -> all-zero map
    key = (keypoints.device, sigma)
    patch = _patch_cache.get(key)
    if patch is None:
        patch = _patch_cache[key] = torch.from_numpy(gaussian_patch(sigma * 3, sigma).reshape(-1)).to(keypoints.device)
    gt, _ = ops.pseudo_label(xy, patch, sigma * 3, 1, heatmap_size, 1, want_gt=True, want_gf=False)
    return gt, weight
