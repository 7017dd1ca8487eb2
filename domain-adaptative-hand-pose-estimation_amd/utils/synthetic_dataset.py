"""A stand-in for the reference's key-point datasets (``uda/dataset/*``, out of scope: PIL/cv2/torchvision CPU
pipeline, data not available offline) with the same item / attribute contract the training script uses:
``__getitem__ -> (image, target, target_weight, meta)``, ``num_keypoints``, ``keypoints_group``,
``group_accuracy`` (reference ``keypoint_dataset.py:58-71,115-147``)."""
import numpy as np
import torch
from torch.utils.data import Dataset

from .synthetic import generate_target

HAND_GROUPS = {"MCP": (1, 5, 9, 13, 17), "PIP": (2, 6, 10, 14, 18), "DIP": (3, 7, 11, 15, 19),
               "fingertip": (4, 8, 12, 16, 20), "all": tuple(range(21))}


class SyntheticHand21(Dataset):
    num_keypoints = 21
    keypoints_group = HAND_GROUPS

    def __init__(self, length=4096, image_size=(256, 256), heatmap_size=(64, 64), sigma=2, seed=1, **_):
        self.length, self.image_size, self.heatmap_size, self.sigma, self.seed = length, image_size, heatmap_size, sigma, seed

    def __len__(self):
        return self.length

    def __getitem__(self, i):
        rng = np.random.default_rng(self.seed * 1000003 + i)
        S = self.image_size[0]
        image = torch.from_numpy(rng.standard_normal((3, S, S), dtype=np.float32))
        kp = rng.uniform(8, S - 8, size=(21, 2))
        target, weight = generate_target(kp, np.ones((21, 1), np.float32), self.heatmap_size, self.sigma, self.image_size)
        meta = {'keypoint2d': torch.from_numpy(kp.astype(np.float32)), 'image_ema': image}
        return image, torch.from_numpy(target), torch.from_numpy(weight), meta

    def group_accuracy(self, accuracies):
        return {name: sum(accuracies[i] for i in ks) / len(ks) for name, ks in self.keypoints_group.items()}
