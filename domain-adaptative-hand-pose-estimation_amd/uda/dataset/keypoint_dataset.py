"""Base classes of the key-point data sets (reference ``uda/dataset/keypoint_dataset.py``): sample list + transforms +
label geometry, ``group_accuracy`` over named key-point groups (:58-71), the 21-joint hand layout (:115-147)."""
import os
from abc import ABC

import numpy as np
from torch.utils.data.dataset import Dataset

# the colours the skeleton drawings use (the reference resolves them through the `webcolors` package)
_RGB = {'yellow': (255, 255, 0), 'green': (0, 128, 0), 'blue': (0, 0, 255), 'purple': (128, 0, 128), 'red': (255, 0, 0),
        'black': (0, 0, 0)}


class KeypointDataset(Dataset, ABC):
    """root, number of key points, list of samples, transforms (callable on (PIL image, **labels)), input / heat-map
    size as (width, height), Gaussian sigma of the labels, key-point groups and coloured skeleton."""

    def __init__(self, root, num_keypoints, samples, transforms=None, image_size=(256, 256), heatmap_size=(64, 64),
                 sigma=2, keypoints_group=None, colored_skeleton=None):
        self.root, self.num_keypoints, self.samples, self.transforms = root, num_keypoints, samples, transforms
        self.image_size, self.heatmap_size, self.sigma = image_size, heatmap_size, sigma
        self.keypoints_group, self.colored_skeleton = keypoints_group, colored_skeleton

    def __len__(self):
        return len(self.samples)

    def visualize(self, image, keypoints, filename):
        """Draw the skeleton over a PIL image and save it (PIL's ImageDraw in place of the reference's cv2 calls)."""
        from PIL import ImageDraw
        assert self.colored_skeleton is not None
        canvas = image.convert('RGB').copy()
        draw = ImageDraw.Draw(canvas)
        pts = np.asarray(keypoints, dtype=np.float64)
        for line, colour in self.colored_skeleton.values():
            for a, b in zip(line[:-1], line[1:]):
                draw.line([tuple(int(v) for v in pts[a]), tuple(int(v) for v in pts[b])], fill=_RGB[colour], width=3)
        for x, y in pts:
            draw.ellipse([int(x) - 3, int(y) - 3, int(x) + 3, int(y) + 3], outline=_RGB['black'])
        canvas.save(filename)

    def group_accuracy(self, accuracies):
        """Mean accuracy of every named key-point group."""
        return {name: sum(accuracies[i] for i in members) / len(members) for name, members in self.keypoints_group.items()}


class Hand21KeypointDataset(KeypointDataset, ABC):
    """21 hand joints: wrist 0, then (MCP, PIP, DIP, tip) of thumb, index, middle, ring and little finger."""
    MCP = (1, 5, 9, 13, 17)
    PIP = (2, 6, 10, 14, 18)
    DIP = (3, 7, 11, 15, 19)
    fingertip = (4, 8, 12, 16, 20)
    all = tuple(range(21))
    thumb = (0, 1, 2, 3, 4)
    index_finger = (0, 5, 6, 7, 8)
    middle_finger = (0, 9, 10, 11, 12)
    ring_finger = (0, 13, 14, 15, 16)
    little_finger = (0, 17, 18, 19, 20)

    def __init__(self, root, samples, **kwargs):
        skeleton = {"thumb": (self.thumb, 'yellow'), "index_finger": (self.index_finger, 'green'),
                    "middle_finger": (self.middle_finger, 'blue'), "ring_finger": (self.ring_finger, 'purple'),
                    "little_finger": (self.little_finger, 'red')}
        groups = {"MCP": self.MCP, "PIP": self.PIP, "DIP": self.DIP, "fingertip": self.fingertip, "all": self.all}
        super().__init__(root, 21, samples, keypoints_group=groups, colored_skeleton=skeleton, **kwargs)

    # ---- shared by the three hand data sets
    def _labels(self, keypoint2d, visible):
        """(target heat-maps, target weights) as torch tensors."""
        import torch
        from .util import generate_target
        target, weight = generate_target(keypoint2d, visible, self.heatmap_size, self.sigma, self.image_size)
        return torch.from_numpy(target), torch.from_numpy(weight)

    @staticmethod
    def _normalised_pose(keypoint3d_camera):
        """Middle-finger MCP (joint 9) at the origin, wrist -> MCP distance 1.  Returns (pose, scale)."""
        rel = keypoint3d_camera - keypoint3d_camera[9:10, :]
        scale = np.sqrt(np.sum(rel[0, :] ** 2))
        return rel / scale, scale


def _require(root, name):
    if not os.path.exists(os.path.join(root, name)):
        raise FileNotFoundError('Dataset directory %s not found under %s (no network access here: place the extracted data '
                                'set there; the reference would download it)' % (name, root))
