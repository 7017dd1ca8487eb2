"""Hand-3D-Studio (real target domain), reference ``uda/dataset/hand_3d_studio.py``: cropped images + annotation.json;
a fixed shuffle (seed 42) puts min(20 %, 3200) samples into the test split (:60-74)."""
import json
import os
import random

import numpy as np
from PIL import Image, ImageFile

from .keypoint_dataset import Hand21KeypointDataset, _require
from .util import keypoint2d_to_3d

ImageFile.LOAD_TRUNCATED_IMAGES = True


class Hand3DStudio(Hand21KeypointDataset):
    def __init__(self, root, split='train', task='noobject', download=True, **kwargs):
        assert split in ['train', 'test', 'all']
        assert task in ['noobject', 'object', 'all']
        self.split, self.task = split, task
        _require(root, "H3D_crop")
        root = os.path.join(root, "H3D_crop")
        annotation_file = os.path.join(root, 'annotation.json')
        print("loading from {}".format(annotation_file))
        with open(annotation_file) as f:
            samples = list(json.load(f))
        if task != 'all':
            want = 1 if task == 'noobject' else 0
            samples = [s for s in samples if int(s['without_object']) == want]
        random.seed(42)
        random.shuffle(samples)
        n_test = min(int(len(samples) * 0.2), 3200)
        if split == 'train':
            samples = samples[n_test:]
        elif split == 'test':
            samples = samples[:n_test]
        super().__init__(root, samples, **kwargs)

    def __getitem__(self, index):
        sample = self.samples[index]
        image = Image.open(os.path.join(self.root, sample['name']))
        Zc = np.array(sample['keypoint3d'])[:, 2]
        image, data = self.transforms(image, keypoint2d=np.array(sample['keypoint2d']),
                                      intrinsic_matrix=np.array(sample['intrinsic_matrix']))
        keypoint2d, K = data['keypoint2d'], data['intrinsic_matrix']
        keypoint3d_camera = keypoint2d_to_3d(keypoint2d, K, Zc)
        target, target_weight = self._labels(keypoint2d, np.ones((self.num_keypoints, 1), dtype=np.float32))
        pose, _ = self._normalised_pose(keypoint3d_camera)
        meta = {'image': sample['name'], 'keypoint2d': keypoint2d, 'keypoint3d': pose,
                'image_ema': data.get('image_ema', image)}
        return image, target, target_weight, meta


class Hand3DStudioAll(Hand3DStudio):
    def __init__(self, root, task='all', **kwargs):
        super().__init__(root, task=task, **kwargs)
