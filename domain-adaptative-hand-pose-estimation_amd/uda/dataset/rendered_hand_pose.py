"""Rendered Hand Pose (synthetic source domain), reference ``uda/dataset/rendered_hand_pose.py``: one sample per
sufficiently large, mostly visible, not overlapped hand (:118-174); crop 1.5x around the hand, mirror left... right hands
so that every sample is a right hand (:60-69), then the transform chain and the labels."""
import os
import pickle

import numpy as np
from PIL import Image

from .keypoint_dataset import Hand21KeypointDataset, _require
from .keypoint_detection import crop, hflip
from .util import area, get_bounding_box, intersection, keypoint2d_to_3d, scale_box

# RHD joint order (wrist, then every finger tip -> base) to ours (wrist, then base -> tip)
_LEFT = [0, 4, 3, 2, 1, 8, 7, 6, 5, 12, 11, 10, 9, 16, 15, 14, 13, 20, 19, 18, 17]
_RIGHT = [i + 21 for i in _LEFT]


class RenderedHandPose(Hand21KeypointDataset):
    def __init__(self, root, split='train', task='all', download=False, **kwargs):
        _require(root, "RHD_published_v2")
        root = os.path.join(root, "RHD_published_v2")
        assert split in ['train', 'test', 'all']
        self.split = split
        parts = ['train', 'test'] if split == 'all' else [split]
        samples = [s for part in parts for s in self.get_samples(root, part)]
        super().__init__(root, samples, **kwargs)

    def __getitem__(self, index):
        sample = self.samples[index]
        image_path = os.path.join(self.root, sample['name'])
        image = Image.open(image_path)
        keypoint2d = np.array(sample['keypoint2d'])
        K = np.array(sample['intrinsic_matrix'])
        Zc = np.array(sample['keypoint3d'])[:, 2]            # cropping / resizing changes Xc, Yc only
        w, h = image.size
        left, upper, right, lower = scale_box(get_bounding_box(keypoint2d), w, h, 1.5)
        image, keypoint2d = crop(image, upper, left, lower - upper, right - left, keypoint2d)
        if sample['left'] is False:                          # (sic: the reference mirrors the samples flagged left=False)
            image, keypoint2d = hflip(image, keypoint2d)
        image, data = self.transforms(image, keypoint2d=keypoint2d, intrinsic_matrix=K)
        keypoint2d, K = data['keypoint2d'], data['intrinsic_matrix']
        keypoint3d_camera = keypoint2d_to_3d(keypoint2d, K, Zc)
        visible = np.array(sample['visible'], dtype=np.float32)[:, np.newaxis]
        target, target_weight = self._labels(keypoint2d, visible)
        pose, scale = self._normalised_pose(keypoint3d_camera)
        meta = {'image': sample['name'], 'keypoint2d': keypoint2d, 'keypoint3d': pose, 'z': pose[:, 2],
                'keypoint3d_camera': keypoint3d_camera, 'cam_param': K, 'image_path': image_path, 'norm_scale': scale,
                'root_deep': keypoint3d_camera[9:10, 2], 'bone_length': scale}
        return image, target, target_weight, meta

    def get_samples(self, root, task, min_size=64):
        part = 'training' if task == 'train' else 'evaluation'
        with open(os.path.join(root, part, 'anno_%s.pickle' % part), 'rb') as fi:
            anno_all = pickle.load(fi)
        samples = []
        w = h = 320
        for sample_id, anno in anno_all.items():
            uv, vis = anno['uv_vis'][:, :2], anno['uv_vis'][:, 2]
            boxes = {True: get_bounding_box(uv[_LEFT]), False: get_bounding_box(uv[_RIGHT])}
            for is_left, idx in ((True, _LEFT), (False, _RIGHT)):
                box = scale_box(boxes[is_left], w, h, 1.5)
                size = max(box[2] - box[0], box[3] - box[1])
                overlap = area(*intersection(box, boxes[not is_left])) / area(*box)
                if size > min_size and np.sum(vis[idx]) > 16 and overlap < 0.3:
                    samples.append({'name': os.path.join(part, 'color', '%.5d.png' % sample_id),
                                    'mask_name': os.path.join(part, 'mask', '%.5d.png' % sample_id),
                                    'keypoint2d': uv[idx], 'visible': vis[idx], 'keypoint3d': anno['xyz'][idx],
                                    'intrinsic_matrix': anno['K'], 'left': is_left})
        return samples
