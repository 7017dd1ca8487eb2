"""Stereo Tracking Benchmark (second real target domain), reference ``uda/dataset/STB.py``: 12 sequences of 640x480
colour frames, labels in the depth camera's frame (``labels/<seq>_SK.mat``, ``handPara`` 3 x 21 x N, millimetres) moved
into the colour camera (:215-221), joints re-ordered to the 21-joint layout, the palm centre pushed out to a wrist
position (:191-205); crop 1.6x around the hand (:117-121).  Sequences B1* are the test split (:88-96)."""
import math
import os

import numpy as np
from PIL import Image

from .keypoint_dataset import Hand21KeypointDataset
from .keypoint_detection import crop
from .util import get_bounding_box, keypoint2d_to_3d, keypoint3d_to_2d, scale_box

# colour camera of the SK (Intel F200) rig and its pose relative to the depth camera
SK_fx_color, SK_fy_color, SK_tx_color, SK_ty_color = 607.92271, 607.88192, 314.78337, 236.42484
SK_rot_vec = [0.00531, -0.01196, 0.00301]
SK_trans_vec = [-24.0381, -0.4563, -1.2326]  # mm


def SK_rot_mx(rot_vec):
    """Rotation vector -> matrix through the unit quaternion (a, b, c, d) = (cos t/2, -axis sin t/2)."""
    theta = np.linalg.norm(rot_vec)
    a = math.cos(theta / 2.0)
    b, c, d = -(np.array(rot_vec) * math.sin(theta / 2.0) / theta)
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c + a * d), 2 * (b * d - a * c)],
                     [2 * (b * c - a * d), a * a + c * c - b * b - d * d, 2 * (c * d + a * b)],
                     [2 * (b * d + a * c), 2 * (c * d - a * b), a * a + d * d - b * b - c * c]])


SK_rot = SK_rot_mx(SK_rot_vec)
intrinsic_matrix0 = np.asarray([[SK_fx_color, 0, SK_tx_color], [0, SK_fy_color, SK_ty_color], [0, 0, 1]])
SEQUENCES = ["B1Counting", "B1Random", "B2Counting", "B2Random", "B3Counting", "B3Random", "B4Counting", "B4Random",
             "B5Counting", "B5Random", "B6Counting", "B6Random"]
# STB joint order (palm, little ... thumb, each base -> tip) to ours (wrist, thumb ... little)
_HAND_INDEX = [0, 17, 18, 19, 20, 13, 14, 15, 16, 9, 10, 11, 12, 5, 6, 7, 8, 1, 2, 3, 4]


def _push_root(pose_xyz, towards, factor):
    """Joint 0 (palm centre) moved along the line from joint `towards` through it: a wrist estimate."""
    out = pose_xyz.copy()
    out[:, 0, :] = pose_xyz[:, towards, :] + factor * (pose_xyz[:, 0, :] - pose_xyz[:, towards, :])
    return out


class STB(Hand21KeypointDataset):
    def __init__(self, root, split='train', task='noobject', download=True, **kwargs):
        root = os.path.join(root, "STB")
        assert split in ['train', 'test', 'all']
        self.split = split
        seqs = SEQUENCES[2:] if split == 'train' else SEQUENCES[:2] if split == 'test' else SEQUENCES
        super().__init__(root, self.get_samples(root, seqs), **kwargs)

    def __getitem__(self, index):
        sample = self.samples[index]
        image = Image.open(os.path.join(self.root, sample['name']))
        keypoint2d = np.array(sample['keypoint2d'])
        Zc = np.array(sample['keypoint3d'])[:, 2]
        w, h = image.size
        left, upper, right, lower = scale_box(get_bounding_box(np.array(sample['keypoint2d2'])), w, h, 1.6)
        image, keypoint2d = crop(image, upper, left, lower - upper, right - left, keypoint2d)
        image, data = self.transforms(image, keypoint2d=keypoint2d, intrinsic_matrix=np.array(sample['intrinsic_matrix']))
        keypoint2d, K = data['keypoint2d'], data['intrinsic_matrix']
        keypoint3d_camera = keypoint2d_to_3d(keypoint2d, K, Zc)
        target, target_weight = self._labels(keypoint2d, np.ones((self.num_keypoints, 1), dtype=np.float32))
        pose, _ = self._normalised_pose(keypoint3d_camera)
        meta = {'image': sample['name'], 'keypoint2d': keypoint2d, 'keypoint3d': pose, 'z': keypoint3d_camera[:, 2],
                'keypoint3d_camera': keypoint3d_camera, 'cam_param': K, 'image_ema': data.get('image_ema', image)}
        return image, target, target_weight, meta

    def get_samples(self, root, image_list):
        import scipy.io as sio
        samples = []
        for seq in image_list:
            mat = sio.loadmat(os.path.join(root, "labels", seq + "_SK.mat"))
            poses = self.SK_xyz_depth2color(mat["handPara"].transpose((2, 1, 0)), SK_trans_vec, SK_rot)
            poses = poses[:, _HAND_INDEX, :] / 10.0
            wrist9, wrist13 = self.palm2wrist(poses), self.palm2wrist0(poses)
            for i in range(poses.shape[0]):
                samples.append({'name': os.path.join(seq, "SK_color_%d.png" % i),
                                'keypoint2d': keypoint3d_to_2d(wrist9[i], intrinsic_matrix0),
                                'keypoint2d2': keypoint3d_to_2d(wrist13[i], intrinsic_matrix0),     # (crop box only)
                                'keypoint3d': wrist9[i], 'intrinsic_matrix': intrinsic_matrix0})
        return samples

    def palm2wrist(self, pose_xyz):
        return _push_root(pose_xyz, 9, 2.1)

    def palm2wrist0(self, pose_xyz):
        return _push_root(pose_xyz, 13, 2.3)

    def SK_xyz_depth2color(self, depth_xyz, trans_vec, rot_mx):
        """(N,21,3) depth-camera coordinates -> colour-camera coordinates."""
        return (depth_xyz - np.asarray(trans_vec)).dot(rot_mx)


STBx1 = STB      # the reference ships a byte-identical second copy under this name (uda/dataset/STBx1.py)
