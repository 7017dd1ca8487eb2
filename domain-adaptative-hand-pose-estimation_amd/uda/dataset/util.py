"""Geometry helpers of the key-point data layer (reference ``uda/dataset/util.py``: ``generate_target`` :9-68,
``keypoint2d_to_3d`` :72-76, ``keypoint3d_to_2d`` :79-83, ``scale_box`` :86-112, ``get_bounding_box`` :115-121,
``area`` / ``intersection`` :136-143).  numpy only: the reference's cv2 / scipy imports serve its unused helpers."""
import numpy as np

from utils.synthetic import generate_target  # noqa: F401  (the label generator is shared with the synthetic batches)


def keypoint2d_to_3d(keypoint2d, intrinsic_matrix, Zc):
    """Back-project pixel coordinates (K,2) with depths Zc (K,) through the camera matrix: (K,3) camera coordinates."""
    homog = np.hstack([np.asarray(keypoint2d, dtype=np.float64), np.ones((len(keypoint2d), 1))])
    rays = np.linalg.inv(intrinsic_matrix) @ (homog.T * Zc)
    return rays.T


def keypoint3d_to_2d(keypoint3d, intrinsic_matrix):
    """Project camera coordinates (K,3) to pixels (K,2)."""
    proj = (np.asarray(intrinsic_matrix) @ np.asarray(keypoint3d).T).T
    return proj[:, :2] / proj[:, 2:3]


def get_bounding_box(keypoint2d):
    """(left, upper, right, lower) of a (K,2) point set."""
    xs, ys = keypoint2d[:, 0], keypoint2d[:, 1]
    return np.min(xs), np.min(ys), np.max(xs), np.max(ys)


def scale_box(box, image_width, image_height, scale):
    """Square box around the centre of `box` with side scale * max(w, h) (capped by the image), shifted back inside the
    image; integer pixel bounds, inclusive."""
    left, upper, right, lower = box
    cx, cy = (left + right) / 2, (upper + lower) / 2
    side = min(round(scale * max(right - left, lower - upper)), min(image_width, image_height))
    left = round(cx - side / 2)
    upper = round(cy - side / 2)
    left = min(max(left, 0), image_width - side)
    upper = min(max(upper, 0), image_height - side)
    return left, upper, left + side - 1, upper + side - 1


def area(left, upper, right, lower):
    return max(right - left + 1, 0) * max(lower - upper + 1, 0)


def intersection(box_a, box_b):
    return max(box_a[0], box_b[0]), max(box_a[1], box_b[1]), min(box_a[2], box_b[2]), min(box_a[3], box_b[3])
