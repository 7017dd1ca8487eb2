"""Key-point aware image transforms (reference ``uda/dataset/keypoint_detection.py``) on PIL + numpy + torch only.

The reference wraps torchvision's PIL transforms; torchvision is not a dependency here, so the few operations it needs
are written out (``ToTensor``, ``Normalize``, ``ColorJitter``: torchvision's PIL semantics; resize / crop / flip / rotate:
the PIL calls torchvision's functional API makes).  Every transform maps ``(image, **labels) -> (image, labels)`` and
keeps ``keypoint2d`` (K,2) and ``intrinsic_matrix`` (3,3) consistent with the pixels; ``Compose`` adds the
``image_ema`` side output (a normalised tensor copy taken right after ``RandomResizedCrop``, reference :171-181)."""
import math
import numbers
import random
import warnings

import numpy as np
import torch
from PIL import Image, ImageEnhance, ImageFilter

BILINEAR = Image.BILINEAR


# ------------------------------------------------------------------ functional layer
def _resize_pil(image, size, interpolation=BILINEAR):
    """int size: the shorter edge becomes `size` (aspect kept); (h, w): exact."""
    if isinstance(size, int):
        w, h = image.size
        if (w <= h and w == size) or (h <= w and h == size):
            return image
        if w < h:
            return image.resize((size, int(size * h / w)), interpolation)
        return image.resize((int(size * w / h), size), interpolation)
    return image.resize((size[1], size[0]), interpolation)


def resize(image, size, interpolation=BILINEAR, keypoint2d=None, intrinsic_matrix=None):
    width, height = image.size
    assert width == height, 'resize expects the square crops the datasets produce'
    factor = float(size) / float(width)
    image = _resize_pil(image, size, interpolation)
    keypoint2d = np.array(keypoint2d, dtype=np.float64, copy=True) * factor
    K = np.array(intrinsic_matrix, dtype=np.float64, copy=True)
    K[0, 0] *= factor; K[0, 2] *= factor; K[1, 1] *= factor; K[1, 2] *= factor
    return image, keypoint2d, K


def crop(image, top, left, height, width, keypoint2d):
    image = image.crop((left, top, left + width, top + height))
    keypoint2d = np.array(keypoint2d, dtype=np.float64, copy=True)
    keypoint2d[:, 0] -= left
    keypoint2d[:, 1] -= top
    return image, keypoint2d


def resized_crop(img, top, left, height, width, size, interpolation=BILINEAR, keypoint2d=None, intrinsic_matrix=None):
    assert isinstance(img, Image.Image), 'img should be PIL Image'
    img, keypoint2d = crop(img, top, left, height, width, keypoint2d)
    return resize(img, size, interpolation, keypoint2d, intrinsic_matrix)


def center_crop(image, output_size, keypoint2d):
    width, height = image.size
    ch, cw = output_size
    return crop(image, int(round((height - ch) / 2.)), int(round((width - cw) / 2.)), ch, cw, keypoint2d)


def hflip(image, keypoint2d):
    width, _ = image.size
    keypoint2d = np.array(keypoint2d, dtype=np.float64, copy=True)
    keypoint2d[:, 0] = width - 1. - keypoint2d[:, 0]
    return image.transpose(Image.FLIP_LEFT_RIGHT), keypoint2d


def rotate(image, angle, keypoint2d):
    """Counter-clockwise by `angle` degrees about the image centre (PIL convention), same canvas."""
    image = image.rotate(angle)
    a = -np.deg2rad(angle)
    R = np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    width, height = image.size
    centre = np.array([width / 2, height / 2])
    keypoint2d = (np.asarray(keypoint2d, dtype=np.float64) - centre) @ R.T + centre
    return image, keypoint2d


def resize_pad(img, keypoint2d, size, interpolation=BILINEAR):
    """Longer edge -> `size`, zero padding centred on the shorter one."""
    w, h = img.size
    if w < h:
        ow, oh = int(size * w / h), size
        pads = ((0, 0), (math.floor((size - ow) / 2), math.ceil((size - ow) / 2)), (0, 0))
        keypoint2d = keypoint2d * oh / h
        keypoint2d[:, 0] += (size - ow) / 2
    else:
        ow, oh = size, int(size * h / w)
        pads = ((math.floor((size - oh) / 2), math.ceil((size - oh) / 2)), (0, 0), (0, 0))
        keypoint2d = keypoint2d * ow / w
        keypoint2d[:, 1] += (size - oh) / 2
        keypoint2d[:, 0] += (size - ow) / 2
    arr = np.pad(np.asarray(img.resize((ow, oh), interpolation)), pads, 'constant', constant_values=0)
    return Image.fromarray(arr), keypoint2d


def to_tensor(image):
    """PIL RGB / L image (or HxWxC uint8 array) -> float32 tensor CxHxW in [0, 1]."""
    arr = np.asarray(image)
    if arr.ndim == 2:
        arr = arr[:, :, None]
    t = torch.from_numpy(np.ascontiguousarray(arr.transpose(2, 0, 1)))
    return t.float().div(255) if t.dtype == torch.uint8 else t.float()


# ------------------------------------------------------------------ label-preserving pixel transforms
class ToTensor:
    def __call__(self, image, **kwargs):
        return to_tensor(image), kwargs


class Normalize:
    def __init__(self, mean, std):
        self.mean = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, image, **kwargs):
        return (image - self.mean) / self.std, kwargs


class ColorJitter:
    """Brightness / contrast / saturation / hue factors drawn uniformly, applied in a random order (torchvision's PIL
    semantics: ImageEnhance for the first three, a hue rotation in HSV for the last)."""

    def __init__(self, brightness=0, contrast=0, saturation=0, hue=0):
        rng = lambda v: None if not v else (max(0.0, 1 - v), 1 + v)
        self.brightness, self.contrast, self.saturation = rng(brightness), rng(contrast), rng(saturation)
        self.hue = None if not hue else (-hue, hue)

    @staticmethod
    def _hue(image, factor):
        h, s, v = image.convert('HSV').split()
        shifted = (np.asarray(h, dtype=np.uint8).astype(np.int16) + int(factor * 255)) % 256
        return Image.merge('HSV', (Image.fromarray(shifted.astype(np.uint8), 'L'), s, v)).convert(image.mode)

    def __call__(self, image, **kwargs):
        ops = []
        if self.brightness:
            f = random.uniform(*self.brightness); ops.append(lambda im, f=f: ImageEnhance.Brightness(im).enhance(f))
        if self.contrast:
            f = random.uniform(*self.contrast); ops.append(lambda im, f=f: ImageEnhance.Contrast(im).enhance(f))
        if self.saturation:
            f = random.uniform(*self.saturation); ops.append(lambda im, f=f: ImageEnhance.Color(im).enhance(f))
        if self.hue:
            f = random.uniform(*self.hue); ops.append(lambda im, f=f: self._hue(im, f))
        random.shuffle(ops)
        for op in ops:
            image = op(image)
        return image, kwargs


class GaussianBlur:
    def __init__(self, low=0, high=0.8):
        self.low, self.high = low, high

    def __call__(self, image, **kwargs):
        radius = np.random.uniform(low=self.low, high=self.high)
        return image.filter(ImageFilter.GaussianBlur(radius)), kwargs


# ------------------------------------------------------------------ geometric transforms
class Compose:
    """Chain of transforms; right after a ``RandomResizedCrop`` a normalised tensor copy of the image is stored as
    ``image_ema`` (the weakly augmented view the reference hands to its EMA branch)."""
    EMA_MEAN, EMA_STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]

    def __init__(self, transforms):
        self.transforms = transforms
        self._ema_norm = Normalize(self.EMA_MEAN, self.EMA_STD)

    def __call__(self, image, **kwargs):
        for t in self.transforms:
            image, kwargs = t(image, **kwargs)
            if type(t).__name__ == 'RandomResizedCrop':
                kwargs['image_ema'] = self._ema_norm(to_tensor(image.copy()))[0]
        return image, kwargs


class Resize:
    def __init__(self, size, interpolation=BILINEAR):
        assert isinstance(size, int)
        self.size, self.interpolation = size, interpolation

    def __call__(self, image, keypoint2d, intrinsic_matrix, **kwargs):
        image, keypoint2d, intrinsic_matrix = resize(image, self.size, self.interpolation, keypoint2d, intrinsic_matrix)
        kwargs.update(keypoint2d=keypoint2d, intrinsic_matrix=intrinsic_matrix)
        if 'depth' in kwargs:
            kwargs['depth'] = _resize_pil(kwargs['depth'], self.size)
        return image, kwargs


class ResizePad:
    def __init__(self, size, interpolation=BILINEAR):
        self.size, self.interpolation = size, interpolation

    def __call__(self, img, keypoint2d, **kwargs):
        image, keypoint2d = resize_pad(img, keypoint2d, self.size, self.interpolation)
        kwargs.update(keypoint2d=keypoint2d)
        return image, kwargs


class CenterCrop:
    def __init__(self, size):
        self.size = (int(size), int(size)) if isinstance(size, numbers.Number) else size

    def __call__(self, image, keypoint2d, **kwargs):
        image, keypoint2d = center_crop(image, self.size, keypoint2d)
        kwargs.update(keypoint2d=keypoint2d)
        if 'depth' in kwargs:
            kwargs['depth'] = center_crop(kwargs['depth'], self.size, np.zeros((1, 2)))[0]
        return image, kwargs


class RandomRotation:
    def __init__(self, degrees):
        if isinstance(degrees, numbers.Number):
            if degrees < 0:
                raise ValueError("If degrees is a single number, it must be positive.")
            degrees = (-degrees, degrees)
        elif len(degrees) != 2:
            raise ValueError("If degrees is a sequence, it must be of len 2.")
        self.degrees = tuple(degrees)

    @staticmethod
    def get_params(degrees):
        return random.uniform(degrees[0], degrees[1])

    def __call__(self, image, keypoint2d, **kwargs):
        angle = self.get_params(self.degrees)
        image, keypoint2d = rotate(image, angle, keypoint2d)
        kwargs.update(keypoint2d=keypoint2d)
        if 'depth' in kwargs:
            kwargs['depth'] = kwargs['depth'].rotate(angle)
        return image, kwargs


class RandomResizedCrop:
    """Square crop covering a random fraction `scale` of the image area (aspect ratio 1), resized to `size`."""

    def __init__(self, size, scale=(0.6, 1.3), interpolation=BILINEAR):
        if scale[0] > scale[1]:
            warnings.warn("range should be of kind (min, max)")
        self.size, self.scale, self.interpolation = size, scale, interpolation

    @staticmethod
    def get_params(img, scale):
        width, height = img.size
        for _ in range(10):
            side = int(round(math.sqrt(random.uniform(*scale) * height * width)))
            if 0 < side <= width and side <= height:
                return random.randint(0, height - side), random.randint(0, width - side), side, side
        return 0, 0, height, width          # no admissible draw: the whole image

    def __call__(self, image, keypoint2d, intrinsic_matrix, **kwargs):
        i, j, h, w = self.get_params(image, self.scale)
        image, keypoint2d, intrinsic_matrix = resized_crop(image, i, j, h, w, self.size, self.interpolation, keypoint2d,
                                                           intrinsic_matrix)
        kwargs.update(keypoint2d=keypoint2d, intrinsic_matrix=intrinsic_matrix)
        if 'depth' in kwargs:
            kwargs['depth'] = _resize_pil(kwargs['depth'].crop((j, i, j + w, i + h)), self.size, self.interpolation)
        return image, kwargs


class RandomApply:
    def __init__(self, transforms, p=0.5):
        self.transforms, self.p = transforms, p

    def __call__(self, image, **kwargs):
        if self.p < random.random():
            return image, kwargs
        for t in self.transforms:
            image, kwargs = t(image, **kwargs)
        return image, kwargs
