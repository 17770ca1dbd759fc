"""Image / box transforms and the augmentations the CenterNet dataset mapper uses (host side, numpy + PIL).

Reference: detectron2/data/transforms/augmentation_impl.py (ResizeShortestEdge :123-173, RandomApply :26-68,
RandomContrast :406-431, RandomBrightness :434-457, RandomSaturation :460-486, RandomLighting :489-515, RandomFlip
:71-101), transform.py (ResizeTransform :83-136).  The Transform base classes (`Transform`, `TransformList`,
`BlendTransform`, `HFlipTransform`, `NoOpTransform`) come from the third-party package fvcore
(`fvcore.transforms.transform`, not in the reference tree and unpinned): restated here from their published behaviour --
`apply_box` maps the four corners and takes the axis-aligned hull; `BlendTransform` computes
`src_weight * src_image + dst_weight * img` in float32 and clips uint8 images to [0, 255]."""
import sys

import numpy as np
from PIL import Image


class Transform:
    def apply_image(self, img):
        raise NotImplementedError

    def apply_coords(self, coords):
        raise NotImplementedError

    def apply_box(self, box):
        """box: [N,4] XYXY -> axis-aligned hull of the transformed corners"""
        box = np.asarray(box, dtype=np.float64).reshape(-1, 4)
        idxs = np.array([(0, 1), (2, 1), (0, 3), (2, 3)]).flatten()
        coords = box[:, idxs].reshape(-1, 2)
        coords = self.apply_coords(coords).reshape((-1, 4, 2))
        minxy, maxxy = coords.min(axis=1), coords.max(axis=1)
        return np.concatenate((minxy, maxxy), axis=1)


class NoOpTransform(Transform):
    def apply_image(self, img):
        return img

    def apply_coords(self, coords):
        return coords


class ResizeTransform(Transform):
    def __init__(self, h, w, new_h, new_w, interp=None):
        self.h, self.w, self.new_h, self.new_w = h, w, new_h, new_w
        self.interp = Image.BILINEAR if interp is None else interp

    def apply_image(self, img, interp=None):
        assert img.shape[:2] == (self.h, self.w), (img.shape, self.h, self.w)
        if img.dtype != np.uint8:
            raise NotImplementedError("the mapper resizes uint8 images (transform.py:105-109); float images are not on the path")
        pil = Image.fromarray(img).resize((self.new_w, self.new_h), interp if interp is not None else self.interp)
        return np.asarray(pil)

    def apply_coords(self, coords):
        coords = np.asarray(coords, dtype=np.float64)
        coords[:, 0] = coords[:, 0] * (self.new_w * 1.0 / self.w)
        coords[:, 1] = coords[:, 1] * (self.new_h * 1.0 / self.h)
        return coords


class HFlipTransform(Transform):
    def __init__(self, width):
        self.width = width

    def apply_image(self, img):
        return np.flip(img, axis=1)

    def apply_coords(self, coords):
        coords = np.asarray(coords, dtype=np.float64)
        coords[:, 0] = self.width - coords[:, 0]
        return coords


class BlendTransform(Transform):
    def __init__(self, src_image, src_weight, dst_weight):
        self.src_image, self.src_weight, self.dst_weight = src_image, src_weight, dst_weight

    def apply_image(self, img, interp=None):
        if img.dtype == np.uint8:
            src = np.asarray(self.src_image)
            if src.size in (1, img.shape[-1]) and img.ndim == 3:
                # per-channel affine map of a byte: tabulate the 256 results with the very expression below (same dtypes,
                # same rounding) and look the pixels up -- 5 float passes over the image become one gather
                ramp = np.arange(256, dtype=np.uint8).reshape(256, 1).repeat(img.shape[-1], 1)
                lut = np.clip(self.src_weight * src.reshape(-1) + self.dst_weight * ramp.astype(np.float32), 0, 255).astype(np.uint8)
                if img.shape[-1] == 3:     # PIL applies a 3 x 256 table to an RGB image in C
                    pil = Image.fromarray(np.ascontiguousarray(img)).point(lut.T.reshape(-1).tolist())
                    return np.asarray(pil)
                return np.stack([np.take(lut[:, c], img[..., c]) for c in range(img.shape[-1])], axis=-1)
            out = self.src_weight * self.src_image + self.dst_weight * img.astype(np.float32)
            return np.clip(out, 0, 255).astype(np.uint8)
        return self.src_weight * self.src_image + self.dst_weight * img

    def apply_coords(self, coords):
        return coords


class TransformList(Transform):
    def __init__(self, transforms):
        self.transforms = []
        for t in transforms:     # flattened, no-ops dropped (fvcore does the same)
            if isinstance(t, TransformList):
                self.transforms.extend(t.transforms)
            elif not isinstance(t, NoOpTransform):
                self.transforms.append(t)

    def apply_image(self, img):
        for t in self.transforms:
            img = t.apply_image(img)
        return img

    def apply_coords(self, coords):
        for t in self.transforms:
            coords = t.apply_coords(coords)
        return coords

    def apply_box(self, box):
        for t in self.transforms:
            box = t.apply_box(box)
        return np.asarray(box, dtype=np.float64).reshape(-1, 4)

    def __len__(self):
        return len(self.transforms)


# ------------------------------------------------------------------------------------------ augmentations
class Augmentation:
    """get_transform(image) -> Transform; draws from numpy's global RNG like the reference"""

    def get_transform(self, image):
        raise NotImplementedError


class ResizeShortestEdge(Augmentation):
    def __init__(self, short_edge_length, max_size=sys.maxsize, sample_style="range", interp=Image.BILINEAR):
        assert sample_style in ("range", "choice"), sample_style
        self.is_range = sample_style == "range"
        if isinstance(short_edge_length, int):
            short_edge_length = (short_edge_length, short_edge_length)
        if self.is_range:
            assert len(short_edge_length) == 2, short_edge_length
        self.short_edge_length, self.max_size, self.interp = tuple(short_edge_length), max_size, interp

    @staticmethod
    def output_size(h, w, size, max_size):
        """the size rule of augmentation_impl.py:162-172"""
        scale = size * 1.0 / min(h, w)
        if h < w:
            newh, neww = size, scale * w
        else:
            newh, neww = scale * h, size
        if max(newh, neww) > max_size:
            scale = max_size * 1.0 / max(newh, neww)
            newh, neww = newh * scale, neww * scale
        return int(newh + 0.5), int(neww + 0.5)

    def get_transform(self, image):
        h, w = image.shape[:2]
        if self.is_range:
            size = np.random.randint(self.short_edge_length[0], self.short_edge_length[1] + 1)
        else:
            size = np.random.choice(self.short_edge_length)
        if size == 0:
            return NoOpTransform()
        newh, neww = self.output_size(h, w, int(size), self.max_size)
        return ResizeTransform(h, w, newh, neww, self.interp)


class RandomApply(Augmentation):
    def __init__(self, aug, prob=0.5):
        assert 0.0 <= prob <= 1.0
        self.aug, self.prob = aug, prob

    def get_transform(self, image):
        if np.random.uniform(0, 1) < self.prob:
            return self.aug.get_transform(image)
        return NoOpTransform()


class RandomFlip(Augmentation):
    def __init__(self, prob=0.5):
        self.prob = prob

    def get_transform(self, image):
        if np.random.uniform(0, 1) < self.prob:
            return HFlipTransform(image.shape[1])
        return NoOpTransform()


class _RandomIntensity(Augmentation):
    def __init__(self, intensity_min, intensity_max):
        self.intensity_min, self.intensity_max = intensity_min, intensity_max


class RandomContrast(_RandomIntensity):
    def get_transform(self, image):
        w = np.random.uniform(self.intensity_min, self.intensity_max)
        return BlendTransform(src_image=image.mean(), src_weight=1 - w, dst_weight=w)


class RandomBrightness(_RandomIntensity):
    def get_transform(self, image):
        w = np.random.uniform(self.intensity_min, self.intensity_max)
        return BlendTransform(src_image=0, src_weight=1 - w, dst_weight=w)


class RandomSaturation(_RandomIntensity):
    def get_transform(self, image):
        assert image.shape[-1] == 3, "RandomSaturation only works on RGB images"
        w = np.random.uniform(self.intensity_min, self.intensity_max)
        # image.dot([0.299, 0.587, 0.114]) in the reference: the same float64 sum channel by channel (numpy's generic
        # uint8 x float64 dot is 6x slower)
        grayscale = (image[..., 0] * 0.299 + image[..., 1] * 0.587 + image[..., 2] * 0.114)[:, :, np.newaxis]
        return BlendTransform(src_image=grayscale, src_weight=1 - w, dst_weight=w)


class RandomLighting(Augmentation):
    """AlexNet's PCA lighting noise with the fixed ImageNet eigen-decomposition (augmentation_impl.py:505-508)"""

    def __init__(self, scale):
        self.scale = scale
        self.eigen_vecs = np.array([[-0.5675, 0.7192, 0.4009], [-0.5808, -0.0045, -0.8140], [-0.5836, -0.6948, 0.4203]])
        self.eigen_vals = np.array([0.2175, 0.0188, 0.0045])

    def get_transform(self, image):
        assert image.shape[-1] == 3, "RandomLighting only works on RGB images"
        weights = np.random.normal(scale=self.scale, size=3)
        return BlendTransform(src_image=self.eigen_vecs.dot(weights * self.eigen_vals), src_weight=1.0, dst_weight=1.0)


def apply_augmentations(augmentations, image):
    """run the augmentations in order on `image` (each sees the output of the previous one, like
    StandardAugInput.apply_augmentations); returns (image, TransformList)"""
    tfms = []
    for aug in augmentations:
        t = aug.get_transform(image) if isinstance(aug, Augmentation) else aug
        image = t.apply_image(image)
        tfms.append(t)
    return image, TransformList(tfms)
