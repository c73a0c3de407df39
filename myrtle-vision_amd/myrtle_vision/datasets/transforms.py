"""torchvision-free transforms driven by the reference's JSON ``transform_ops_*`` sections.

Same operations and parameters as the reference builds from torchvision (datasets/resisc45.py:40-69,
datasets/dlrsd.py:39-66, transforms/segmentation.py): Resize((s, s)) bilinear (NEAREST for masks),
RandomResizedCrop(s) with scale (0.08, 1) and ratio (3/4, 4/3), CenterCrop, RandomHorizontalFlip(p=0.5), ToTensor,
Normalize.  Every op takes and returns ``(image, mask_or_None)`` so classification and segmentation share one
pipeline; geometric ops apply the same parameters to both.
"""
import math
import random

import numpy as np
import torch
from PIL import Image

_BILINEAR = Image.Resampling.BILINEAR
_NEAREST = Image.Resampling.NEAREST


class Resize:
    def __init__(self, size):
        self.size = (size, size)

    def __call__(self, img, mask=None):
        return img.resize(self.size, _BILINEAR), (mask.resize(self.size, _NEAREST) if mask is not None else None)


class RandomResizedCrop:
    def __init__(self, size, scale=(0.08, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0)):
        self.size, self.scale, self.ratio = (size, size), scale, ratio

    def _params(self, w, h):
        area = w * h
        log_ratio = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        for _ in range(10):
            target = area * random.uniform(*self.scale)
            ar = math.exp(random.uniform(*log_ratio))
            cw, ch = int(round(math.sqrt(target * ar))), int(round(math.sqrt(target / ar)))
            if 0 < cw <= w and 0 < ch <= h:
                return random.randint(0, h - ch), random.randint(0, w - cw), ch, cw
        in_ratio = w / h                                     # fallback: central crop at the nearest allowed ratio
        if in_ratio < self.ratio[0]:
            cw, ch = w, int(round(w / self.ratio[0]))
        elif in_ratio > self.ratio[1]:
            ch, cw = h, int(round(h * self.ratio[1]))
        else:
            cw, ch = w, h
        return (h - ch) // 2, (w - cw) // 2, ch, cw

    def __call__(self, img, mask=None):
        top, left, ch, cw = self._params(*img.size)
        box = (left, top, left + cw, top + ch)
        # torchvision F.resized_crop: crop, THEN resize (the filter support clamps at the crop edge; PIL's resize(box=)
        # would reach outside the box)
        img = img.crop(box).resize(self.size, _BILINEAR)
        if mask is not None:
            mask = mask.crop(box).resize(self.size, _NEAREST)
        return img, mask


class CenterCrop:
    def __init__(self, size):
        self.size = size

    def __call__(self, img, mask=None):
        w, h = img.size
        left, top = (w - self.size) // 2, (h - self.size) // 2
        box = (left, top, left + self.size, top + self.size)
        return img.crop(box), (mask.crop(box) if mask is not None else None)


class RandomHorizontalFlip:
    def __call__(self, img, mask=None):
        if random.random() < 0.5:
            img = img.transpose(Image.Transpose.FLIP_LEFT_RIGHT)
            if mask is not None:
                mask = mask.transpose(Image.Transpose.FLIP_LEFT_RIGHT)
        return img, mask


class ToTensorNormalize:
    """ToTensor (HWC uint8 -> CHW float in [0,1]) followed by the optional Normalize(mean, std)."""

    def __init__(self, mean=None, std=None):
        self.mean = None if mean is None else torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1)
        self.std = None if std is None else torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, img, mask=None):
        a = np.asarray(img.convert("RGB"), dtype=np.uint8)
        t = torch.from_numpy(a.copy()).permute(2, 0, 1).float().div_(255.0)
        if self.mean is not None:
            t = (t - self.mean) / self.std
        if mask is not None:
            mask = torch.from_numpy(np.asarray(mask, dtype=np.uint8).copy()).to(torch.int64)
        return t, mask


class Compose:
    def __init__(self, ops):
        self.ops = ops

    def __call__(self, img, mask=None):
        for op in self.ops:
            img, mask = op(img, mask)
        return img, mask


def build_transform(transform_config):
    """Order of the reference: Resize, RandomResizedCrop, CenterCrop, RandomHorizontalFlip, ToTensor, Normalize."""
    ops = []
    if "Resize" in transform_config:
        ops.append(Resize(transform_config["Resize"]))
    if "RandomResizedCrop" in transform_config:
        ops.append(RandomResizedCrop(transform_config["RandomResizedCrop"]))
    if "CenterCrop" in transform_config:
        ops.append(CenterCrop(transform_config["CenterCrop"]))
    if "RandomHorizontalFlip" in transform_config:
        ops.append(RandomHorizontalFlip())
    n = transform_config.get("Normalize")
    ops.append(ToTensorNormalize(n["Mean"], n["Std"]) if n else ToTensorNormalize())
    return Compose(ops)
