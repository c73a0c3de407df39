"""DLRSD segmentation dataset (reference: src/myrtle_vision/datasets/dlrsd.py).  Labels are PNG value - 1, i.e.
0..16 (reference dlrsd.py:80); image and mask receive the same geometric transform."""
import os
import random

import torch
import torch.utils.data
from PIL import Image

from myrtle_vision.datasets.transforms import build_transform
from myrtle_vision.utils.utils import load_imagepaths_and_segmaps


class Dlrsd(torch.utils.data.Dataset):
    def __init__(self, mode, dataset_path, imagepaths, label_map_path, transform_config, device_plan=None):
        if mode not in ["train", "eval", "test"]:
            raise ValueError(f"unknown mode={mode}")
        self.mode, self.dataset_path, self.label_map_path = mode, dataset_path, label_map_path
        self.imagepaths_and_segmaps = load_imagepaths_and_segmaps(dataset_path, imagepaths)
        if mode == "train":
            random.shuffle(self.imagepaths_and_segmaps)
        self.device_plan = device_plan                     # see Resisc45: geometric ops + normalisation on the GPU
        self.transform = build_transform(transform_config) if device_plan is None else None

    def __getitem__(self, index):
        img_path, seg_path = self.imagepaths_and_segmaps[index]
        image = Image.open(os.path.join(self.dataset_path, img_path))
        segmap = Image.open(os.path.join(self.dataset_path, seg_path))
        if self.device_plan is not None:
            return self.device_plan(image, segmap), None       # the "- 1" below is mv_mask_prepare's `add`
        image, segmap = self.transform(image, segmap)
        return image, segmap - 1

    def __len__(self):
        return len(self.imagepaths_and_segmaps)


def collate_both(batch):
    """reference transforms/segmentation.py ``collate_both``: stack images and masks."""
    return torch.stack([b[0] for b in batch]), torch.stack([b[1] for b in batch])
