"""NWPU-RESISC45 classification dataset (reference: src/myrtle_vision/datasets/resisc45.py)."""
import os
import random

import torch.utils.data
from PIL import Image

from myrtle_vision.datasets.transforms import build_transform
from myrtle_vision.utils.utils import get_label_number, load_imagepaths_and_labels


class Resisc45(torch.utils.data.Dataset):
    def __init__(self, mode, dataset_path, imagepaths, label_map_path, transform_config, device_plan=None):
        if mode not in ["train", "eval"]:
            raise ValueError(f"unknown mode={mode}")
        self.mode, self.dataset_path, self.label_map_path = mode, dataset_path, label_map_path
        self.imagepaths_and_labels = load_imagepaths_and_labels(dataset_path, imagepaths)
        if mode == "train":
            random.shuffle(self.imagepaths_and_labels)
        # device_plan (datasets/device_transforms.DevicePlan): the worker only decodes and draws the random parameters;
        # crop/resize/flip/normalize run on the GPU for the whole batch (extension)
        self.device_plan = device_plan
        self.transform = build_transform(transform_config) if device_plan is None else None

    def __getitem__(self, index):
        path, text_label = self.imagepaths_and_labels[index]
        label = get_label_number(self.dataset_path, self.label_map_path, text_label)
        pil = Image.open(os.path.join(self.dataset_path, path))
        if self.device_plan is not None:
            return self.device_plan(pil), label
        img, _ = self.transform(pil)
        return img, label

    def __len__(self):
        return len(self.imagepaths_and_labels)
