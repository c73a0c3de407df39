"""NWPU-RESISC45 classification dataset (reference: src/myrtle_vision/datasets/resisc45.py)."""
import os
import random

import torch.utils.data
from PIL import Image

from myrtle_vision.datasets.transforms import build_transform
from myrtle_vision.utils.utils import get_label_number, load_imagepaths_and_labels


class Resisc45(torch.utils.data.Dataset):
    def __init__(self, mode, dataset_path, imagepaths, label_map_path, transform_config):
        if mode not in ["train", "eval"]:
            raise ValueError(f"unknown mode={mode}")
        self.mode, self.dataset_path, self.label_map_path = mode, dataset_path, label_map_path
        self.imagepaths_and_labels = load_imagepaths_and_labels(dataset_path, imagepaths)
        if mode == "train":
            random.shuffle(self.imagepaths_and_labels)
        self.transform = build_transform(transform_config)

    def __getitem__(self, index):
        path, text_label = self.imagepaths_and_labels[index]
        img, _ = self.transform(Image.open(os.path.join(self.dataset_path, path)))
        return img, get_label_number(self.dataset_path, self.label_map_path, text_label)

    def __len__(self):
        return len(self.imagepaths_and_labels)
