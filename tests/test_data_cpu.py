"""CPU tests of the data layer: synthetic datasets in the reference's on-disk layout, JSON-driven transforms."""
import json
import os
import random

import numpy as np
import torch
from PIL import Image

from conftest import ROOT


def test_resisc45_layout_and_transforms(tmp_path):
    from myrtle_vision.datasets.resisc45 import Resisc45
    from myrtle_vision.datasets.synthetic import make_resisc45
    from myrtle_vision.utils.utils import get_label_list, load_imagepaths_and_labels
    root = make_resisc45(str(tmp_path / "NWPU-RESISC45"), classes=6, per_class=5)
    cfg = json.load(open(os.path.join(ROOT, "classification", "data_configs", "data_config.json")))
    pairs = load_imagepaths_and_labels(root, cfg["train_files"])
    assert all(p.startswith("images/") and lab == p.split("/")[1] for p, lab in pairs)
    assert get_label_list(root, cfg["label_map"]) == [f"class{c:02d}" for c in range(6)]
    random.seed(0)
    train = Resisc45("train", root, cfg["train_files"], cfg["label_map"], cfg["transform_ops_train"])
    val = Resisc45("eval", root, cfg["valid_files"], cfg["label_map"], cfg["transform_ops_val"])
    x, y = train[0]
    assert x.shape == (3, 224, 224) and x.dtype == torch.float32 and -1.0 <= float(x.min()) and float(x.max()) <= 1.0
    assert isinstance(y, int) and 0 <= y < 6
    xv, _ = val[0]
    # eval transform is deterministic: Resize(224) bilinear + Normalize(0.5, 0.5)
    p = os.path.join(root, val.imagepaths_and_labels[0][0])
    ref = np.asarray(Image.open(p).resize((224, 224), Image.Resampling.BILINEAR), dtype=np.float32) / 255.0
    assert torch.allclose(xv, torch.from_numpy((ref - 0.5) / 0.5).permute(2, 0, 1), atol=1e-6)


def test_dlrsd_paired_transforms(tmp_path):
    from myrtle_vision.datasets.dlrsd import Dlrsd, collate_both
    from myrtle_vision.datasets.synthetic import make_dlrsd
    root = make_dlrsd(str(tmp_path / "DLRSD_dataset"), count=8)
    cfg = json.load(open(os.path.join(ROOT, "segmentation", "data_configs", "data_config.json")))
    random.seed(1)
    ds = Dlrsd("train", root, cfg["train_files"], cfg["label_map"], cfg["transform_ops_train"])
    imgs, masks = collate_both([ds[0], ds[1]])
    assert imgs.shape == (2, 3, 224, 224) and masks.shape == (2, 224, 224) and masks.dtype == torch.int64
    assert int(masks.min()) >= 0 and int(masks.max()) <= 16                # PNG value - 1 (reference dlrsd.py:80)


def test_miou_matches_definition():
    from myrtle_vision.utils.miou import MIoU
    g = torch.Generator().manual_seed(0)
    pred, gt = torch.randint(0, 5, (3, 32, 32), generator=g), torch.randint(0, 5, (3, 32, 32), generator=g)
    m = MIoU(5, "cpu")
    for i in range(3):
        m.add_img(pred[i], gt[i])
    want = np.mean([((pred == c) & (gt == c)).sum().item() / (((pred == c) | (gt == c)).sum().item()) for c in range(5)])
    assert abs(m.get_miou() - want) < 1e-12


def test_miou_drops_out_of_range_like_histc():
    """reference utils/miou.py:35-40: torch.histc(min=0, max=C-1) ignores unlabeled (-1) and out-of-range pixels"""
    from myrtle_vision.utils.miou import intersect_and_union
    g = torch.Generator().manual_seed(1)
    pred, gt = torch.randint(0, 5, (64, 64), generator=g), torch.randint(-1, 7, (64, 64), generator=g)
    got = intersect_and_union(pred, gt, 5)
    inter = pred[pred == gt]
    want = [torch.histc(t.float(), bins=5, min=0, max=4).double() for t in (inter, pred, gt)]
    assert torch.equal(got[0], want[0]) and torch.equal(got[2], want[1]) and torch.equal(got[3], want[2])
    assert torch.equal(got[1], want[1] + want[2] - want[0])
