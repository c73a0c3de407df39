"""Image preparation: the numpy oracle of Pillow's resampler is pinned against Pillow itself; the package's
vectorised table builder against the oracle; and (GPU) the HIP kernels against the Pillow pipeline of
datasets/transforms.py -- all bit for bit."""
import random

import numpy as np
import pytest
import torch
from PIL import Image

from oracle import resample_oracle as ro

BIL, NEA = Image.Resampling.BILINEAR, Image.Resampling.NEAREST
SIZES = [(256, 256, 224, 224), (256, 256, 256, 256), (73, 91, 224, 224), (300, 211, 64, 80), (600, 777, 224, 224), (17, 17, 224, 224),
         (224, 224, 223, 225)]


@pytest.mark.parametrize("h,w,oh,ow", SIZES)
def test_oracle_matches_pillow(h, w, oh, ow):
    rng = np.random.default_rng(h * 1000 + w)
    img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    want = np.asarray(Image.fromarray(img).resize((ow, oh), BIL))
    assert np.array_equal(ro.resize_bilinear_u8(img, (oh, ow)), want)
    m = rng.integers(0, 18, (h, w), dtype=np.uint8)
    assert np.array_equal(ro.resize_nearest_u8(m, (oh, ow)), np.asarray(Image.fromarray(m).resize((ow, oh), NEA)))


@pytest.mark.parametrize("n_in,n_out", [(256, 224), (224, 224), (72, 224), (1000, 224), (255, 7), (13, 224), (224, 223)])
def test_table_builder_matches_oracle(n_in, n_out):
    from myrtle_vision.datasets.device_transforms import bilinear_tables, nearest_table
    b, k = bilinear_tables(n_in, n_out)
    bo, ko = ro.bilinear_coeffs(n_in, 0.0, float(n_in), n_out)
    assert np.array_equal(b, bo) and np.array_equal(k, ko)
    assert np.array_equal(nearest_table(n_in, n_out), np.clip(ro.nearest_index(n_in, n_out), 0, n_in - 1))


def _cfg(kind):
    norm = {"Mean": [0.5, 0.5, 0.5], "Std": [0.5, 0.5, 0.5]}
    return {"train": {"RandomResizedCrop": 224, "RandomHorizontalFlip": None, "Normalize": norm},
            "val": {"Resize": 224, "Normalize": norm},
            "crop": {"Resize": 256, "CenterCrop": 224, "RandomHorizontalFlip": None, "Normalize": {"Mean": [0.485, 0.456, 0.406],
                                                                                                "Std": [0.229, 0.224, 0.225]}},
            "plain": {"RandomHorizontalFlip": None},
            # segmentation/data_configs/data_config.json transform_ops_train: TWO resamplings (Resize, then the crop's)
            "seg_train": {"Resize": 224, "RandomResizedCrop": 224, "RandomHorizontalFlip": None, "Normalize": norm}}[kind]


def _resample(raw, kh, bh, kv, bv):
    """Pillow's two-pass 8-bit resize through the plan's tables (int64 arithmetic): uint8 HWC -> uint8 [oh, ow, 3]."""
    oh, ow = kv.shape[0], kh.shape[0]
    half = 1 << 21
    tmp = np.zeros((raw.shape[0], ow, 3), np.int64)                          # horizontal pass, rounded to uint8
    for X in range(ow):
        acc = np.full((raw.shape[0], 3), half, np.int64)
        for x in range(bh[X, 1]):
            acc += raw[:, bh[X, 0] + x] * int(kh[X, x])
        tmp[:, X] = np.clip(acc >> 22, 0, 255)
    out = np.zeros((oh, ow, 3), np.int64)
    for Y in range(oh):
        acc = np.full((ow, 3), half, np.int64)
        for y in range(bv[Y, 1]):
            acc += tmp[bv[Y, 0] + y] * int(kv[Y, y])
        out[Y] = np.clip(acc >> 22, 0, 255)
    return out


def _pil_batch(n, h, w, seed, masks):
    rng = np.random.default_rng(seed)
    imgs = [Image.fromarray(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)) for _ in range(n)]
    ms = [Image.fromarray(rng.integers(1, 18, (h, w), dtype=np.uint8)) for _ in range(n)] if masks else [None] * n
    return imgs, ms


@pytest.mark.parametrize("kind,h,w,masks", [("train", 256, 256, False), ("val", 256, 256, True), ("train", 200, 333, True),
                                             ("crop", 300, 280, False), ("plain", 96, 96, True), ("seg_train", 256, 256, True),
                                             ("seg_train", 300, 180, True)])
def test_device_plan_equals_pillow_pipeline_via_oracle(kind, h, w, masks):
    """CPU: the plan's tables + the oracle's arithmetic reproduce datasets/transforms.py (Pillow) exactly, with the same
    random draws -- i.e. what mv_image_prepare is asked to compute IS the reference pipeline."""
    from myrtle_vision.datasets.device_transforms import DevicePlan
    from myrtle_vision.datasets.transforms import build_transform
    cfg = _cfg(kind)
    imgs, ms = _pil_batch(3, h, w, 7, masks)
    plan, cpu = DevicePlan(cfg), build_transform(cfg)
    for img, m in zip(imgs, ms):
        random.seed(1234)
        want_img, want_mask = cpu(img, m)
        random.seed(1234)
        s = plan(img, m)
        raw = s["raw"].numpy().astype(np.int64)
        if "kh1" in s:                                                        # Resize first, as its own uint8 image
            raw = _resample(raw, *(s[k].numpy() for k in ("kh1", "bh1", "kv1", "bv1")))
        out = _resample(raw, *(s[k].numpy() for k in ("kh", "bh", "kv", "bv"))).astype(np.uint8)
        if s["flip"]:
            out = out[:, ::-1]
        t = out.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
        t = (t - np.asarray(plan.mean, np.float32).reshape(3, 1, 1)) / np.asarray(plan.std, np.float32).reshape(3, 1, 1)
        assert np.array_equal(t, want_img.numpy())
        if masks:
            mm = s["mask"].numpy()
            if "yi1" in s:
                mm = mm[s["yi1"].numpy()][:, s["xi1"].numpy()]
            mm = mm[s["yi"].numpy()][:, s["xi"].numpy()]
            if s["flip"]:
                mm = mm[:, ::-1]
            assert np.array_equal(mm.astype(np.int64), want_mask.numpy())


@pytest.mark.gpu
@pytest.mark.parametrize("kind,h,w,masks", [("train", 256, 256, False), ("val", 256, 256, True), ("train", 200, 333, True),
                                             ("crop", 300, 280, False), ("plain", 96, 96, True), ("train", 1024, 900, False),
                                             ("seg_train", 256, 256, True), ("seg_train", 500, 380, True)])
def test_gpu_image_prepare_is_bit_exact(kind, h, w, masks):
    """mv_image_prepare / mv_mask_prepare == the Pillow pipeline, bit for bit, for a collated batch (mixed frame sizes)."""
    from myrtle_vision.datasets.device_transforms import DevicePlan
    from myrtle_vision.datasets.transforms import build_transform
    cfg = _cfg(kind)
    imgs, ms = _pil_batch(5, h, w, 11, masks)
    if kind != "plain":                                                      # a smaller frame in the same batch: padding path
        extra, extra_m = _pil_batch(1, h - 9, w - 5, 12, masks)
        imgs, ms = imgs + extra, ms + extra_m
    plan, cpu = DevicePlan(cfg), build_transform(cfg)
    random.seed(99)
    want = [cpu(i, m) for i, m in zip(imgs, ms)]
    random.seed(99)
    packed, labels = DevicePlan.collate([(plan(i, m), 3) for i, m in zip(imgs, ms)])
    assert labels.tolist() == [3] * len(imgs)
    got_img, got_mask = plan.apply(packed, torch.device("cuda"), mask_add=-1)
    torch.cuda.synchronize()
    assert got_img.dtype == torch.float32 and got_img.shape == (len(imgs), 3) + tuple(want[0][0].shape[1:])
    for i, (wi, wm) in enumerate(want):
        assert torch.equal(got_img[i].cpu(), wi), i
        if masks:
            assert torch.equal(got_mask[i].cpu(), wm - 1), i
