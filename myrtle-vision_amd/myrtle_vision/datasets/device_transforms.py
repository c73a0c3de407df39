"""Image preparation split between the DataLoader worker and the GPU.

The reference's workers run the whole torchvision/Pillow chain per image and ship fp32 tensors
(datasets/resisc45.py:40-69, datasets/dlrsd.py:39-66, ``num_workers=1, pin_memory=False``: classification/train.py:117-125).
Here the worker only DECODES the file and draws the random parameters; it hands over

  * the decoded frame as uint8 HWC (3x fewer bytes over PCIe than the fp32 CHW tensor), and
  * per axis, Pillow's resampling tables for that image's crop box (a few KB: ``bilinear_tables``),

and one HIP kernel (``mv_image_prepare``; ``mv_mask_prepare`` for segmentation masks) does crop + antialiased bilinear
resize + flip + ToTensor + Normalize for the whole batch, bit-exact to the Pillow path in ``datasets/transforms.py``
(tests/test_image_prep.py).  Supported chains (all the reference's configs): [Resize] -> [RandomResizedCrop] ->
[CenterCrop] -> [RandomHorizontalFlip] -> ToTensor -> [Normalize].  Resize FOLLOWED BY RandomResizedCrop (the
segmentation training config) is two resamplings, each rounded to uint8 by Pillow: the GPU runs them as two kernels
(``mv_image_resize_u8`` / ``mv_mask_resize_u8`` keep the intermediate uint8 image), with a second set of tables.
"""
import math
import random

import numpy as np
import torch

from myrtle_vision.datasets.transforms import RandomResizedCrop

PRECISION_BITS = 22          # Pillow Resample.c: 32 - 8 - 2
MAX_TAPS = 64                # mv_image_prepare's limit on taps per output pixel (downscale factor < ~31)


def bilinear_tables(in_size, out_size):
    """Pillow's precompute_coeffs + normalize_coeffs_8bpc for BILINEAR over a whole axis of ``in_size`` pixels, vectorised
    over the output pixels with the C code's operation order (IEEE double: identical results).
    -> (bounds int32 [out, 2] = (first source index, tap count), kk int32 [out, ksize])"""
    scale = filterscale = float(in_size) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    ss = 1.0 / filterscale
    xx = np.arange(out_size, dtype=np.float64)
    center = 0.0 + (xx + 0.5) * scale
    xmin = np.trunc(center - support + 0.5).astype(np.int64)
    xmin = np.maximum(xmin, 0)
    xmax = np.trunc(center + support + 0.5).astype(np.int64)
    xmax = np.minimum(xmax, in_size) - xmin
    k = np.arange(ksize, dtype=np.int64)[None, :]
    valid = k < xmax[:, None]
    v = ((k + xmin[:, None]).astype(np.float64) - center[:, None] + 0.5) * ss
    v = np.abs(v)
    w = np.where(valid & (v < 1.0), 1.0 - v, 0.0)
    ww = np.zeros(out_size, np.float64)
    for j in range(ksize):                                   # same left-to-right accumulation as the C loop
        ww = ww + w[:, j]
    w = np.where((ww != 0.0)[:, None], w / np.where(ww == 0.0, 1.0, ww)[:, None], w)
    kk = np.trunc(0.5 + w * float(1 << PRECISION_BITS)).astype(np.int32)
    kk[~valid] = 0
    return np.stack([xmin, xmax], axis=1).astype(np.int32), kk


def nearest_table(in_size, out_size):
    """Source index per output pixel of Pillow's NEAREST resize of a whole axis (Geometry.c ImagingScaleAffine:
    ``xo = a0 / 2`` advanced by ``xo += a0`` -- the running sum, not a product), clamped into range."""
    a0 = float(in_size) / out_size
    xo = np.empty(out_size, np.float64)
    acc = 0.0 + a0 * 0.5
    for x in range(out_size):
        xo[x] = acc
        acc += a0
    return np.clip(np.trunc(xo).astype(np.int64), 0, in_size - 1).astype(np.int32)


class DevicePlan:
    """Built from a reference ``transform_ops_*`` section.  ``__call__(pil_image, pil_mask)`` -> one sample dict of
    small CPU tensors (decoded frame + tables); ``collate`` stacks them; ``apply`` runs the kernels on the GPU."""

    def __init__(self, transform_config):
        unknown = set(transform_config) - {"Resize", "RandomResizedCrop", "CenterCrop", "RandomHorizontalFlip", "Normalize"}
        if unknown:
            raise ValueError(f"transform chain not supported on the device path: {sorted(transform_config)}")
        self.resize = transform_config.get("Resize")
        self.rrc = RandomResizedCrop(transform_config["RandomResizedCrop"]) if "RandomResizedCrop" in transform_config else None
        self.center = transform_config.get("CenterCrop")
        self.flip = "RandomHorizontalFlip" in transform_config
        n = transform_config.get("Normalize")
        self.mean = tuple(float(v) for v in n["Mean"]) if n else (0.0, 0.0, 0.0)
        self.std = tuple(float(v) for v in n["Std"]) if n else (1.0, 1.0, 1.0)
        self._cache = {}

    # ---- worker side ------------------------------------------------------------------------------------------
    def _tables(self, n_in, n_out):
        t = self._cache.get((n_in, n_out))
        if t is None:
            if len(self._cache) > 4096:
                self._cache.clear()
            t = self._cache[(n_in, n_out)] = bilinear_tables(n_in, n_out) + (nearest_table(n_in, n_out),)
        return t

    def __call__(self, img, mask=None):
        a = np.asarray(img.convert("RGB"), dtype=np.uint8)
        H, W = a.shape[:2]
        first = None
        if self.rrc is not None and self.resize is not None:
            # stage one: Resize((S, S)) of the whole frame; the crop below then lives on that S x S uint8 image
            S = int(self.resize)
            b1v, k1v, y1 = self._tables(H, S)
            b1h, k1h, x1 = self._tables(W, S)
            first = (b1v, k1v, y1, b1h, k1h, x1)
            H = W = S
        # same draws, in the same order, as transforms.RandomResizedCrop / RandomHorizontalFlip
        if self.rrc is not None:
            top, left, ch, cw = self.rrc._params(W, H)
            oh = ow = self.rrc.size[0]
        else:
            top, left, ch, cw = 0, 0, H, W
            oh = ow = self.resize if self.resize is not None else None
        if oh is None:
            oh, ow = ch, cw
        bv, kv, yi = self._tables(ch, oh)
        bh, kh, xi = self._tables(cw, ow)
        bv, bh, yi, xi = bv.copy(), bh.copy(), yi + top, xi + left
        bv[:, 0] += top
        bh[:, 0] += left
        if self.center is not None:                          # CenterCrop of the resized image = a window of the tables
            c = int(self.center)
            t0, l0 = (oh - c) // 2, (ow - c) // 2
            bv, kv, yi, bh, kh, xi = bv[t0:t0 + c], kv[t0:t0 + c], yi[t0:t0 + c], bh[l0:l0 + c], kh[l0:l0 + c], xi[l0:l0 + c]
        flip = 1 if (self.flip and random.random() < 0.5) else 0
        s = {"raw": torch.from_numpy(a.copy()), "kh": torch.from_numpy(np.ascontiguousarray(kh)),
             "bh": torch.from_numpy(np.ascontiguousarray(bh)), "kv": torch.from_numpy(np.ascontiguousarray(kv)),
             "bv": torch.from_numpy(np.ascontiguousarray(bv)), "flip": flip}
        if first is not None:
            s.update(kv1=torch.from_numpy(np.ascontiguousarray(first[1])), bv1=torch.from_numpy(np.ascontiguousarray(first[0])),
                     kh1=torch.from_numpy(np.ascontiguousarray(first[4])), bh1=torch.from_numpy(np.ascontiguousarray(first[3])))
        if mask is not None:
            m = np.asarray(mask, dtype=np.uint8)
            if m.shape != a.shape[:2]:
                raise ValueError(f"mask {m.shape} does not match image {a.shape[:2]}")
            if first is not None:
                s.update(yi1=torch.from_numpy(np.ascontiguousarray(first[2])), xi1=torch.from_numpy(np.ascontiguousarray(first[5])))
            s.update(mask=torch.from_numpy(m.copy()), yi=torch.from_numpy(np.ascontiguousarray(yi)),
                     xi=torch.from_numpy(np.ascontiguousarray(xi)))
        return s

    # ---- collate (worker) -------------------------------------------------------------------------------------
    @staticmethod
    def collate(batch):
        """batch: list of (sample dict, label | None).  Frames of different sizes are padded to the largest (the tables
        only address valid pixels); tap counts are padded to the batch maximum."""
        samples = [b[0] for b in batch]
        B = len(samples)
        Hs, Ws = max(s["raw"].shape[0] for s in samples), max(s["raw"].shape[1] for s in samples)
        ks = max(max(s["kh"].shape[1], s["kv"].shape[1]) for s in samples)
        if ks > MAX_TAPS:
            raise ValueError(f"downscale factor too large for the device path ({ks} taps)")
        oh, ow = samples[0]["kv"].shape[0], samples[0]["kh"].shape[0]
        labels = None if batch[0][1] is None else torch.as_tensor([b[1] for b in batch])
        has_mask = "mask" in samples[0]
        # frames (and masks): one stack when they share a shape -- the common case and the expensive copies (196 KB each)
        # --, zero-padded to the largest otherwise
        if all(s["raw"].shape == samples[0]["raw"].shape for s in samples):
            raw = torch.stack([s["raw"] for s in samples])
            mask = torch.stack([s["mask"] for s in samples]) if has_mask else None
        else:
            raw = torch.zeros(B, Hs, Ws, 3, dtype=torch.uint8)
            mask = torch.zeros(B, Hs, Ws, dtype=torch.uint8) if has_mask else None
            for i, s in enumerate(samples):
                h, w = s["raw"].shape[:2]
                raw[i, :h, :w] = s["raw"]
                if has_mask:
                    mask[i, :h, :w] = s["mask"]
        # tables: a few KB per sample; tap counts differ with the crop size (3 when upscaling, 5+ when downscaling)
        kh, kv = torch.zeros(B, ow, ks, dtype=torch.int32), torch.zeros(B, oh, ks, dtype=torch.int32)
        for i, s in enumerate(samples):
            if s["kv"].shape[0] != oh or s["kh"].shape[0] != ow:
                raise ValueError("samples of one batch must share the output size")
            kh[i, :, :s["kh"].shape[1]] = s["kh"]
            kv[i, :, :s["kv"].shape[1]] = s["kv"]
        out = {"raw": raw, "kh": kh, "bh": torch.stack([s["bh"] for s in samples]), "kv": kv,
               "bv": torch.stack([s["bv"] for s in samples]),
               "flip": torch.tensor([s["flip"] for s in samples], dtype=torch.uint8)}
        if has_mask:
            out.update(mask=mask, yi=torch.stack([s["yi"] for s in samples]), xi=torch.stack([s["xi"] for s in samples]))
        if "kh1" in samples[0]:                                              # Resize -> RandomResizedCrop: stage-one tables
            k1 = max(max(s["kh1"].shape[1], s["kv1"].shape[1]) for s in samples)
            if k1 > MAX_TAPS:
                raise ValueError(f"downscale factor too large for the device path ({k1} taps)")
            S = samples[0]["kv1"].shape[0]
            kh1, kv1 = torch.zeros(B, S, k1, dtype=torch.int32), torch.zeros(B, S, k1, dtype=torch.int32)
            for i, s in enumerate(samples):
                kh1[i, :, :s["kh1"].shape[1]] = s["kh1"]
                kv1[i, :, :s["kv1"].shape[1]] = s["kv1"]
            out.update(kh1=kh1, kv1=kv1, bh1=torch.stack([s["bh1"] for s in samples]), bv1=torch.stack([s["bv1"] for s in samples]))
            if has_mask:
                out.update(yi1=torch.stack([s["yi1"] for s in samples]), xi1=torch.stack([s["xi1"] for s in samples]))
        return out, labels

    # ---- GPU side ---------------------------------------------------------------------------------------------
    def apply(self, packed, device, mask_add=0):
        """Packed CPU batch (pinned by the DataLoader) -> (images fp32 [B, 3, h, w], masks int64 [B, h, w] | None) on
        ``device``; the copies are asynchronous and the kernels run on the current stream."""
        from myrtle_vision.hip import ops
        d = {k: v.to(device, non_blocking=True) for k, v in packed.items()}
        raw, mask = d["raw"], d.get("mask")
        if "kh1" in d:                                                       # Resize first, as its own uint8 image
            raw = ops.image_resize_u8(raw, d["kh1"], d["bh1"], d["kv1"], d["bv1"])
            if mask is not None:
                mask = ops.mask_resize_u8(mask, d["yi1"], d["xi1"])
        imgs = ops.image_prepare(raw, d["kh"], d["bh"], d["kv"], d["bv"], d["flip"], self.mean, self.std)
        masks = None
        if mask is not None:
            masks = ops.mask_prepare(mask, d["yi"], d["xi"], d["flip"], mask_add)
        return imgs, masks
