"""TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): numpy restatement of the image preparation the reference's
DataLoader workers run through torchvision + Pillow (src/myrtle_vision/datasets/resisc45.py:40-69,
datasets/dlrsd.py:39-66, transforms/segmentation.py):

    RandomResizedCrop / Resize  ->  F.resized_crop = PIL crop + Image.resize(BILINEAR)   (NEAREST for masks)
    RandomHorizontalFlip        ->  PIL transpose(FLIP_LEFT_RIGHT)
    ToTensor                    ->  uint8 HWC -> float32 CHW / 255
    Normalize(mean, std)        ->  (x - mean) / std   in float32

The resampling algorithm lives in a third-party dependency of the reference (Pillow, src/libImaging/Resample.c and
Geometry.c; torchvision 0.11 / Pillow 8-9 in the reference's environment, Pillow 12.2 in this image -- the 8-bit
resampler is unchanged across them), so it is restated here from its published source and PINNED against Pillow itself:
tests/test_image_prep.py compares every function below bit for bit with PIL on random images, boxes and sizes.

Pillow's 8-bit-per-channel BILINEAR resize is a separable triangle filter whose support grows with the downscale
factor (antialiasing), evaluated in fixed point:
  * coefficients in double precision, normalised per output pixel, then rounded to 22-bit fixed point;
  * horizontal pass first, its result ROUNDED AND CLIPPED TO uint8, then the vertical pass on that.
"""
import numpy as np

PRECISION_BITS = 32 - 8 - 2          # Resample.c: PRECISION_BITS


def bilinear_coeffs(in_size, in0, in1, out_size):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the BILINEAR (triangle, support 1) filter.
    -> (bounds int32 [out, 2] = (xmin, count), kk int32 [out, ksize])"""
    scale = filterscale = (in1 - in0) / out_size
    if filterscale < 1.0:
        filterscale = 1.0
    support = 1.0 * filterscale
    ksize = int(np.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), np.int32)
    kk = np.zeros((out_size, ksize), np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = in0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = np.zeros(ksize, np.float64)
        ww = 0.0
        for x in range(xmax):
            v = (x + xmin - center + 0.5) * ss
            if v < 0.0:
                v = -v
            w[x] = 1.0 - v if v < 1.0 else 0.0
            ww += w[x]
        for x in range(xmax):
            if ww != 0.0:
                w[x] /= ww
        for x in range(ksize):
            kk[xx, x] = int(-0.5 + w[x] * (1 << PRECISION_BITS)) if w[x] < 0 else int(0.5 + w[x] * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return bounds, kk


def _clip8(v):
    return np.clip(v >> PRECISION_BITS, 0, 255).astype(np.uint8)


def resize_bilinear_u8(img, size):
    """``Image.fromarray(img).resize((size_w, size_h), BILINEAR)`` for uint8 HWC ``img``; size = (h, w)."""
    h, w = img.shape[:2]
    oh, ow = size
    src = img.astype(np.int64)
    if ow != w:                                           # horizontal pass (Resample.c: need_horizontal)
        bh, kh = bilinear_coeffs(w, 0.0, float(w), ow)
        tmp = np.empty((h, ow) + img.shape[2:], np.uint8)
        for xx in range(ow):
            x0, n = bh[xx]
            acc = np.full((h,) + img.shape[2:], 1 << (PRECISION_BITS - 1), np.int64)
            for x in range(n):
                acc += src[:, x0 + x] * int(kh[xx, x])
            tmp[:, xx] = _clip8(acc)
        src = tmp.astype(np.int64)
        img = tmp
    if oh != h:                                           # vertical pass
        bv, kv = bilinear_coeffs(h, 0.0, float(h), oh)
        out = np.empty((oh,) + img.shape[1:], np.uint8)
        for yy in range(oh):
            y0, n = bv[yy]
            acc = np.full(img.shape[1:], 1 << (PRECISION_BITS - 1), np.int64)
            for y in range(n):
                acc += src[y0 + y] * int(kv[yy, y])
            out[yy] = _clip8(acc)
        img = out
    return img


def nearest_index(in_size, out_size):
    """Source index per output pixel of ``Image.resize(NEAREST)`` on a whole image (Geometry.c ImagingScaleAffine:
    xo = a0 * 0.5 accumulated by ``xo += a0`` in double, COORD(v) = (int) v; out-of-range -> clamped by the caller)."""
    a0 = in_size / out_size
    idx = np.empty(out_size, np.int32)
    xo = 0.0 + a0 * 0.5
    for x in range(out_size):
        xin = -1 if xo < 0.0 else int(xo)
        idx[x] = xin
        xo += a0
    return idx


def resize_nearest_u8(mask, size):
    oh, ow = size
    yi, xi = nearest_index(mask.shape[0], oh), nearest_index(mask.shape[1], ow)
    return mask[np.clip(yi, 0, mask.shape[0] - 1)][:, np.clip(xi, 0, mask.shape[1] - 1)]


def prepare_image(img, box, size, flip, mean, std):
    """The full chain on one uint8 HWC image: crop(box = left, top, right, bottom) -> resize(size) -> hflip? ->
    ToTensor -> Normalize.  -> float32 [3, size_h, size_w]"""
    left, top, right, bottom = box
    a = resize_bilinear_u8(img[top:bottom, left:right], size)
    if flip:
        a = a[:, ::-1]
    t = a.astype(np.float32).transpose(2, 0, 1) / np.float32(255.0)
    if mean is not None:
        t = (t - np.asarray(mean, np.float32).reshape(-1, 1, 1)) / np.asarray(std, np.float32).reshape(-1, 1, 1)
    return np.ascontiguousarray(t, dtype=np.float32)


def prepare_mask(mask, box, size, flip):
    left, top, right, bottom = box
    m = resize_nearest_u8(mask[top:bottom, left:right], size)
    if flip:
        m = m[:, ::-1]
    return np.ascontiguousarray(m)
