"""Bit-level numpy restatement of the fake-quantisers the reference reaches.

TEST INFRASTRUCTURE (see ``oracle/__init__.py``).  PARITY UNPINNED for the
rounding itself: the arithmetic lives in the third-party package
``qtorch==0.3.0`` (reference ``setup.py:10``), which is absent from the
container, and the reference holds no vectors for it.  This file restates the
published qtorch algorithm (``quant_cpu.cpp``: ``float_quantize_nearest``,
``round_bitwise_nearest``, ``clip_exponent``, ``fixed_point_quantize_nearest``)
and is anchored on the reference's call sites:

* ``utils/quantize.py:46-50``  FloatingPoint(exp=5, man=10), nearest  (FP16)
* ``utils/quantize.py:53-57``  FloatingPoint(exp=8, man=10), nearest  (TF32)
* ``utils/quantize.py:58-72``  FixedPoint(wl=11, fl=9|8|7), nearest
* ``utils/quantize.py:84``     ``quant(X.data.float()).to(dtype)`` (fp32 in/out)

Cross-checks available without qtorch (used in tests): for (5,10) the result
equals ``x.half().float()`` except on exact ties (round-half-away here, RNE in
IEEE) and on overflow (saturate to 65504 here, inf in IEEE).
"""
import numpy as np


def float_quantize(x, exp: int, man: int):
    """Round fp32 ``x`` to a float with ``exp`` exponent / ``man`` mantissa bits.

    Nearest rounding on the fp32 bit pattern: add half an ulp of the target
    mantissa to the *magnitude bits* and truncate (ties away from zero);
    exponent saturates to the largest normal; values below the smallest normal
    are rounded on the target's subnormal grid (add/subtract 2^min_exp trick).
    """
    a = np.ascontiguousarray(x, dtype=np.float32)
    bits = a.view(np.uint32)
    sign = bits & np.uint32(0x80000000)
    mask = np.uint32((1 << (23 - man)) - 1)
    half = np.uint32(1 << (23 - man - 1))

    def round_bits(b):
        return (b + half) & ~mask

    texp = ((bits & np.uint32(0x7FFFFFFF)) >> np.uint32(23)).astype(np.int32) - 127
    min_exp = -((1 << (exp - 1)) - 2)
    max_store = (1 << (exp - 1)) - 1 + 127
    min_store = min_exp + 127
    sub = texp < min_exp

    # normal path
    q = round_bits(bits)
    qexp = ((q & np.uint32(0x7FFFFFFF)) >> np.uint32(23)).astype(np.int64)
    max_man = np.uint32(((0x7FFFFF >> (23 - man)) << (23 - man)))
    max_num = np.uint32((max_store << 23)) | max_man
    over = qexp > max_store
    q = np.where(over, sign | max_num, q)
    under = (qexp < min_store) & (q != 0)
    min_num = np.uint32(min_store << 23)
    mid_num = np.uint32((min_store - 1) << 23)
    mag = q & np.uint32(0x7FFFFFFF)
    q = np.where(under, np.where(mag > mid_num, sign | min_num, np.uint32(0)), q)
    normal = q.astype(np.uint32).view(np.float32)

    # subnormal path: val = a + sign*2^min_exp, round mantissa, subtract again
    shift_bits = (np.uint32((127 + min_exp) << 23) | sign).astype(np.uint32)
    shift = shift_bits.view(np.float32)
    with np.errstate(over="ignore", invalid="ignore"):
        val = (a + shift).astype(np.float32)
        qs = round_bits(val.view(np.uint32)).astype(np.uint32).view(np.float32) - shift
    out = np.where(sub, qs.astype(np.float32), normal)
    return out.astype(np.float32).reshape(np.shape(x))


def fixed_point_quantize(x, wl: int, fl: int, clamp: bool = True, symmetric: bool = False):
    """Nearest fixed-point: floor(x*2^fl + 0.5)*2^-fl, clamped to the signed
    ``wl``-bit range [-2^(wl-fl-1), 2^(wl-fl-1) - 2^-fl]."""
    a = np.asarray(x, dtype=np.float32)
    scale = np.float32(2.0 ** fl)
    r = np.floor(a * scale + np.float32(0.5)) / scale
    if clamp:
        t_max = np.float32(2.0 ** (wl - fl - 1) - 2.0 ** (-fl))
        t_min = np.float32(-(2.0 ** (wl - fl - 1)))
        if symmetric:
            t_min = np.float32(t_min + 2.0 ** (-fl))
        r = np.clip(r, t_min, t_max)
    return r.astype(np.float32)


def affine_qparams(min_val: float, max_val: float, qmin: int, qmax: int, symmetric: bool):
    """``MinMaxObserver.calculate_qparams`` (torch.ao) restated: the observer the
    reference installs at ``utils/quantize.py:242-249``."""
    # torch evaluates this in fp32 tensor arithmetic; do the same so scales agree bit for bit
    f32 = np.float32
    eps = f32(np.finfo(np.float32).eps)
    min_neg = min(f32(min_val), f32(0.0))
    max_pos = max(f32(max_val), f32(0.0))
    if symmetric:
        m = max(-min_neg, max_pos)
        scale = max(f32(m / f32(float(qmax - qmin) / 2)), eps)
        zp = 0 if qmin < 0 else 128
    else:
        scale = max(f32((max_pos - min_neg) / f32(qmax - qmin)), eps)
        zp = qmin - int(np.rint(f32(min_neg / scale)))
        zp = int(min(max(zp, qmin), qmax))
    return np.float32(scale), int(zp)


def fake_quant_affine(x, scale, zero_point: int, qmin: int, qmax: int):
    """``torch.fake_quantize_per_tensor_affine`` restated:
    (clamp(nearbyint(x/scale) + zp, qmin, qmax) - zp) * scale, round-half-even."""
    a = np.asarray(x, dtype=np.float32)
    inv = np.float32(1.0) / np.float32(scale)
    q = np.clip(np.rint(a * inv) + zero_point, qmin, qmax)
    return ((q - zero_point) * np.float32(scale)).astype(np.float32)
