"""Known-answer and cross checks for the restated quantisers (CPU)."""
import numpy as np
import torch

from oracle import quant_oracle as q


def test_float_5_10_equals_ieee_half_except_ties_and_overflow():
    g = torch.Generator().manual_seed(7)
    x = (torch.randn(200000, generator=g) * torch.exp(torch.randn(200000, generator=g) * 4)).numpy()
    r = q.float_quantize(x, 5, 10)
    h = torch.from_numpy(x).half().float().numpy()
    fin = np.isfinite(h)
    diff = r[fin] != h[fin]
    # differences only on exact ties (round-half-away vs RNE): one fp16 ulp apart
    assert diff.mean() < 1e-3
    assert np.all(np.abs(r[fin][diff] - h[fin][diff]) <= np.abs(h[fin][diff]) * 2.0 ** -10 + 2.0 ** -24)
    assert np.all(np.abs(r[~fin]) == 65504.0)          # saturate, never inf


def test_float_known_answers():
    x = np.array([0.0, 1.0, 1.0 + 2.0 ** -11, 1.0 + 2.0 ** -11 + 2.0 ** -20, -1.0 - 2.0 ** -11, 65504.0, 65520.0,
                  1e9, 2.0 ** -14, 2.0 ** -24, 2.0 ** -25, 2.0 ** -26], dtype=np.float32)
    want = np.array([0.0, 1.0, 1.0 + 2.0 ** -10, 1.0 + 2.0 ** -10, -1.0 - 2.0 ** -10, 65504.0, 65504.0,
                     65504.0, 2.0 ** -14, 2.0 ** -24, 2.0 ** -24, 0.0], dtype=np.float32)
    np.testing.assert_array_equal(q.float_quantize(x, 5, 10), want)


def test_tf32_is_13bit_mantissa_truncation_with_half_away():
    g = torch.Generator().manual_seed(3)
    x = torch.randn(50000, generator=g).numpy()
    b = x.view(np.uint32)
    want = ((b + np.uint32(1 << 12)) & ~np.uint32((1 << 13) - 1)).view(np.float32)
    np.testing.assert_array_equal(q.float_quantize(x, 8, 10), want)


def test_fixed_point_11_9():
    x = np.array([0.0, 0.001, 0.00098, -0.001, 1.9990234375, 5.0, -5.0, -2.0, 0.5 / 512], dtype=np.float32)
    r = q.fixed_point_quantize(x, 11, 9)
    want = np.array([0.0, 1 / 512, 1 / 512, -1 / 512, 2 - 1 / 512, 2 - 1 / 512, -2.0, -2.0, 1 / 512], dtype=np.float32)
    np.testing.assert_array_equal(r, want)


def test_affine_matches_torch_fake_quantize():
    g = torch.Generator().manual_seed(5)
    x = torch.randn(4096, generator=g) * 3
    obs = torch.ao.quantization.MinMaxObserver(reduce_range=False)
    obs(x)
    s, z = obs.calculate_qparams()
    s2, z2 = q.affine_qparams(x.min().item(), x.max().item(), 0, 255, symmetric=False)
    assert abs(float(s) - float(s2)) < 1e-9 and int(z) == z2
    want = torch.fake_quantize_per_tensor_affine(x, float(s), int(z), 0, 255).numpy()
    np.testing.assert_array_equal(q.fake_quant_affine(x.numpy(), s2, z2, 0, 255), want)
    obs = torch.ao.quantization.MinMaxObserver(qscheme=torch.per_tensor_symmetric, dtype=torch.qint8)
    obs(x)
    s, z = obs.calculate_qparams()
    s2, z2 = q.affine_qparams(x.min().item(), x.max().item(), -128, 127, symmetric=True)
    assert abs(float(s) - float(s2)) < 1e-9 and int(z) == z2 == 0
