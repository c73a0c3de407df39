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


def test_philox_oracle_known_answers():
    """Philox4x32-10 known-answer vectors published with Random123 (kat_vectors): pins oracle/dropout_oracle.py, which in
    turn pins the dropout mask of the HIP kernel (tests/test_train_gpu.py)."""
    from oracle.dropout_oracle import dropout_keep_mask, philox4x32_10
    kats = [((0, 0, 0, 0), (0, 0), (0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8)),
            ((0xffffffff,) * 4, (0xffffffff, 0xffffffff), (0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd)),
            ((0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344), (0xa4093822, 0x299f31d0),
             (0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1))]
    for ctr, key, want in kats:
        got = philox4x32_10(*[[c] for c in ctr], *key)
        assert tuple(int(g[0]) for g in got) == want
    m = dropout_keep_mask(100000, 0.25, 42, 7)
    assert abs(m.mean() - 0.75) < 6e-3 and m.shape == (100000,)
