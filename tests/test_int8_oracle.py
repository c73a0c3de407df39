"""Pin the model-level oracle of BASELINE config 5 (oracle/int8_oracle.py) against torch's OWN observer and fake-quant
operators on the CPU: the reference installs exactly those (utils/quantize.py:237-249), its converted model cannot run
(SURVEY 9.2), so torch's operators composed at the reference's Linear sites ARE the specification."""
import numpy as np
import torch

from oracle import int8_oracle
from oracle.detinit import det_images, det_state_dict
from oracle.vit_oracle import ViTConfig, vit_forward

MICRO = dict(decoder="classification", image_size=224, patch_size=16, num_classes=45, dim=192, depth=2, heads=3, mlp_dim=768)


def _torch_composition(params, calib, img, cfg):
    from torch.ao.quantization import MinMaxObserver
    lins = set(int8_oracle.linear_names(cfg))
    obs = {}

    def record(site, t):
        kind, _, name = site.partition(":")
        if name in lins and kind in ("act", "w"):
            if site not in obs:
                obs[site] = (MinMaxObserver(reduce_range=False) if kind == "act" else
                             MinMaxObserver(qscheme=torch.per_tensor_symmetric, dtype=torch.qint8))
            obs[site](t)
        return t

    with torch.no_grad():
        for b in calib:
            vit_forward(params, b, cfg, record)
    frozen = {}
    for site, o in obs.items():
        s, z = o.calculate_qparams()
        frozen[site] = (float(s), int(z), o.quant_min, o.quant_max)

    def quant(site, t):
        if site not in frozen:
            return t
        s, z, lo, hi = frozen[site]
        return torch.fake_quantize_per_tensor_affine(t, s, z, lo, hi)

    with torch.no_grad():
        return vit_forward(params, img, cfg, quant), frozen


def test_int8_oracle_equals_torch_observers_and_fake_quant():
    cfg = ViTConfig(**MICRO)
    params = det_state_dict(cfg.param_shapes())
    calib = [det_images(f"int8-calib{i}", 2, 224) for i in range(3)]
    img = det_images("int8-eval", 4, 224)
    want, frozen = _torch_composition(params, calib, img, cfg)
    qp = int8_oracle.qparams(int8_oracle.calibrate(params, calib, cfg))
    assert set(qp) == set(frozen) and len(qp) == 2 * (4 * cfg.depth + 2)      # one act + one weight quantiser per Linear
    for site, (s, z, lo, hi) in frozen.items():
        assert (lo, hi) == qp[site][2:] and z == qp[site][1], site
        assert abs(s - float(qp[site][0])) <= 1e-9 * s, site
    got = int8_oracle.int8_forward(params, img, cfg, qp)
    np.testing.assert_array_equal(got.numpy(), want.numpy())                   # same operators, same order: same bits
    # and it IS a different function from the fp32 model (8-bit quantisers at 10 sites), by a bounded amount
    plain = vit_forward(params, img, cfg)
    d = float((got - plain).abs().max() / plain.abs().max())
    assert 1e-3 < d < 0.2, d


def test_int8_oracle_running_minmax_spans_batches():
    cfg = ViTConfig(**MICRO)
    params = det_state_dict(cfg.param_shapes())
    a, b = det_images("int8-a", 1, 224), det_images("int8-b", 1, 224) * 3.0
    ra, rb, rab = (int8_oracle.calibrate(params, x, cfg) for x in ([a], [b], [a, b]))
    for site in rab:
        assert rab[site][0] == min(ra[site][0], rb[site][0]) and rab[site][1] == max(ra[site][1], rb[site][1])
