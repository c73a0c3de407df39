"""Pin the CPU oracle (oracle/vit_oracle.py) against golden vectors produced by the
reference's own ViT (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import quant_oracle
from oracle.detinit import det_images, det_labels, det_state_dict, summarize
from oracle.vit_oracle import ViTConfig, loss_and_grads, vit_forward

from conftest import load_golden

FP32_CASES = ["micro_cls", "micro_cls_256", "micro_seg", "micro_seg_256", "tiny_cls", "base_cls", "base_seg", "base_seg_256"]


def _setup(name):
    arrays, meta = load_golden(name)
    cfg = ViTConfig(patch_size=16, **meta["kwargs"])
    shapes = cfg.param_shapes()
    assert {k: list(v) for k, v in shapes.items()} == meta["param_shapes"]
    assert list(shapes) == meta["state_keys"]                # the reference's state-dict ORDER (checkpoint wire format)
    params = det_state_dict(shapes)
    b = meta["batch"]
    img = det_images(name, b, cfg.image_size)
    if cfg.decoder == "classification":
        labels = det_labels(name, (b,), cfg.num_classes)
    else:
        labels = det_labels(name, (b, cfg.image_size, cfg.image_size), cfg.num_classes)
    return arrays, meta, cfg, params, img, labels


def _check_summary(got, want, rtol, name):
    # [sum, l2, absmax, weighted sum] are O(sqrt(n)) accumulations: compare against the l2 scale
    scale = max(float(want[1]), 1e-12)
    np.testing.assert_allclose(got[:4] / scale, want[:4] / scale, atol=rtol, err_msg=name)
    np.testing.assert_allclose(got[4:], want[4:], atol=rtol * max(float(want[2]), 1e-12), err_msg=name)


@pytest.mark.parametrize("name", FP32_CASES)
def test_oracle_matches_reference_fp32(name):
    arrays, meta, cfg, params, img, labels = _setup(name)
    torch.set_num_threads(8)
    logits, loss, grads = loss_and_grads(params, img, labels, cfg)
    if "logits" in arrays:
        want = arrays["logits"]
        np.testing.assert_allclose(logits.numpy(), want, atol=2e-5 * np.abs(want).max())
        assert (logits.argmax(1).numpy() == want.argmax(1)).all()
    else:
        want = arrays["logits_sub"]
        np.testing.assert_allclose(logits[:, :, ::7, ::7].numpy(), want, atol=2e-5 * np.abs(want).max())
        assert (logits.argmax(1)[:, ::7, ::7].numpy() == arrays["argmax_sub"]).all()
        _check_summary(summarize(logits).numpy(), arrays["logits_summary"], 1e-5, "logits")
    np.testing.assert_allclose(loss.numpy(), arrays["loss"], rtol=1e-5)
    # the two detection-only parameters never receive a gradient
    assert sorted(k for k, g in grads.items() if g is None) == sorted(meta["unused_params"])
    for k, g in grads.items():
        if g is None:
            continue
        _check_summary(summarize(g).numpy(), arrays[f"gsum:{k}"], 2e-4, k)
        if f"grad:{k}" in arrays:
            w = arrays[f"grad:{k}"]
            np.testing.assert_allclose(g.numpy(), w, atol=2e-4 * max(np.abs(w).max(), 1e-12), err_msg=k)


@pytest.mark.parametrize("name", ["base_cls_b32", "tiny_cls_b64"])
def test_oracle_matches_reference_logits_only(name):
    """Round-3 top-1 fixtures (eval forward of the reference over 32 / 64 images): oracle logits and EVERY argmax."""
    arrays, meta, cfg, params, img, labels = _setup(name)
    torch.set_num_threads(8)
    with torch.no_grad():
        logits = vit_forward(params, img, cfg)
    want = arrays["logits"]
    assert want.shape[0] == meta["batch"] >= 32
    np.testing.assert_allclose(logits.numpy(), want, atol=2e-5 * np.abs(want).max())
    assert (logits.argmax(1).numpy() == want.argmax(1)).all()


def test_oracle_taps_match_reference():
    arrays, meta, cfg, params, img, labels = _setup("micro_cls")
    taps = {}
    vit_forward(params, img, cfg, taps=taps)
    for i in range(cfg.depth):
        w = arrays[f"block{i}_head"]
        np.testing.assert_allclose(taps[f"block{i}"][:, :3, :].numpy(), w, atol=1e-5 * np.abs(w).max())
        _check_summary(summarize(taps[f"block{i}"]).numpy(), arrays[f"block{i}_summary"], 1e-5, f"block{i}")
    w = arrays["attn0_head"]
    np.testing.assert_allclose(taps["attn0"][:, :, :4, :].numpy(), w, atol=1e-6)


class _STE(torch.autograd.Function):
    """reference QuantizerFunction (utils/quantize.py:77-89): quantise forward, identity backward."""

    @staticmethod
    def forward(ctx, x, exp, man):
        return torch.from_numpy(quant_oracle.float_quantize(x.detach().numpy(), exp, man))

    @staticmethod
    def backward(ctx, g):
        return g, None, None


@pytest.mark.parametrize("name,exp,man,outputs", [("micro_cls_fp16_32", 5, 10, False), ("micro_cls_tf32", 8, 10, False),
                                                  ("micro_cls_fp16_16", 5, 10, True), ("micro_seg_fp16_32", 5, 10, False)])
def test_oracle_quant_sites_match_reference(name, exp, man, outputs):
    arrays, meta, cfg, params, img, labels = _setup(name)
    sites = []

    def quant(site, t):
        sites.append(list(t.shape))
        return _STE.apply(t, exp, man)

    logits, loss, grads = loss_and_grads(params, img, labels, cfg, quant, quant_outputs=outputs)
    assert sites == [s for _, s in meta["sites"]]            # same calls, same order, same shapes
    # Tolerance note: a fake-quantiser is discontinuous.  The oracle's explicit LayerNorm/GELU
    # formulas differ from ATen's kernels by a few fp32 ulps, which flips the fp16 rounding of
    # ~1e-3 of all quantised elements (each flip = one fp16 ulp = 2^-11 relative).  Measured
    # effect: ~5e-4 of max|logit|.  So quantised paths are compared at 2e-3, not 2e-5.
    if "logits" in arrays:
        want = arrays["logits"]
        np.testing.assert_allclose(logits.numpy(), want, atol=2e-3 * np.abs(want).max())
        assert (logits.argmax(1).numpy() == want.argmax(1)).all()
    else:
        want = arrays["logits_sub"]
        np.testing.assert_allclose(logits[:, :, ::7, ::7].numpy(), want, atol=2e-3 * np.abs(want).max())
    np.testing.assert_allclose(loss.numpy(), arrays["loss"], rtol=2e-3)
    for k, g in grads.items():
        if g is not None:
            _check_summary(summarize(g).numpy(), arrays[f"gsum:{k}"], 1e-2, k)
