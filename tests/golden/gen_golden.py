#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own ViT in this container.

Run from the repo root (build container only -- /root/reference does not exist
on the GPU box and nothing under tests/ reads it at test time):

    python tests/golden/gen_golden.py

The reference is imported from /root/reference/src.  Its only missing import
is the third-party ``qtorch`` package; a 3-name in-memory stand-in is registered
whose ``Quantizer`` calls the restated quantiser in ``oracle/quant_oracle.py``
and records every call site.  For q_format FP32 the
stand-in is never called, so those fixtures pin the reference arithmetic itself;
for FP16_32/TF32 they pin the reference's *placement* of quantisers around the
restated rounding (rounding itself: parity unpinned, see oracle/__init__.py).

Fixtures hold outputs only; parameters and inputs are formulas in
``oracle/detinit.py``.
"""
import json
import os
import sys
import types
import warnings

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/src")
warnings.filterwarnings("ignore")

from oracle import quant_oracle  # noqa: E402
from oracle.detinit import det_images, det_labels, det_param, summarize  # noqa: E402

SITES = []


def _install_qtorch_standin():
    q = types.ModuleType("qtorch")
    qq = types.ModuleType("qtorch.quant")

    class FixedPoint:
        def __init__(self, wl, fl):
            self.wl, self.fl = wl, fl

    class FloatingPoint:
        def __init__(self, exp, man):
            self.exp, self.man = exp, man

    class Quantizer(torch.nn.Module):
        def __init__(self, forward_number=None, forward_rounding="nearest", **kw):
            super().__init__()
            assert forward_rounding == "nearest"
            self.number = forward_number

        def forward(self, x):
            n = self.number
            SITES.append((type(n).__name__, tuple(x.shape)))
            a = x.detach().cpu().numpy()
            if isinstance(n, FloatingPoint):
                r = quant_oracle.float_quantize(a, n.exp, n.man)
            else:
                r = quant_oracle.fixed_point_quantize(a, n.wl, n.fl)
            return torch.from_numpy(r).to(x.device)

    q.FixedPoint, q.FloatingPoint = FixedPoint, FloatingPoint
    qq.Quantizer = Quantizer
    q.quant = qq
    sys.modules["qtorch"] = q
    sys.modules["qtorch.quant"] = qq


_install_qtorch_standin()
from myrtle_vision.models.vit import ViT  # noqa: E402  (the reference)

assert "/root/reference/" in sys.modules["myrtle_vision.models.vit"].__file__

MICRO = dict(dim=192, depth=2, heads=3, mlp_dim=768)
TINY = dict(dim=192, depth=12, heads=3, mlp_dim=768)
BASE = dict(dim=768, depth=12, heads=12, mlp_dim=3072)

CASES = {
    # name: (vit kwargs, batch, q_format applied after loading, convert?)
    "micro_cls": (dict(decoder="classification", image_size=224, num_classes=45, **MICRO), 2, None, False),
    "micro_cls_256": (dict(decoder="classification", image_size=256, num_classes=45, **MICRO), 1, None, False),
    "micro_seg": (dict(decoder="segmentation", image_size=224, num_classes=17, **MICRO), 2, None, False),
    "tiny_cls": (dict(decoder="classification", image_size=224, num_classes=45, **TINY), 8, None, False),
    "base_cls": (dict(decoder="classification", image_size=224, num_classes=1000, **BASE), 2, None, False),
    "micro_cls_fp16_32": (dict(decoder="classification", image_size=224, num_classes=45, **MICRO), 2, "FP16_32", False),
    "micro_cls_tf32": (dict(decoder="classification", image_size=224, num_classes=45, **MICRO), 2, "TF32", False),
    "micro_cls_fp16_16": (dict(decoder="classification", image_size=224, num_classes=45, **MICRO), 2, "FP16_16", False),
    "micro_cls_fp16_32_conv": (dict(decoder="classification", image_size=224, num_classes=45, **MICRO), 2, "FP16_32", True),
    # round 2: N = 257 tokens + bicubic pos-emb resize + segmentation tail in ONE model; a ViT-B-size segmentation model;
    # the fake-quantised segmentation decoder (prepare_qat wraps decoder.norm / decoder.linear too, quantize.py:289-327)
    "micro_seg_256": (dict(decoder="segmentation", image_size=256, num_classes=17, **MICRO), 1, None, False),
    "base_seg": (dict(decoder="segmentation", image_size=224, num_classes=17, **BASE), 2, None, False),
    "micro_seg_fp16_32": (dict(decoder="segmentation", image_size=224, num_classes=17, **MICRO), 2, "FP16_32", False),
    # round 3: logits-only fixtures over enough images for a top-1 agreement RATE (eval-mode forward, no backward)
    "base_cls_b32": (dict(decoder="classification", image_size=224, num_classes=1000, **BASE), 32, None, "logits_only"),
    "tiny_cls_b64": (dict(decoder="classification", image_size=224, num_classes=45, **TINY), 64, None, "logits_only"),
}


def canonical(name: str) -> str:
    """Undo the ``Sequential(QuantStub, module)`` renaming prepare_qat introduces
    (``patch_to_embedding.1.weight`` -> ``patch_to_embedding.weight``)."""
    parts = name.split(".")
    if len(parts) >= 3 and parts[-2] == "1" and parts[-1] in ("weight", "bias"):
        parts = parts[:-2] + parts[-1:]
    return ".".join(parts)


def run_case(name, kwargs, batch, q_format, convert):
    torch.manual_seed(0)
    torch.set_num_threads(8)
    vit = ViT(patch_size=16, q_format="FP32", **kwargs)
    shapes = {k: tuple(v.shape) for k, v in vit.state_dict().items()}
    vit.load_state_dict({k: det_param(k, s) for k, s in shapes.items()})
    if q_format is not None:
        vit.quantizer.prepare_qat(q_format)
    out = {}
    meta = {"kwargs": kwargs, "batch": batch, "q_format": q_format, "convert": convert,
            "param_shapes": {k: list(s) for k, s in shapes.items()},
            # ORDER of the reference's state dict (a JSON object written with sort_keys loses it; a list does not)
            "state_keys": list(shapes.keys()),
            "state_keys_prepared": list(vit.state_dict().keys()),
            "torch": torch.__version__}

    img = det_images(name, batch, kwargs["image_size"])
    if kwargs["decoder"] == "classification":
        labels = det_labels(name, (batch,), kwargs["num_classes"])
    else:
        labels = det_labels(name, (batch, kwargs["image_size"], kwargs["image_size"]), kwargs["num_classes"])

    if convert == "logits_only":
        vit.eval()
        with torch.no_grad():
            out["logits"] = vit(img).numpy()
        return out, meta

    if convert:
        vit.train()
        with torch.no_grad():
            vit(img)                       # one calibration pass (test_quantize.py:26-34)
        vit.convert()
        vit.eval()
        SITES.clear()
        with torch.no_grad():
            logits = vit(img)
        meta["sites"] = [[k, list(s)] for k, s in SITES]
        out["logits"] = logits.numpy()
        meta["state_keys_after_convert"] = list(vit.state_dict().keys())
        return out, meta

    taps = {}
    hooks = []
    for i, blk in enumerate(vit.transformer.layers):
        # Transformer.forward calls blk[0] and blk[1] directly (vit.py:156-160), so hook blk[1]
        hooks.append(blk[1].register_forward_hook(
            lambda m, a, o, i=i: taps.__setitem__(f"block{i}", o.detach())))
    # the reference's documented hook point for attention maps (vit.py:80-82,94)
    attn0 = vit.transformer.layers[0][0].fn.fn
    hooks.append(attn0.attn_output.register_forward_hook(
        lambda m, a, o: taps.__setitem__("attn0", o.detach())))

    vit.train()
    SITES.clear()
    logits = vit(img)
    meta["sites"] = [[k, list(s)] for k, s in SITES]
    loss = torch.nn.functional.cross_entropy(logits, labels)   # CrossEntropyLoss(), train.py:170,250
    loss.backward()
    for h in hooks:
        h.remove()

    if logits.dim() == 4:                      # segmentation: 2x17x224x224 is too big to commit whole
        out["logits_sub"] = logits.detach()[:, :, ::7, ::7].contiguous().numpy()
        out["logits_summary"] = summarize(logits).numpy()
        out["argmax_sub"] = logits.detach().argmax(dim=1)[:, ::7, ::7].contiguous().numpy()
    else:
        out["logits"] = logits.detach().numpy()
    out["loss"] = loss.detach().numpy()
    for k, v in taps.items():
        if k.startswith("block"):
            out[f"{k}_head"] = v[:, :3, :].contiguous().numpy()    # cls + first two patch tokens
            out[f"{k}_summary"] = summarize(v).numpy()
        else:
            out[f"{k}_head"] = v[:, :, :4, :].contiguous().numpy()
            out[f"{k}_summary"] = summarize(v).numpy()
    unused = []
    for pname, p in vit.named_parameters():
        c = canonical(pname)
        if p.grad is None:
            unused.append(c)
            continue
        out[f"gsum:{c}"] = summarize(p.grad).numpy()
        if p.numel() <= 4096 and not c.startswith("transformer.layers.1") :
            out[f"grad:{c}"] = p.grad.detach().numpy()
    meta["unused_params"] = unused
    return out, meta


def main():
    only = set(sys.argv[1:])
    here = os.path.dirname(os.path.abspath(__file__))
    for name, (kwargs, batch, qf, conv) in CASES.items():
        if only and name not in only:
            continue
        out, meta = run_case(name, kwargs, batch, qf, conv)
        np.savez_compressed(os.path.join(here, f"{name}.npz"), **out)
        with open(os.path.join(here, f"{name}.json"), "w") as f:
            json.dump(meta, f, indent=1, sort_keys=True)
        sz = os.path.getsize(os.path.join(here, f"{name}.npz"))
        print(f"{name}: {len(out)} arrays, {sz/1024:.0f} KiB, sites={len(meta.get('sites', []))}")


if __name__ == "__main__":
    main()
