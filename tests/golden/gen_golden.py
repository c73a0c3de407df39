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
from oracle.detinit import det_images, det_labels, det_param, summarize, timm_source_shapes  # noqa: E402

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
    # round 4: BASELINE config 4 at its own shape -- ViT-B width, 256^2 inputs (257 tokens, bicubic pos-emb resize, seg tail)
    "base_seg_256": (dict(decoder="segmentation", image_size=256, num_classes=17, **BASE), 1, None, False),
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


# ---------------------------------------------------------------- round 4: the reference's timm renaming (row f4)
TIMM_CFG = dict(embed_dim=64, depth=2, heads=1, mlp_dim=128, patch_size=16, image_size=224, num_classes=10)


def gen_timm_rename(here):
    """Run the REFERENCE's ``rename_timm_state_dict`` (utils/models.py:154-223, rule table :157-188) on a deterministic
    timm-keyed state dict.  ``timm.create_model`` needs the network; an in-memory ``timm`` module whose ``create_model``
    returns that state dict stands in for it (as does a one-name ``torchvision.models``: utils/models.py:6 imports
    ``resnet50`` for the distillation teacher only).  The renaming itself -- rule table, head filter, conv->linear
    permutation -- is the reference's code, unmodified."""
    calls = []

    class _TimmViT:
        def __init__(self, sd):
            self._sd = sd

        def state_dict(self):
            return self._sd

    def create_model(name, pretrained=False, num_classes=None, **kw):
        calls.append([name, bool(pretrained), num_classes])
        return _TimmViT({k: det_param("timm:" + k, s) for k, s in timm_source_shapes(TIMM_CFG).items()})

    t = types.ModuleType("timm")
    t.create_model = create_model
    sys.modules["timm"] = t
    tv, tvm = types.ModuleType("torchvision"), types.ModuleType("torchvision.models")
    tvm.resnet50 = lambda *a, **k: (_ for _ in ()).throw(RuntimeError("distillation teacher is out of scope"))
    tv.models = tvm
    sys.modules.setdefault("torchvision", tv)
    sys.modules.setdefault("torchvision.models", tvm)
    from myrtle_vision.utils import models as ref_models
    assert "/root/reference/" in ref_models.__file__

    vit_config = {"embed_dim": TIMM_CFG["embed_dim"], "patch_size": TIMM_CFG["patch_size"]}
    out = ref_models.rename_timm_state_dict("vit_nano_patch16_224", vit_config, TIMM_CFG["num_classes"])
    src = timm_source_shapes(TIMM_CFG)
    meta = {"cfg": TIMM_CFG, "create_model_calls": calls, "timm_keys": list(src.keys()),
            "renamed_keys": list(out.keys()),                       # ORDER of the returned dict
            "shapes": {k: list(v.shape) for k, v in out.items()},
            "torch": torch.__version__}
    # which timm key each output came from: values are unique per key, so match on content
    origin = {}
    for nk, v in out.items():
        for ok, s in src.items():
            w = det_param("timm:" + ok, s)
            if w.numel() == v.numel() and torch.equal(w.flatten().sort().values, v.flatten().sort().values):
                origin[nk] = ok
        assert nk in origin, nk
    meta["origin"] = origin
    meta["dropped"] = [k for k in src if k not in origin.values()]
    # the reference loads the result with strict=False and asserts no unexpected keys (segmentation/train.py:164-175)
    vit = ViT(patch_size=16, q_format="FP32", decoder="classification", image_size=TIMM_CFG["image_size"],
              num_classes=TIMM_CFG["num_classes"], dim=TIMM_CFG["embed_dim"], depth=TIMM_CFG["depth"],
              heads=TIMM_CFG["heads"], mlp_dim=TIMM_CFG["mlp_dim"])
    res = vit.load_state_dict(out, strict=False)
    assert res.unexpected_keys == []
    meta["missing_keys_after_load"] = list(res.missing_keys)
    arrays = {"patch_to_embedding.weight": out["patch_to_embedding.weight"].contiguous().numpy()}
    for k, v in out.items():
        arrays["sum:" + k] = summarize(v).numpy()
    np.savez_compressed(os.path.join(here, "timm_rename.npz"), **arrays)
    with open(os.path.join(here, "timm_rename.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print(f"timm_rename: {len(out)} renamed keys, dropped {meta['dropped']}, missing after load {meta['missing_keys_after_load']}")


def main():
    only = set(sys.argv[1:])
    here = os.path.dirname(os.path.abspath(__file__))
    if not only or "timm_rename" in only:
        gen_timm_rename(here)
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
