"""CPU oracle of BASELINE config 5: the converted PyTorchINT8 ViT as a whole model (TEST INFRASTRUCTURE).

The reference's own converted int8 model does not run (``quantized::cat`` on a float positional embedding,
SURVEY.md 9.2), so its *semantics* are what its prepare step installs (``utils/quantize.py:230-251``) and what its
calibration loop does (``classification/test_quantize.py:26-34,109``), composed here on the pinned fp32 oracle:

* **prepare**: a ``MinMaxObserver`` (quint8, per-tensor affine, ``reduce_range=False``) on the input of every
  ``nn.Linear`` and a ``MinMaxObserver`` (qint8, per-tensor symmetric) on every Linear weight;
* **calibrate**: forward passes of the still-fp32 model under ``no_grad`` update running min / max;
* **convert**: ``calculate_qparams`` freezes (scale, zero_point); afterwards every Linear computes
  ``linear(fake_quantize(x), fake_quantize(W)) + b`` -- an 8-bit integer product with the scales factored out --
  and everything between the Linears (LayerNorm, softmax attention, GELU, residual adds) stays fp32, with
  ``nn.GELU`` evaluated in front of fc2's input quantiser (the reference's QGELU: dequant -> gelu -> quant,
  ``utils/quantize.py:169-184``).

The arithmetic of the two building blocks is ``oracle/quant_oracle.py`` (``affine_qparams`` = torch's
``MinMaxObserver.calculate_qparams``, ``fake_quant_affine`` = ``torch.fake_quantize_per_tensor_affine``), pinned
bit for bit against torch on the CPU in ``tests/test_quant_oracle.py``; the composition is pinned against the same
model driven by torch's OWN observer and fake-quant operators in ``tests/test_int8_oracle.py``.  The placement of
the quantisers is the reference's QAT site list (``vit_oracle.vit_forward``: "act:<linear>" / "w:<linear>"),
itself pinned by the reference-generated site fixtures.  Parity with a running reference: **unpinned** (there is none).

Never imported by the product package.
"""
from typing import Dict, Iterable, Tuple

import numpy as np
import torch

from . import quant_oracle
from .vit_oracle import ViTConfig, vit_forward


def linear_names(cfg: ViTConfig):
    """State-dict prefixes of every nn.Linear of the model, in forward order."""
    names = ["patch_to_embedding"]
    for i in range(cfg.depth):
        p = f"transformer.layers.{i}"
        names += [f"{p}.0.fn.fn.to_qkv", f"{p}.0.fn.fn.to_out.0", f"{p}.1.fn.fn.net.0", f"{p}.1.fn.fn.net.3"]
    return names + ["decoder.linear"]


def calibrate(params: Dict[str, torch.Tensor], batches: Iterable[torch.Tensor], cfg: ViTConfig) -> Dict[str, Tuple[float, float]]:
    """Running (min, max) of every Linear's input over the calibration batches (``test_quantize.py:26-34``; the
    observers only record in prepared mode, ``utils/quantize.py:242-249``) and of every Linear weight."""
    lins = set(linear_names(cfg))
    rng = {}

    def record(site, t):
        kind, _, name = site.partition(":")
        if kind == "act" and name in lins:
            lo, hi = float(t.min()), float(t.max())
            old = rng.get(site)
            rng[site] = (lo, hi) if old is None else (min(old[0], lo), max(old[1], hi))
        return t

    with torch.no_grad():
        for img in batches:
            vit_forward(params, img, cfg, record)
    for name in lins:
        w = params[f"{name}.weight"]
        rng[f"w:{name}"] = (float(w.min()), float(w.max()))
    return rng


def qparams(ranges: Dict[str, Tuple[float, float]]) -> Dict[str, Tuple[np.float32, int, int, int]]:
    """site -> (scale, zero_point, qmin, qmax): activations quint8 affine, weights qint8 symmetric."""
    out = {}
    for site, (lo, hi) in ranges.items():
        if site.startswith("w:"):
            s, z = quant_oracle.affine_qparams(lo, hi, -128, 127, symmetric=True)
            out[site] = (s, z, -128, 127)
        else:
            s, z = quant_oracle.affine_qparams(lo, hi, 0, 255, symmetric=False)
            out[site] = (s, z, 0, 255)
    return out


def int8_forward(params: Dict[str, torch.Tensor], img: torch.Tensor, cfg: ViTConfig, qp) -> torch.Tensor:
    """The converted model: fake-quantised operands at every Linear, fp32 everywhere else."""
    def quant(site, t):
        if site not in qp:
            return t                                               # LayerNorm inputs: no quantiser in this format
        s, z, lo, hi = qp[site]
        return torch.from_numpy(quant_oracle.fake_quant_affine(t.detach().numpy(), s, z, lo, hi))

    with torch.no_grad():
        return vit_forward(params, img, cfg, quant)
