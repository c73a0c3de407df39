"""Rounding-error budget of the bf16 forward pass, by source (CPU, emulated on the oracle).

The HIP bf16 path rounds to bf16 at a known set of places and nowhere else: the operands of every
Linear product (its input activation and its weight), the to_qkv output, the attention probabilities (P feeds the P.V
product as bf16) -- the residual stream, LayerNorm statistics, softmax, GELU and all accumulators are fp32.  Emulating
exactly those roundings on the pinned oracle, ONE SOURCE AT A TIME, attributes the end-to-end logit error of the
benchmarked arithmetic to its sources, and doing the same with IEEE half in the forward operands says what an
f16-operand forward would buy.  The GPU-measured error of the real kernels is checked
against this emulation in tests/test_vit_parity.py::test_top1_agreement_rate (same fixtures).

    python tests/test_error_budget.py [base_cls_b32|tiny_cls_b64] [n_images]     # prints the table (profiles/r03_bf16_error_budget.txt)
"""
import sys

import numpy as np
import torch

if __name__ == "__main__":
    import os
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import conftest  # noqa: F401  (sets sys.path)

from conftest import load_golden
from oracle.detinit import det_images, det_state_dict
from oracle.vit_oracle import ViTConfig, vit_forward

SOURCES = {
    "weights": lambda s: s.startswith("w:"),
    "patches (patch_to_embedding input)": lambda s: s == "act:patch_to_embedding",
    "LayerNorm outputs (to_qkv / fc1 / head inputs)": lambda s: s.startswith("act:") and s.endswith(("to_qkv", "net.0", "decoder.linear")),
    "q, k, v (to_qkv output)": lambda s: s.startswith("out:") and s.endswith("to_qkv"),
    "attention probabilities": lambda s: s == "attn:probs",
    "attention output (to_out input)": lambda s: s.startswith("act:") and s.endswith("to_out.0"),
    "gelu(h) (fc2 input)": lambda s: s.startswith("act:") and s.endswith("net.3"),
}
GEMM_OPERANDS = ["weights", "patches (patch_to_embedding input)", "LayerNorm outputs (to_qkv / fc1 / head inputs)",
                 "attention output (to_out input)", "gelu(h) (fc2 input)"]


def _round(dtype):
    return lambda t: t.to(dtype).float()


def run(name, n_images, plan):
    """plan: {source name: dtype}.  Returns logits [n_images, classes] with those roundings applied."""
    arrays, meta = load_golden(name)
    cfg = ViTConfig(patch_size=16, **meta["kwargs"])
    params = det_state_dict(cfg.param_shapes())
    img = det_images(name, meta["batch"], cfg.image_size)[:n_images]
    rules = [(SOURCES[k], _round(dt)) for k, dt in plan.items()]

    def quant(site, t):
        for match, fn in rules:
            if match(site):
                return fn(t)
        return t

    with torch.no_grad():
        return vit_forward(params, img, cfg, quant if rules else None, quant_outputs=True, probs_site=True).numpy()


def budget(name, n_images):
    want = load_golden(name)[0]["logits"][:n_images]
    scale = np.abs(want).max()
    err = lambda lg: float(np.abs(lg - want).max() / scale)          # the statistic of every parity test
    rows = {"no rounding (oracle vs reference)": err(run(name, n_images, {}))}
    for k in SOURCES:
        rows[f"bf16: {k} only"] = err(run(name, n_images, {k: torch.bfloat16}))
    rows["bf16: ALL sources (the benchmarked arithmetic, emulated)"] = err(run(name, n_images, {k: torch.bfloat16 for k in SOURCES}))
    mixed = {k: (torch.float16 if k in GEMM_OPERANDS else torch.bfloat16) for k in SOURCES}
    rows["f16 Linear operands, bf16 q/k/v and probabilities"] = err(run(name, n_images, mixed))
    rows["f16: ALL sources"] = err(run(name, n_images, {k: torch.float16 for k in SOURCES}))
    return rows


def test_budget_sources_add_up_in_quadrature():
    """ViT-Tiny, 8 images: the per-source errors are independent roundings, so the all-sources error is of the size of
    their root-sum-square (within 2x either way: max-norm statistics), bf16 as a whole lands in the envelope the GPU tests
    pin (1.5e-2) and above the 1e-3 bar, and f16 everywhere is about 8x (3 significand bits) below bf16."""
    torch.set_num_threads(8)
    rows = budget("tiny_cls_b64", 8)
    assert rows["no rounding (oracle vs reference)"] < 2e-5
    single = [v for k, v in rows.items() if k.endswith(" only")]
    rss = float(np.sqrt(np.sum(np.square(single))))
    total = rows["bf16: ALL sources (the benchmarked arithmetic, emulated)"]
    assert 0.5 * rss < total < 2.0 * rss
    assert 1e-3 < total < 1.5e-2
    assert rows["f16: ALL sources"] < total / 4
    assert rows["f16: ALL sources"] <= rows["f16 Linear operands, bf16 q/k/v and probabilities"] <= total * 1.2


if __name__ == "__main__":
    name = sys.argv[1] if len(sys.argv) > 1 else "base_cls_b32"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else load_golden(name)[1]["batch"]
    torch.set_num_threads(8)
    print(f"# rounding-error budget, {name}, {n} images: max |logit error| / max |reference logit| (CPU emulation on the oracle)")
    for k, v in budget(name, n).items():
        print(f"{v:.3e}  {k}")
