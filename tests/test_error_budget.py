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


# ---------------------------------------------------------------------------------------------------------------------------
# Study for a faster tolerance-meeting mode (DESIGN section 8, 00 iii): bf16x3 computes a b = a0 b0 + a0 b1 + a1 b0 as three bf16
# products.  The two correction products are 2^-8 of the result, so they need only ~8 bits of relative accuracy themselves -- would
# FP8 operands (e4m3, the format of gfx950's double-rate matrix path) do?  Emulated here on the oracle's nn.Linear products:
#     y = a0 b0  (bf16 pieces, exact products, fp32 sums)  +  Q(a0) Q(b1)  +  Q(a1) Q(b0),     Q = e4m3 with one power-of-two scale
# per operand ROW (token / output feature; the hardware's block scales are finer: 32 elements).  Attention, LayerNorm, GELU and the
# residual stream stay fp32, as in precision="bf16x3".
# ---------------------------------------------------------------------------------------------------------------------------
def _pieces(t):
    p0 = t.to(torch.bfloat16).float()
    return p0, (t - p0).to(torch.bfloat16).float()


def _e4m3_rows(t):
    """Round to e4m3 (3 significand bits, finite max 448) after scaling every row by a power of two so that its largest magnitude
    lands in [128, 256); returns the dequantised values."""
    amax = t.abs().amax(dim=-1, keepdim=True).clamp_min(1e-30)
    scale = torch.exp2(7.0 - torch.floor(torch.log2(amax)))
    return (t * scale).to(torch.float8_e4m3fn).float() / scale


def _linear_x3(correction):
    def fn(name, x, w, bias):
        a0, a1 = _pieces(x)
        b0, b1 = _pieces(w)
        if correction == "bf16":
            y = a0 @ b0.t() + a0 @ b1.t() + a1 @ b0.t()
        else:
            y = a0 @ b0.t() + _e4m3_rows(a0) @ _e4m3_rows(b1).t() + _e4m3_rows(a1) @ _e4m3_rows(b0).t()
        return y + bias
    return fn


def fp8_correction_study(name, n_images):
    arrays, meta = load_golden(name)
    cfg = ViTConfig(patch_size=16, **meta["kwargs"])
    params = det_state_dict(cfg.param_shapes())
    img = det_images(name, meta["batch"], cfg.image_size)[:n_images]
    want = arrays["logits"][:n_images]
    scale = np.abs(want).max()
    rows = {}
    with torch.no_grad():
        for label, corr in [("bf16x3 (three bf16 products)", "bf16"), ("bf16 main product + two e4m3 correction products", "e4m3")]:
            lg = vit_forward(params, img, cfg, linear_fn=_linear_x3(corr)).numpy()
            rows[label] = (float(np.abs(lg - want).max() / scale), bool((lg.argmax(1) == want.argmax(1)).all()))
    return rows


def _prod(a, b, correction):
    """a [M, K] . b [N, K]^T with both operands split along K's rows: the bf16x3 pairings, corrections in bf16 or e4m3."""
    a0, a1 = _pieces(a)
    b0, b1 = _pieces(b)
    if correction == "bf16":
        return a0 @ b0.t() + a0 @ b1.t() + a1 @ b0.t()
    return a0 @ b0.t() + _e4m3_rows(a0) @ _e4m3_rows(b1).t() + _e4m3_rows(a1) @ _e4m3_rows(b0).t()


class _EmuLinear(torch.autograd.Function):
    """nn.Linear whose three products (forward, dX = dY W, dW = dY^T X) all run the emulated scheme."""

    @staticmethod
    def forward(ctx, x, w, bias, correction):
        ctx.save_for_backward(x, w)
        ctx.correction = correction
        return _prod(x.reshape(-1, x.shape[-1]), w, correction).view(*x.shape[:-1], w.shape[0]) + bias

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        dx = _prod(dy2, w.t().contiguous(), ctx.correction).view_as(x)
        dw = _prod(dy2.t().contiguous(), x2.t().contiguous(), ctx.correction)
        return dx, dw, dy2.sum(0), None


def fp8_correction_gradient_study(name, n_images):
    """-> {scheme: (worst relative L2 error over the gradient tensors against exact fp32 autograd, name of that tensor)}."""
    from oracle.detinit import det_labels
    arrays, meta = load_golden(name)
    cfg = ViTConfig(patch_size=16, **meta["kwargs"])
    img = det_images(name, meta["batch"], cfg.image_size)[:n_images]
    labels = det_labels(name, (meta["batch"],), cfg.num_classes)[:n_images]

    def grads(linear_fn):
        params = {k: v.clone().requires_grad_(True) for k, v in det_state_dict(cfg.param_shapes()).items()}
        loss = torch.nn.functional.cross_entropy(vit_forward(params, img, cfg, linear_fn=linear_fn), labels)
        loss.backward()
        return {k: v.grad for k, v in params.items() if v.grad is not None}

    exact = grads(None)
    out = {}
    for label, corr in [("bf16x3 (three bf16 products)", "bf16"), ("bf16 main product + two e4m3 correction products", "e4m3")]:
        got = grads(lambda n, x, w, b, c=corr: _EmuLinear.apply(x, w, b, c))
        errs = {k: float((got[k] - exact[k]).norm() / exact[k].norm().clamp_min(1e-30)) for k in exact if float(exact[k].norm()) > 0}
        worst = max(errs, key=errs.get)
        out[label] = (errs[worst], worst)
    return out


def test_fp8_correction_products_would_stay_inside_the_tolerance():
    """ViT-Tiny, 8 images (forward): the emulated scheme's logits stay inside north_star's 1e-3 with every arg-max equal, within 30x
    of bf16x3 itself -- the feasibility bound quoted in DESIGN section 8 for a next round's kernel, not a claim about a shipped path."""
    torch.set_num_threads(8)
    rows = fp8_correction_study("tiny_cls_b64", 8)
    x3, f8 = rows["bf16x3 (three bf16 products)"], rows["bf16 main product + two e4m3 correction products"]
    assert x3[0] < 1e-4 and x3[1]
    assert f8[0] < 1e-3 and f8[1]
    g = fp8_correction_gradient_study("tiny_cls_b64", 4)
    assert g["bf16x3 (three bf16 products)"][0] < 1e-4 and g["bf16 main product + two e4m3 correction products"][0] < 1e-3


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
    print("# split-operand products with FP8 correction terms (emulated): logits error, every arg-max equal")
    for k, (v, same) in fp8_correction_study(name, n).items():
        print(f"{v:.3e}  {same}  {k}")
    print("# the same schemes in all three products of every nn.Linear: worst relative L2 error of a gradient tensor vs exact autograd")
    for k, (v, which) in fp8_correction_gradient_study(name, min(n, 8)).items():
        print(f"{v:.3e}  ({which})  {k}")
