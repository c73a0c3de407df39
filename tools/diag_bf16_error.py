#!/usr/bin/env python3
"""Diagnostic (GPU): per-parameter relative L2 error of bf16-mode gradients against fp32-mode gradients
(fp32 mode is pinned to the reference by tests/test_vit_parity.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "myrtle-vision_amd"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import torch
from test_vit_parity import build
from myrtle_vision.hip.functional import cross_entropy

for name in sys.argv[1:] or ["micro_cls", "micro_seg", "tiny_cls", "base_cls"]:
    res = {}
    for prec in ("fp32", "bf16"):
        vit, img, labels, arrays, meta = build(name, prec)
        logits = vit(img)
        loss = cross_entropy(logits, labels)
        loss.backward()
        res[prec] = (logits.detach().float(), {k: p.grad.float() for k, p in vit.named_parameters() if p.grad is not None})
    l32, l16 = res["fp32"][0], res["bf16"][0]
    print(f"== {name}: logits rel-max {float((l32-l16).abs().max()/l32.abs().max()):.3e}  rel-l2 {float((l32-l16).norm()/l32.norm()):.3e}")
    rows = []
    for k, g32 in res["fp32"][1].items():
        g16 = res["bf16"][1][k]
        rows.append((float((g32 - g16).norm() / g32.norm().clamp_min(1e-30)), float((g32 - g16).abs().max() / g32.abs().max().clamp_min(1e-30)), k))
    rows.sort(reverse=True)
    for r in rows[:6]:
        print(f"   grad rel-l2 {r[0]:.3e} rel-max {r[1]:.3e}  {r[2]}")
    print(f"   median rel-l2 {sorted(r[0] for r in rows)[len(rows)//2]:.3e}")
