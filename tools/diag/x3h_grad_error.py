#!/usr/bin/env python3
"""Per-tensor gradient error of precision "bf16x3h" (and "bf16") against the fp32 mode on a golden fixture's model (GPU):
relative L2 per parameter tensor, and the relative error of the plain SUM over the tensor (what the fixtures' summaries hold)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd")); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from test_vit_parity import build
from myrtle_vision.hip.functional import cross_entropy
name = sys.argv[1] if len(sys.argv) > 1 else "base_cls"
res = {}
for prec in ("fp32", "bf16x3", "bf16x3h", "bf16"):
    vit, img, labels, _, _ = build(name, prec)
    vit.train()
    cross_entropy(vit(img), labels).backward()
    res[prec] = {k: p.grad.double() for k, p in vit.named_parameters() if p.grad is not None}
for prec in ("bf16x3", "bf16x3h", "bf16"):
    worst = sorted(((float((res[prec][k] - res["fp32"][k]).norm() / res["fp32"][k].norm()), k) for k in res["fp32"]), reverse=True)
    sums = sorted(((float(abs((res[prec][k] - res["fp32"][k]).sum()) / res["fp32"][k].norm()), k) for k in res["fp32"]), reverse=True)
    print(f"{name} {prec}: worst rel-L2 {worst[0][0]:.3e} ({worst[0][1]}), median {worst[len(worst)//2][0]:.3e}; "
          f"worst |sum err| / l2 {sums[0][0]:.3e} ({sums[0][1]})")
    for e, k in worst[:4]:
        print(f"    {e:.3e} {k}")
