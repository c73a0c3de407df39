#!/usr/bin/env python3
"""Which device-to-device copies does one bf16 training step issue?  (torch profiler, grouped by the Python stack line that issued them.)"""
import os, sys, collections
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "myrtle-vision_amd"))
import torch
from torch.profiler import profile, ProfilerActivity
from myrtle_vision.hip.functional import cross_entropy
from myrtle_vision.models.vit import ViT
from myrtle_vision.utils.optim import AdamW, ParamArena

dev = torch.device("cuda", 0)
vit = ViT(decoder="classification", image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072,
          precision=os.environ.get("PRECISION", "bf16"), q_format="FP32").to(dev)
opt = AdamW(ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names()), lr=1e-4)
img, lab = torch.randn(64, 3, 224, 224, device=dev), torch.randint(0, 1000, (64,), device=dev)
vit.train()
def step():
    opt.zero_grad(); cross_entropy(vit(img), lab).backward(); opt.step()
for _ in range(3): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(); torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::fill_", "aten::zero_", "aten::mul_", "aten::add_"):
        st = [s for s in (ev.stack or []) if "myrtle" in s or "bench" in s or "find_copies" in s]
        cnt[(ev.name, tuple(ev.input_shapes) if ev.input_shapes else None, st[0] if st else "?")] += 1
for (name, shp, where), n in sorted(cnt.items(), key=lambda kv: -kv[1])[:25]:
    print(n, name, shp, where)
