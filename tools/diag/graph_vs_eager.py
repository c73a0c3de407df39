#!/usr/bin/env python3
"""Diagnostic (GPU): graphed step vs eager step at a given model size; prints per-step loss and parameter differences."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "myrtle-vision_amd")):
    sys.path.insert(0, p)
import torch
from myrtle_vision.hip.functional import cross_entropy
from myrtle_vision.models.vit import ViT
from myrtle_vision.utils.graph import GraphedTrainStep
from myrtle_vision.utils.optim import AdamW, ParamArena
from myrtle_vision.utils.utils import seed_everything

dim, depth, heads, mlp, B, nc = (int(os.environ.get(k, d)) for k, d in (("DIM", 768), ("DEPTH", 2), ("HEADS", 12), ("MLP", 3072), ("B", 32), ("NC", 1000)))
kw = dict(decoder="classification", image_size=224, patch_size=16, num_classes=nc, dim=dim, depth=depth, heads=heads, mlp_dim=mlp)
g = torch.Generator().manual_seed(9)
batches = [(torch.randn(B, 3, 224, 224, generator=g).cuda(), torch.randint(0, nc, (B,), generator=g).cuda()) for _ in range(3)]
loss_fn = lambda m, x, y: cross_entropy(m(x), y)


def build():
    seed_everything(21)
    vit = ViT(precision="bf16", q_format="FP32", **kw).cuda().train()
    opt = AdamW(ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names()), lr=6.25e-5, weight_decay=0.05)
    return vit, opt


vit_e, opt_e = build()
if os.environ.get("DEV_SCALARS_EAGER"):
    opt_e.use_device_scalars()
le = []
for i in [0, 0, 0, 1, 2, 1, 2]:
    opt_e.zero_grad()
    loss = loss_fn(vit_e, *batches[i])
    loss.backward()
    opt_e.step()
    le.append(float(loss))
vit_g, opt_g = build()
gs = GraphedTrainStep(vit_g, opt_g, loss_fn, *batches[0], warmup=3)
torch.cuda.synchronize()
print("after warm-up: params equal to eager after 3 steps?", end=" ")
vit_c, opt_c = build()
for i in [0, 0, 0]:
    opt_c.zero_grad(); loss_fn(vit_c, *batches[i]).backward(); opt_c.step()
print(bool(torch.equal(opt_c.arena.flat_param, opt_g.arena.flat_param)))
lg = []
for i in (1, 2, 1, 2):
    lg.append(float(gs(*batches[i])))
    print("  hyper", {k: v.tolist() for k, v in opt_g._hyper.items()}, "step", opt_g.step_count,
          "param finite", bool(torch.isfinite(opt_g.arena.flat_param).all()), "grad finite", bool(torch.isfinite(opt_g.arena.flat_grad).all()),
          "grad norm", float(opt_g.arena.flat_grad.norm()))
print("eager losses", le[3:])
print("graph losses", lg)
print("param max diff", float((opt_g.arena.flat_param - opt_e.arena.flat_param).abs().max()))
