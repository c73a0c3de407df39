#!/usr/bin/env python3
"""Diagnostic (GPU): where does the captured step's loss scalar get clobbered?  Clones of it are captured after the forward,
after the backward and after the optimizer step."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "myrtle-vision_amd")):
    sys.path.insert(0, p)
import torch
from myrtle_vision.hip.functional import cross_entropy
from myrtle_vision.models.vit import ViT
from myrtle_vision.utils import graph as G
from myrtle_vision.utils.optim import AdamW, ParamArena
from myrtle_vision.utils.utils import seed_everything

B = int(os.environ.get("B", 32))
kw = dict(decoder="classification", image_size=224, patch_size=16, num_classes=1000, dim=768, depth=2, heads=12, mlp_dim=3072)
g = torch.Generator().manual_seed(9)
x, y = torch.randn(B, 3, 224, 224, generator=g).cuda(), torch.randint(0, 1000, (B,), generator=g).cuda()
seed_everything(21)
vit = ViT(precision="bf16", q_format="FP32", **kw).cuda().train()
opt = AdamW(ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names()), lr=6.25e-5, weight_decay=0.05)
snaps = {}


class Step(G.GraphedTrainStep):
    def _eager_step(self, advance=True):
        self.optimizer.zero_grad()
        loss = self.loss_fn(self.model, self.inputs, self.labels)
        snaps["ptr"] = loss.data_ptr()
        snaps["fwd"] = loss.detach().clone()
        loss.backward()
        snaps["bwd"] = loss.detach().clone()
        self.optimizer.step()
        snaps["opt"] = loss.detach().clone()
        return loss.detach()


gs = Step(vit, opt, lambda m, a, b: cross_entropy(m(a), b), x, y)
for _ in range(2):
    out = gs(x, y)
    torch.cuda.synchronize()
    print("loss", float(out), "| after fwd", float(snaps["fwd"]), "after bwd", float(snaps["bwd"]), "after opt", float(snaps["opt"]),
          "| ptr", hex(snaps["ptr"]), hex(out.data_ptr()))
# which tensors of the optimizer / arena / caches sit near that address?
near = []
from myrtle_vision.hip import ops
for name, t in [("flat_param", opt.arena.flat_param), ("flat_grad", opt.arena.flat_grad), ("exp_avg", opt.exp_avg), ("exp_avg_sq", opt.exp_avg_sq)]:
    near.append((name, hex(t.data_ptr()), hex(t.data_ptr() + t.numel() * t.element_size())))
for k, w in ops._workspaces.items():
    near.append((f"workspace{k}", hex(w.data_ptr()), hex(w.data_ptr() + w.numel())))
print(near)
