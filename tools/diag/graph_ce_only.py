#!/usr/bin/env python3
"""Diagnostic (GPU): mv_cross_entropy alone in a captured graph, replayed; and a plain torch memset of a small tensor."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "myrtle-vision_amd")):
    sys.path.insert(0, p)
import torch
from myrtle_vision.hip import ops
B, C = int(os.environ.get("B", 32)), 1000
g = torch.Generator().manual_seed(1)
logits = torch.randn(B, C, generator=g).cuda()
labels = torch.randint(0, C, (B,), generator=g).cuda()
want = float(torch.nn.functional.cross_entropy(logits, labels))
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    for _ in range(3):
        ops.cross_entropy(logits, labels, want_grad=True)
torch.cuda.synchronize()
gr = torch.cuda.CUDAGraph()
with torch.cuda.graph(gr):
    junk = [torch.empty(4, device="cuda").fill_(float(i)) for i in range(8)]       # neighbours in the small-block pool
    loss, dl, _ = ops.cross_entropy(logits, labels, want_grad=True)
    keep = loss.clone()
    junk2 = [torch.empty(4, device="cuda").fill_(1e30) for i in range(8)]
for i in range(4):
    logits.add_(0.0)
    gr.replay()
    torch.cuda.synchronize()
    print(i, "graph loss", float(loss), "clone", float(keep), "want", want, "stat", loss._base.tolist() if loss._base is not None else None)
