#!/usr/bin/env python3
"""LayerNorm backward as the training step calls it (bf16 dy, fp32 x / dx_add / dx, bf16 copy + column sums of dx), rows = 50 432,
dim = 768: microseconds per launch and HBM rate over 16 B / element, rotating over buffer sets larger than the Infinity Cache.
MV_LIB_PATH selects the library (a second build to compare against: see the LIBS mode of tools/ab_step.py)."""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "myrtle-vision_amd"))
import torch
from myrtle_vision.hip import ops

rows, dim, sets, reps = int(os.environ.get("ROWS", 50432)), 768, 4, 40
dev = "cuda"
g = torch.Generator(device=dev).manual_seed(0)
bufs = []
for _ in range(sets):
    x = torch.randn(rows, dim, device=dev, generator=g)
    dy = torch.randn(rows, dim, device=dev, generator=g).bfloat16()
    add = torch.randn(rows, dim, device=dev, generator=g)
    mean = x.mean(1).contiguous(); rstd = (x.var(1, unbiased=False) + 1e-5).rsqrt().contiguous()
    bufs.append((x, dy, add, mean, rstd, torch.empty_like(x), torch.empty(rows, dim, dtype=torch.bfloat16, device=dev)))
gamma = torch.randn(dim, device=dev, generator=g)
cs = torch.empty(dim, device=dev)

def one(i):
    x, dy, add, mean, rstd, dx, dx16 = bufs[i % sets]
    return ops.layernorm_bwd(dy, x, dim, gamma, mean, rstd, add, dx, dim, rows, dim, dx16=dx16, dx_colsum=cs)

for i in range(8): one(i)
torch.cuda.synchronize()
best = []
for trial in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for i in range(reps): one(i)
    e1.record(); torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / reps * 1e3)
us = min(best)                                       # (includes the ~7 us finishing launch)
x, dy, add, mean, rstd, dx, dx16 = bufs[0]
dg, db = one(0); torch.cuda.synchronize()
chk = float(dx.double().sum()), float(dx16.double().sum()), float(dg.double().sum()), float(cs.double().sum())
print(json.dumps({"lib": os.environ.get("MV_LIB_PATH", "default"), "us": round(us, 1), "TBps": round(rows * dim * 16 / us / 1e6, 2),
                  "trials": [round(b, 1) for b in best], "check": [round(c, 3) for c in chk]}))
