#!/usr/bin/env python3
"""Tiny driver for rocprofv3 --pmc runs: a few launches of chosen GEMM shapes (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import torch
from myrtle_vision.hip import ops
M = 50432
dev = "cuda"
def rnd(*s): return (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
which = sys.argv[1] if len(sys.argv) > 1 else "all"
if which in ("all", "nt"):
    for N, K in [(2304, 768), (768, 3072)]:
        x, w, b = rnd(M, K), torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        for _ in range(3): ops.linear_fwd(x, M, K, w, b, out, N)
if which in ("all", "tn"):
    for N, K in [(3072, 768)]:
        dy, x = rnd(M, N), rnd(M, K)
        for _ in range(3): ops.linear_dw(dy, x, M, N, K, want_bias=False)
torch.cuda.synchronize()
