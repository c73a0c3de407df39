#!/usr/bin/env python3
"""Diagnostic (GPU): does the NT GEMM's L2 -> LDS stream depend on the operands' row stride?  Rows of K = 768 bf16 are 1536 B =
12 lines apart; if the L2 channel of a line is a simple function of its index, the 128 rows of a slot hit only a few channels.
Times mv_gemm_nt_bf16 (bf16 out, bias) with A and B stored with leading dimension K + pad."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import torch
from myrtle_vision.hip import ops
from myrtle_vision.hip.lib import lib, check, MV_BF16

M = int(os.environ.get("M", 50432))
dev = "cuda"


def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


which = os.environ.get("WHICH", "both")          # both | a | b: which operand gets the padded leading dimension
for N, K in [(2304, 768), (3072, 768), (768, 3072), (768, 768), (768, 2304)]:
    res = []
    for pad in (0, 64, 128, 256):
        lda = K + (pad if which in ("both", "a") else 0)
        ldb = K + (pad if which in ("both", "b") else 0)
        a = (torch.randn(M, lda, device=dev) * 0.5).bfloat16()
        w = (torch.randn(N, ldb, device=dev) * K ** -0.5).bfloat16()
        bias = torch.randn(N, device=dev)
        out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        s = torch.cuda.current_stream().cuda_stream
        f = lambda: check(lib().mv_gemm_nt_bf16(a.data_ptr(), lda, w.data_ptr(), ldb, out.data_ptr(), N, MV_BF16, M, N, K,
                                                bias.data_ptr(), 0, None, 0, 0, None, 0, s), "nt")
        best = min(timeit(f) for _ in range(3))
        res.append((pad, 2.0 * M * N * K / best / 1e12))
        del a, w, out
    print(f"[{which}] N={N} K={K}: " + "  ".join(f"pad {p}: {t:7.1f} TF" for p, t in res))
