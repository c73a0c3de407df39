#!/usr/bin/env python3
"""fp32 GEMM microbenchmark on the ViT-B shapes: FMA kernel, f32-input MFMA kernels (generic / fast) and the bf16x6 products
(fp32-equivalent TFLOP/s), one process."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import torch
from myrtle_vision.hip import ops
from myrtle_vision.hip.lib import lib

M = int(os.environ.get("M", 197 * 64))


def timeit(fn, iters=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


rows = []
for name, N, K in [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072)]:
    x, w, b = torch.randn(M, K, device="cuda"), torch.randn(N, K, device="cuda"), torch.randn(N, device="cuda")
    dy, out, dx = torch.randn(M, N, device="cuda"), torch.empty(M, N, device="cuda"), torch.empty(M, K, device="cuda")
    for label, fn in [("fwd", lambda: ops.linear_fwd(x, M, K, w, b, out, N)), ("dx", lambda: ops.linear_dx(dy, M, N, w, dx, K)),
                      ("dw", lambda: ops.linear_dw(dy, x, M, N, K))]:
        t = {}
        ops.set_f32_gemm("mfma")
        for mode in (1, 2, 0):
            lib().mv_gemm_f32_force_fma(mode)
            t[mode] = timeit(fn)
        lib().mv_gemm_f32_force_fma(0)
        ops.set_f32_gemm("bf16x6")
        t[3] = timeit(fn)                      # includes the operand splits (the weight's is cached, as in a step)
        fl = 2.0 * M * N * K
        rows.append((f"{name} {label}", fl / t[1] / 1e12, fl / t[2] / 1e12, fl / t[0] / 1e12, fl / t[3] / 1e12))
B, H, N_, dh = 64, 12, 197, 64
qkv = torch.randn(B, N_, 3 * H * dh, device="cuda")
for mode in (1, 0):
    lib().mv_gemm_f32_force_fma(mode)
    t = timeit(lambda: ops.attention_probs_fp32(qkv, B, N_, H, dh, 0.125))
    rows.append((f"attn probs ({'fma' if mode else 'mfma'})", 2.0 * B * H * N_ * N_ * dh / t / 1e12, 0.0, 0.0, 0.0))
lib().mv_gemm_f32_force_fma(0)
print(f"{'shape (M=%d)' % M:24s} {'FMA':>8s} {'generic':>8s} {'fast':>8s} {'bf16x6':>8s}  fp32-equivalent TFLOP/s")
for r in rows:
    print(f"{r[0]:24s} {r[1]:8.1f} {r[2]:8.1f} {r[3]:8.1f} {r[4]:8.1f}")
