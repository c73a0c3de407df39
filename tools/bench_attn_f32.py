#!/usr/bin/env python3
"""Microbenchmark of the fused fp32 attention kernels (f32 MFMA) at ViT-B shapes (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import torch
from myrtle_vision.hip import ops
B, N, H = int(os.environ.get("B", 256)), int(os.environ.get("N", 197)), 12
qkv = torch.randn(B, N, 3 * H * 64, device="cuda") * 0.8
dout = torch.randn(B, N, H * 64, device="cuda")
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
out, lse = ops.attention_fwd_f32_lse(qkv, B, N, H, 0.125)
tf = timeit(lambda: ops.attention_fwd_f32(qkv, B, N, H, 0.125))
tb = timeit(lambda: ops.attention_bwd_f32_fused(qkv, out, dout, lse, B, N, H, 0.125))
fl_f, fl_b = 4.0 * B * H * N * N * 64, 14.0 * B * H * N * N * 64
print(f"fp32 attention fwd {tf:8.1f} us  {fl_f/tf/1e6:7.1f} TFLOP/s | bwd {tb:8.1f} us  {fl_b/tb/1e6:7.1f} TFLOP/s (7 products)  (B={B}, N={N}, H={H})")
