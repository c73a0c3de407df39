#!/usr/bin/env python3
"""Microbenchmark of the fused attention kernels at the ViT-B benchmark shape (GPU)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import torch
from myrtle_vision.hip import ops
B, N, H = int(os.environ.get("B", 256)), int(os.environ.get("N", 197)), 12
qkv = (torch.randn(B, N, 3 * H * 64, device="cuda") * 0.8).to(torch.bfloat16)
dout = torch.randn(B, N, H * 64, device="cuda").to(torch.bfloat16)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
if os.environ.get("ATTN_BWD"):                      # 2 two-pass | 4 dS-through-LDS | 8 eight-wave (mv_attention_bwd_force)
    from myrtle_vision.hip.lib import lib
    lib().mv_attention_bwd_force(int(os.environ["ATTN_BWD"]))
out, lse = ops.attention_fwd(qkv, B, N, H, 0.125)
if os.environ.get("FWD_AB"):                        # forward variants 1 (one query tile per wave pass) and 2 (pairs), alternating
    from myrtle_vision.hip.lib import lib
    best = {1: 1e9, 2: 1e9, 3: 1e9}
    outs = {}
    for _ in range(4):
        for v in (1, 2, 3):
            lib().mv_attention_fwd_force(v)
            best[v] = min(best[v], timeit(lambda: ops.attention_fwd(qkv, B, N, H, 0.125)))
            outs[v] = ops.attention_fwd(qkv, B, N, H, 0.125)
    lib().mv_attention_fwd_force(0)
    d = float((outs[1][0].float() - outs[2][0].float()).abs().max())
    d3 = float((outs[1][0].float() - outs[3][0].float()).abs().max())
    print(f"attention fwd variants: single tile {best[1]:7.1f} us | tile pairs {best[2]:7.1f} us | 13 tiles, 3 WG/CU {best[3]:7.1f} us | "
          f"max |o1 - o2| {d:.3e} |o1 - o3| {d3:.3e}  lse equal {bool(torch.equal(outs[1][1], outs[2][1]))} {bool(torch.equal(outs[1][1], outs[3][1]))}")
if os.environ.get("BWD_AB"):                        # backward variants alternating in one process, e.g. BWD_AB=4,5
    from myrtle_vision.hip.lib import lib
    vs = [int(v) for v in os.environ["BWD_AB"].split(",")]
    best, res = {v: 1e9 for v in vs}, {}
    for _ in range(4):
        for v in vs:
            lib().mv_attention_bwd_force(v)
            best[v] = min(best[v], timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, N, H, 0.125)))
            res[v] = ops.attention_bwd(qkv, out, dout, lse, B, N, H, 0.125).float()
    lib().mv_attention_bwd_force(0)
    print("attention bwd variants: " + " | ".join(f"{v}: {best[v]:7.1f} us" for v in vs) + " | max |d| vs first: " +
          " ".join(f"{float((res[v] - res[vs[0]]).abs().max()):.3e}" for v in vs[1:]))
tf = timeit(lambda: ops.attention_fwd(qkv, B, N, H, 0.125))
tb = timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, N, H, 0.125))
fl_f, fl_b = 4.0 * B * H * N * N * 64, 10.0 * B * H * N * N * 64
print(f"attention fwd {tf:8.1f} us  {fl_f/tf/1e6:7.1f} TFLOP/s | bwd {tb:8.1f} us  {fl_b/tb/1e6:7.1f} TFLOP/s  (B={B}, N={N}, H={H})")
