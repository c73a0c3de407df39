#!/usr/bin/env python3
"""Does the attention backward's time depend on WHERE its four tensors sit relative to each other?  (GPU)
The standard microbenchmark (fresh process, torch.empty_like output) reads 225-250 us, the same kernel inside an A/B process
190-196: the buffers differ only in their addresses.  Here qkv, out, dout and dqkv are carved out of one arena at controlled
offsets: the output is moved by PAD bytes (a sweep), everything else fixed."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import torch
from myrtle_vision.hip import ops
from myrtle_vision.hip.lib import lib
from myrtle_vision.hip.ops import _p, _s, check
B, N, H = 256, 197, 12
D = H * 64
n_qkv, n_out = B * N * 3 * D, B * N * D
arena = torch.empty((2 * n_qkv + 2 * n_out) * 2 + (64 << 20), dtype=torch.uint8, device="cuda")
base = arena.data_ptr()
def carve(off, n):
    return arena[off: off + 2 * n].view(torch.bfloat16)
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
al = lambda x, a=2 << 20: (x + a - 1) // a * a
o_qkv = al(base) - base
o_out = al(o_qkv + 2 * n_qkv)
o_dout = al(o_out + 2 * n_out)
o_dq = al(o_dout + 2 * n_out)
qkv = carve(o_qkv, n_qkv).view(B, N, 3 * D); qkv.copy_((torch.randn(B, N, 3 * D, device="cuda") * 0.8).to(torch.bfloat16))
dout = carve(o_dout, n_out).view(B, N, D); dout.copy_(torch.randn(B, N, D, device="cuda").to(torch.bfloat16))
out_t, lse = ops.attention_fwd(qkv, B, N, H, 0.125)
out = carve(o_out, n_out).view(B, N, D); out.copy_(out_t)
print(f"arena base % 2 MiB = {base % (2 << 20)}; qkv/out/dout at 2 MiB-aligned offsets")
for pad in [0, 256, 1024, 4096, 16384, 65536, 262144, 1 << 20, (1 << 20) + 4096, 3 << 19, (2 << 20) + 8192, 5 << 20, (8 << 20) + 65536, 16 << 20, 33 << 20]:
    dq = carve(o_dq + pad, n_qkv).view(B, N, 3 * D)
    f = lambda: check(lib().mv_attention_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(dq), None, B, N, H, 0.125, _s()), "attention_bwd")
    ts = [timeit(f) for _ in range(3)]
    print(f"dqkv offset pad {pad:>9d} B: {min(ts):7.1f} .. {max(ts):7.1f} us")
# and the way the standard benchmark allocates
ts = [timeit(lambda: ops.attention_bwd(qkv, out, dout, lse, B, N, H, 0.125)) for _ in range(3)]
print(f"torch.empty_like output per call: {min(ts):7.1f} .. {max(ts):7.1f} us")
