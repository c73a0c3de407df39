#!/usr/bin/env python3
"""Loop ONE kernel for a few seconds (for tools/clock_watch.sh: which kernels pull the shader clock down?).
usage: python3 tools/clock_kernel_loop.py {nt_qkv|nt_fc2|tn_fc1|ln|attn|copy} [seconds]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import torch
from myrtle_vision.hip import ops

kind, secs = sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 5.0
M, dev = 50432, "cuda"
rnd = lambda *s: (torch.randn(*s, device=dev) * 0.5).to(torch.bfloat16)
flops = byts = 0
if kind in ("nt_qkv", "nt_fc2"):
    N, K = (2304, 768) if kind == "nt_qkv" else (768, 3072)
    sets = [(rnd(M, K), torch.empty(M, N, device=dev, dtype=torch.bfloat16)) for _ in range(4)]
    w, b = torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev)
    fns = [lambda x=x, o=o: ops.linear_fwd(x, M, K, w, b, o, N) for x, o in sets]
    flops = 2.0 * M * N * K
elif kind == "tn_fc1":
    N, K = 3072, 768
    sets = [(rnd(M, N), rnd(M, K)) for _ in range(2)]
    fns = [lambda dy=dy, x=x: ops.linear_dw(dy, x, M, N, K) for dy, x in sets]
    flops = 2.0 * M * N * K
elif kind == "ln":
    D = 768
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    sets = [torch.randn(M, D, device=dev) for _ in range(4)]
    fns = [lambda x=x: ops.layernorm_fwd(x, D, M, D, g, b, torch.bfloat16) for x in sets]
    byts = M * D * 6.0
elif kind == "attn":
    B, H, S, Dh = 256, 12, 197, 64
    sets = [rnd(B, S, 3 * H * Dh) for _ in range(4)]
    fns = [lambda q=q: ops.attention_fwd(q, B, S, H, Dh ** -0.5) for q in sets]
    flops = 4.0 * B * H * S * S * Dh
else:
    sets = [(torch.randn(M, 3072, device=dev), torch.empty(M, 3072, device=dev)) for _ in range(2)]
    fns = [lambda a=a, o=o: o.copy_(a) for a, o in sets]
    byts = M * 3072 * 8.0
for f in fns: f()
torch.cuda.synchronize()
t0, n = time.time(), 0
while time.time() - t0 < secs:
    for _ in range(50):
        fns[n % len(fns)](); n += 1
    torch.cuda.synchronize()
dt = (time.time() - t0) / n
print(f"{kind}: {dt * 1e6:.1f} us/launch  " + (f"{flops / dt / 1e12:.1f} TFLOP/s" if flops else f"{byts / dt / 1e9:.0f} GB/s"))
