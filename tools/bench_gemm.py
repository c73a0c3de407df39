#!/usr/bin/env python3
"""Per-shape microbenchmark of the MFMA GEMM entry points on random data (GPU).  TFLOP/s per ViT-B layer shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import torch
from myrtle_vision.hip import ops

M = int(os.environ.get("M", 50432))
ROT = int(os.environ.get("ROTATE", "1"))       # ROTATE=4: cycle through 4 buffer sets (> Infinity Cache), as inside a real step
dev = "cuda"
def rnd(*s, dt=torch.bfloat16): return (torch.randn(*s, device=dev) * 0.5).to(dt)

class Rot:
    """fn factory over ROT independent buffer sets: call i uses set i % ROT."""
    def __init__(self, make):
        self.fns = [make() for _ in range(ROT)]
        self.i = 0
    def __call__(self):
        self.fns[self.i % ROT]()
        self.i += 1

def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

# NT variants A/B-ed in ONE process on one device, rounds interleaved (timings from different boxes differ by up to 10 %)
VARIANTS = [int(v) for v in os.environ.get("NT_VARIANTS", "0").split(",")]
from myrtle_vision.hip.lib import lib as _lib

def timeit_variants(fn, rounds=3, iters=8):
    best = {v: 1e9 for v in VARIANTS}
    for _ in range(rounds):
        for v in VARIANTS:
            _lib().mv_gemm_force_variant(v, 0)
            best[v] = min(best[v], timeit(fn, iters))
    _lib().mv_gemm_force_variant(0, 0)
    return best

rows = []
for name, N, K, epi in [("qkv fwd", 2304, 768, "none"), ("proj fwd +res", 768, 768, "res"), ("fc1 fwd +gelu", 3072, 768, "gelu"),
                        ("fc1 fwd +gelu+grad", 3072, 768, "gelugrad"),
                        ("fc2 fwd +res", 768, 3072, "res"), ("plain bf16 out N=3072", 3072, 768, "none"), ("plain K=3072", 768, 3072, "none"),
                        ("plain K=3072 fp32 out", 768, 3072, "none32"), ("plain N=K=768", 768, 768, "none"),
                        ("plain N=K=768 fp32 out", 768, 768, "none32")]:
    w, b = torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev)
    def make(N=N, K=K, epi=epi, w=w, b=b):
        x = rnd(M, K)
        if epi == "none":
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            return lambda: ops.linear_fwd(x, M, K, w, b, out, N)
        if epi == "none32":                 # fp32 output without the residual read: separates the epilogue's store from its load
            out = torch.empty(M, N, device=dev)
            return lambda: ops.linear_fwd(x, M, K, w, b, out, N)
        if epi == "res":
            out, res = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
            return lambda: ops.linear_fwd(x, M, K, w, b, out, N, epi=ops.EPI_RESIDUAL, aux=res, ld_aux=N)
        out, h = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.bfloat16)
        code = ops.EPI_GELU if epi == "gelu" else ops.EPI_GELU_GRAD
        return lambda: ops.linear_fwd(x, M, K, w, b, out, N, epi=code, out2=h, ld_out2=N)
    f = Rot(make)
    tv = timeit_variants(f); rows.append((f"NT {name}", [2.0 * M * N * K / tv[v] / 1e12 for v in VARIANTS], tv[VARIANTS[-1]] * 1e6))
for name, N, K, epi in [("dX qkv (N=768,K=2304)", 2304, 768, "none"), ("dX fc2 +dgelu (->3072)", 768, 3072, "dgelu"),
                        ("dX fc2 +mul   (->3072)", 768, 3072, "mul"), ("dX fc1 (K=3072)", 3072, 768, "none")]:
    # linear_dx(dy[M,N], W[N,K]) -> [M,K]
    dy, w = rnd(M, N), torch.randn(N, K, device=dev) * K ** -0.5
    out = torch.empty(M, K, device=dev, dtype=torch.bfloat16)
    if epi in ("dgelu", "mul"):
        h = rnd(M, K)
        part = torch.empty((M + 63) // 64, K, device=dev)
        code = ops.EPI_DGELU if epi == "dgelu" else ops.EPI_MUL
        f = lambda: ops.linear_dx(dy, M, N, w, out, K, epi=code, aux=h, ld_aux=K, colsum_partial=part)
    else:
        f = lambda: ops.linear_dx(dy, M, N, w, out, K)
    tv = timeit_variants(f); rows.append((f"NT {name}", [2.0 * M * N * K / tv[v] / 1e12 for v in VARIANTS], tv[VARIANTS[-1]] * 1e6))
if os.environ.get("NT_ONLY"):
    print(f"{'shape':38s} " + " ".join(f"{v:>8d}" for v in VARIANTS) + "   (TFLOP/s per forced NT variant; 0 = automatic)")
    for r in rows:
        print(f"{r[0]:38s} " + " ".join(f"{x:8.1f}" for x in r[1]) + f"   {r[2]:8.1f} us")
    sys.exit(0)
rows = [(r[0], r[1][-1], r[2]) for r in rows]
TN_VARIANTS = [int(v) for v in os.environ.get("TN_VARIANTS", "0").split(",")]
trows = []
for name, N, K in [("dW qkv", 2304, 768), ("dW proj", 768, 768), ("dW fc1", 3072, 768), ("dW fc2", 768, 3072)]:
    dy, x = rnd(M, N), rnd(M, K)
    for label, f in [("(+reduce+colsum)", lambda: ops.linear_dw(dy, x, M, N, K)),
                     ("(+reduce)", lambda: ops.linear_dw(dy, x, M, N, K, want_bias=False))]:
        best = {v: 1e9 for v in TN_VARIANTS}
        for _ in range(3):
            for v in TN_VARIANTS:
                _lib().mv_gemm_force_variant(0, v)
                best[v] = min(best[v], timeit(f, 8))
        _lib().mv_gemm_force_variant(0, 0)
        trows.append((f"TN {name} {label}", [2.0 * M * N * K / best[v] / 1e12 for v in TN_VARIANTS], best[TN_VARIANTS[-1]] * 1e6))
for r in rows:
    print(f"{r[0]:38s} {r[1]:8.1f} TFLOP/s  {r[2]:9.1f} us")
print(f"{'shape':38s} " + " ".join(f"{v:>8d}" for v in TN_VARIANTS) + "   (TFLOP/s per forced TN variant; 0 = automatic)")
for r in trows:
    print(f"{r[0]:38s} " + " ".join(f"{x:8.1f}" for x in r[1]) + f"   {r[2]:8.1f} us")
