#!/usr/bin/env python3
"""The split-operand NT product three ways, per ViT-B shape (M = 50 432): bf16x3 (three bf16 segments: mv_gemm_nt_bf16 over 3 K),
the bf16 + e4m3-correction form (mv_gemm_nt_f8c) and plain bf16 for scale.  Rotating buffer sets; us per launch and bf16-equivalent
TFLOP/s (2 M N K / t)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "myrtle-vision_amd"))
import torch
from myrtle_vision.hip import ops

M, ROT = int(os.environ.get("M", 50432)), 3
dev = "cuda"

def timeit(fns, iters=8):
    for f in fns: f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fns[i % len(fns)]()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3

for name, N, K in [("qkv", 2304, 768), ("proj", 768, 768), ("fc1", 3072, 768), ("fc2", 768, 3072), ("dX qkv", 768, 2304)]:
    w = torch.randn(N, K, device=dev) * K ** -0.5
    xs = [torch.randn(M, K, device=dev) for _ in range(ROT)]
    outs = [torch.empty(M, N, device=dev) for _ in range(ROT)]
    ea, eb = ops.f8c_exponent(xs[0]) - 1, ops.f8c_exponent(w)
    w8 = ops.split_f8c(w, N, K, 1, eb)
    a8 = [ops.split_f8c(x, M, K, 0, ea) for x in xs]
    f8 = [lambda a=a, o=o: ops.gemm_nt_f8c(a, w8, M, N, K, ea, eb, o, N) for a, o in zip(a8, outs)]
    with ops.segments(3):
        a3 = [ops.split3(x, M, K, K, 0) for x in xs]
        w3 = ops.split3(w, N, K, K, 1)
        from myrtle_vision.hip.lib import lib, check
        def x3(a, o):
            check(lib().mv_gemm_nt_bf16(a.data_ptr(), 3 * K, w3.data_ptr(), 3 * K, o.data_ptr(), N, 0, M, N, 3 * K, None, 0, None, 0, 0,
                                        None, 0, torch.cuda.current_stream().cuda_stream), "nt x3")
        f3 = [lambda a=a, o=o: x3(a, o) for a, o in zip(a3, outs)]
        t3 = timeit(f3)
    x16 = [x.bfloat16() for x in xs]
    f16 = [lambda a=a, o=o: ops.linear_fwd(a, M, K, w, None, o, N) for a, o in zip(x16, outs)]
    t16, t8 = timeit(f16), timeit(f8)
    fl = 2.0 * M * N * K
    print(f"{name:8s} N {N:5d} K {K:5d}   bf16 {t16:7.1f} us ({fl / t16 / 1e6:6.0f} TF/s)   bf16x3 {t3:7.1f} us ({fl / t3 / 1e6:5.0f})   "
          f"bf16 + e4m3 corrections {t8:7.1f} us ({fl / t8 / 1e6:5.0f})   f8c / x3 = {t8 / t3:.2f}")
