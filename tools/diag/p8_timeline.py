#!/usr/bin/env python3
"""Per-workgroup timeline of gemm_nt_8phase_kernel (GPU; needs the MV_ABLATE=32 build: MODES=32 tools/ablate_gemm8.sh, run with
MV_LIB_PATH=tools/_ablate/libgemm_ablate32.so).  Each workgroup stamps s_memrealtime (100 MHz) at entry, at its first MFMA phase,
at the end of its main loop, when its stores are issued and when they are acknowledged; the table says where a tile's time goes,
how far apart the compute units run, and how long a unit sits between two workgroups."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import numpy as np, torch
from myrtle_vision.hip import ops
from myrtle_vision.hip.lib import lib as _lib
dbg = ctypes.CDLL(os.environ["MV_LIB_PATH"])
M, dev = 50432, "cuda"
def rnd(*s, dt=torch.bfloat16): return (torch.randn(*s, device=dev) * 0.5).to(dt)

def cases():
    for name, N, K, epi in [("qkv fwd", 2304, 768, "none"), ("plain N=3072", 3072, 768, "none"), ("fc1 gelu+grad8", 3072, 768, "gg8"),
                            ("fc2 +res (K=3072)", 768, 3072, "res")]:
        x, w, b = rnd(M, K), torch.randn(N, K, device=dev) * K ** -0.5, torch.randn(N, device=dev)
        if epi == "none":
            out = torch.empty(M, N, device=dev, dtype=torch.bfloat16)
            yield name, N, K, (lambda x=x, w=w, b=b, out=out, N=N, K=K: ops.linear_fwd(x, M, K, w, b, out, N))
        elif epi == "res":
            out, res = torch.empty(M, N, device=dev), torch.randn(M, N, device=dev)
            yield name, N, K, (lambda x=x, w=w, b=b, out=out, res=res, N=N, K=K:
                               ops.linear_fwd(x, M, K, w, b, out, N, epi=ops.EPI_RESIDUAL, aux=res, ld_aux=N))
        else:
            out, h = torch.empty(M, N, device=dev, dtype=torch.bfloat16), torch.empty(M, N, device=dev, dtype=torch.uint8)
            yield name, N, K, (lambda x=x, w=w, b=b, out=out, h=h, N=N, K=K:
                               ops.linear_fwd(x, M, K, w, b, out, N, epi=ops.EPI_GELU_GRAD8, out2=h, ld_out2=N))
    dy, w = rnd(M, 768), torch.randn(768, 3072, device=dev) * 768 ** -0.5
    out, h = torch.empty(M, 3072, device=dev, dtype=torch.bfloat16), torch.randint(0, 255, (M, 3072), device=dev, dtype=torch.uint8)
    part = torch.empty((M + 63) // 64, 3072, device=dev)
    yield "dX fc2 * gelu'8", 3072, 768, (lambda: ops.linear_dx(dy, M, 768, w, out, 3072, epi=ops.EPI_MUL8, aux=h, ld_aux=3072, colsum_partial=part))

buf = np.zeros(8 * 4096, dtype=np.uint64)
variants = [int(v) for v in os.environ.get("NT_VARIANTS", "0").split(",")]
for name, N, K, fn in cases():
    for v in variants:
        _lib().mv_gemm_force_variant(v, 0)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        fn(); torch.cuda.synchronize()
        assert dbg.mv_debug_p8_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
        tiles = ((M + 255) // 256) * (N // 256)
        tr = buf.reshape(4096, 8)[: min(4096, tiles + 256)].astype(np.int64)
        tr = tr[tr[:, 0] > 0]
        t = (tr[:, :5] - tr[:, 0].min()) * 0.01                      # us from the first workgroup's entry
        hw, xcc, half = tr[:, 5], tr[:, 6] & 15, tr[:, 7] >> 32
        cu = ((xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15))   # xcc | se | sh | cu
        full = half == 0
        pro, main, epi, drain = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
        span = t[:, 4].max()
        gaps = []
        for c in np.unique(cu):
            idx = np.where(cu == c)[0]
            o = idx[np.argsort(t[idx, 0])]
            gaps += list(t[o[1:], 0] - t[o[:-1], 4])
        gaps = np.array(gaps)
        def st(a): return f"{a.mean():6.2f} (p10 {np.percentile(a, 10):5.2f} p90 {np.percentile(a, 90):5.2f})"
        print(f"{name:20s} variant {v:5d}: span {span:7.1f} us, {len(t)} workgroups ({int(full.sum())} whole) on {len(np.unique(cu))} units | "
              f"per whole tile [us]: fill {st(pro[full])}  main {st(main[full])}  epilogue issue {st(epi[full])}  store drain {st(drain[full])}"
              f"  idle between workgroups {st(gaps)}")
        # how synchronised is the chip?  fraction of the span during which >= 75 % / <= 25 % of the units are in their epilogue
        grid = np.arange(0, span, 0.05)
        in_epi = ((grid[None, :] >= t[:, 2:3]) & (grid[None, :] < t[:, 4:5])).sum(0) / len(np.unique(cu))
        print(f"{'':20s}   units inside epilogue+drain at a time: mean {in_epi.mean():.2f}; share of time with > 0.75 of the chip there "
              f"{(in_epi > 0.75).mean():.2f}, with < 0.25 {(in_epi < 0.25).mean():.2f}")
_lib().mv_gemm_force_variant(0, 0)
