#!/usr/bin/env python3
"""Per-workgroup timeline of attn_bwd4_kernel (GPU; needs a -DMV_ATTN_TRACE=1 build of attention.hip linked into the library
given by MV_LIB_PATH).  Wave 0 of each workgroup stamps s_memrealtime (100 MHz) at entry, after the K/V prologue, at the end of the
query-pair loop, when its last stores are issued and when they are acknowledged, and adds up inside the loop the time spent in
the S phase, at the barrier after it, in the dQ phase and at the barrier after that."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import numpy as np, torch
from myrtle_vision.hip import ops
dbg = ctypes.CDLL(os.environ["MV_LIB_PATH"])
B, N, H = 256, 197, 12
qkv = (torch.randn(B, N, 3 * H * 64, device="cuda") * 0.8).to(torch.bfloat16)
dout = torch.randn(B, N, H * 64, device="cuda").to(torch.bfloat16)
out, lse = ops.attention_fwd(qkv, B, N, H, 0.125)
if os.environ.get("ATTN_BWD"):                      # 4: four waves per workgroup | 5: two waves of 512 registers
    from myrtle_vision.hip.lib import lib
    lib().mv_attention_bwd_force(int(os.environ["ATTN_BWD"]))
for _ in range(3): ops.attention_bwd(qkv, out, dout, lse, B, N, H, 0.125)
torch.cuda.synchronize()
ops.attention_bwd(qkv, out, dout, lse, B, N, H, 0.125); torch.cuda.synchronize()
buf = np.zeros(16 * 4096, dtype=np.uint64)
assert dbg.mv_debug_attn_trace(buf.ctypes.data_as(ctypes.c_void_p)) == 0
tr = buf.reshape(4096, 16)[: B * H].astype(np.int64)
t = (tr[:, :5] - tr[:, 0].min()) * 0.01
acc = tr[:, 5:9] * 0.01
hw, xcc = tr[:, 9], tr[:, 10] & 15
cu = (xcc << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 15)
def st(a): return f"{a.mean():6.2f} (p10 {np.percentile(a, 10):5.2f}, p90 {np.percentile(a, 90):5.2f})"
print(f"span {t[:, 4].max():.1f} us, {len(t)} workgroups on {len(np.unique(cu))} units")
print("per workgroup [us]: prologue", st(t[:, 1] - t[:, 0]), "| loop", st(t[:, 2] - t[:, 1]), "| tail stores issued", st(t[:, 3] - t[:, 2]),
      "| acknowledged", st(t[:, 4] - t[:, 3]), "| whole", st(t[:, 4] - t[:, 0]))
print("inside the loop (sum over 7 query pairs): S phase", st(acc[:, 0]), "| barrier", st(acc[:, 1]), "| dQ phase", st(acc[:, 2]),
      "| barrier", st(acc[:, 3]))
gaps, conc = [], []
for c in np.unique(cu):
    idx = np.where(cu == c)[0]
    o = idx[np.argsort(t[idx, 0])]
    conc.append(len(o))
    # two slots per unit: a workgroup starts when one of the two running before it has ended
    ends = sorted(t[o[:2], 4]) if len(o) >= 2 else [t[o[0], 4]]
    for k in o[2:]:
        e = ends.pop(0)
        gaps.append(t[k, 0] - e)
        ends.append(t[k, 4]); ends.sort()
print("workgroups per unit", min(conc), "..", max(conc), "| idle slot time between workgroups", st(np.array(gaps)))
