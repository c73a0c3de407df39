#!/usr/bin/env python3
"""A/B of GEMM kernel variants INSIDE the ViT-B/16 training step, one process, one device: the variants take turns in
blocks of STEPS steps for ROUNDS rounds (cdna guide rule 24); prints mean, std and min of ms/step per variant.

    NT_VARIANTS=3000,3002 ROUNDS=10 STEPS=10 python tools/ab_step.py      # column bands off / on (compare PAIRS only: with more
                                                                           # than two arms each one always follows the same predecessor)
    GELU_BITS=8,16 ROUNDS=8 python tools/ab_step.py        # instead: width of the gelu' the MLP block keeps (functional.GELU_GRAD_BITS)
    LIBS=a.so,b.so python tools/ab_step.py                 # instead: two BUILDS of the library take turns in the one process (variants
                                                           # 0 and 1; both are loaded, every op goes through the selected handle)
"""
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import torch  # noqa: E402

from myrtle_vision.hip.functional import cross_entropy  # noqa: E402
from myrtle_vision.hip.lib import lib  # noqa: E402
from myrtle_vision.models.vit import ViT  # noqa: E402
from myrtle_vision.utils.optim import AdamW, ParamArena  # noqa: E402
from myrtle_vision.utils.utils import seed_everything  # noqa: E402

import myrtle_vision.hip.functional as _F  # noqa: E402

GELU_AB = "GELU_BITS" in os.environ
LIB_AB = "LIBS" in os.environ
TN_AB = "TN_VARIANTS" in os.environ              # TN_VARIANTS=128,256: dW kernels forced
ATTN_AB = "ATTN_BWD_VARIANTS" in os.environ      # ATTN_BWD_VARIANTS=4,5: attention backward with four / two waves per workgroup
HANDLES = []
if LIB_AB:
    import ctypes
    import myrtle_vision.hip.lib as _L
    for path in os.environ["LIBS"].split(","):
        h = ctypes.CDLL(os.path.abspath(path))
        for name, (kinds, ret) in _L.SIGNATURES.items():
            fn = getattr(h, name, None)
            if fn is None:                          # an older build without a newer entry point: fine as long as the step does not call it
                continue
            fn.argtypes = [_L._KIND[k] for k in kinds]
            fn.restype = ret
        HANDLES.append(h)
    os.environ["NT_VARIANTS"] = ",".join(str(i) for i in range(len(HANDLES)))
VARIANTS = [int(v) for v in os.environ.get("GELU_BITS" if GELU_AB else "TN_VARIANTS" if TN_AB else "ATTN_BWD_VARIANTS" if ATTN_AB
                                           else "NT_VARIANTS", "3000,3002").split(",")]


def select(v):
    if LIB_AB:
        _L._lib = HANDLES[v]
    elif GELU_AB:
        _F.GELU_GRAD_BITS = v
    elif TN_AB:
        lib().mv_gemm_force_variant(0, v)
    elif ATTN_AB:
        lib().mv_attention_bwd_force(v)
    else:
        lib().mv_gemm_force_variant(v, 0)


ROUNDS, STEPS, BATCH = int(os.environ.get("ROUNDS", 10)), int(os.environ.get("STEPS", 10)), int(os.environ.get("BATCH", 256))
dev = torch.device("cuda", 0)
seed_everything(1234)
vit = ViT(decoder="classification", image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12,
          mlp_dim=3072, precision="bf16", q_format="FP32").to(dev)
arena = ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names())
opt = AdamW(arena, lr=6.25e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)
g = torch.Generator().manual_seed(1234)
img = torch.randn(BATCH, 3, 224, 224, generator=g).to(dev)
labels = torch.randint(0, 1000, (BATCH,), generator=g).to(dev)
vit.train()


def step():
    opt.zero_grad()
    loss = cross_entropy(vit(img), labels)
    loss.backward()
    opt.step()


for _ in range(5):
    step()
times = {v: [] for v in VARIANTS}
for r in range(ROUNDS):
    for v in VARIANTS:
        select(v)
        step()                                    # one untimed step after the switch
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        s.record()
        for _ in range(STEPS):
            step()
        e.record()
        torch.cuda.synchronize()
        times[v].append(s.elapsed_time(e) / STEPS)
if not GELU_AB and not LIB_AB:
    lib().mv_gemm_force_variant(0, 0)
for v in VARIANTS:
    t = times[v]
    print(f"variant {v:6d}: mean {statistics.mean(t):7.3f} ms/step  std {statistics.pstdev(t):6.3f}  min {min(t):7.3f}  "
          f"({BATCH / statistics.mean(t) * 1e3:7.0f} img/s)  blocks {len(t)} x {STEPS} steps")
