#!/usr/bin/env python3
"""Where in the backward pass each gradient bucket's all-reduce is handed to RCCL (one GPU, MV_FORCE_DIST-style: a
one-rank process group, so the collective itself is trivial -- the point is WHEN it is enqueued).

An event is recorded on the compute stream at every bucket launch (the collective waits for exactly that point of the
stream, then runs on RCCL's own stream beside the rest of backward) and at the end of backward; the table gives, per bucket,
the GPU time of backward already executed and still to come when its all-reduce could start.

    python tools/ddp_overlap_timeline.py            # ViT-B/16, batch 256, the product's bucket size (BUCKET_MIB overrides)
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
dev = torch.device("cuda", 0)
torch.cuda.set_device(dev)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

from myrtle_vision.hip.functional import cross_entropy  # noqa: E402
from myrtle_vision.models.vit import ViT  # noqa: E402
from myrtle_vision.utils.ddp import GradAllReducer  # noqa: E402
from myrtle_vision.utils.optim import AdamW, ParamArena  # noqa: E402
from myrtle_vision.utils.utils import seed_everything  # noqa: E402

seed_everything(1234)
vit = ViT(decoder="classification", image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12, mlp_dim=3072,
          precision="bf16", q_format="FP32").to(dev)
arena = ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names())
opt = AdamW(arena, lr=6.25e-5, weight_decay=0.05)
red = (GradAllReducer(arena, bucket_bytes=int(os.environ["BUCKET_MIB"]) << 20) if os.environ.get("BUCKET_MIB")
       else GradAllReducer(arena))                    # the product default (48 MiB)
red.enabled = True
marks = []
orig = red._launch


def launch(b):
    e = torch.cuda.Event(enable_timing=True)
    e.record()
    marks.append((b, e))
    orig(b)


red._launch = launch
g = torch.Generator().manual_seed(1)
img, labels = torch.randn(256, 3, 224, 224, generator=g).to(dev), torch.randint(0, 1000, (256,), generator=g).to(dev)
for it in range(4):
    marks.clear()
    opt.zero_grad()
    loss = cross_entropy(vit(img), labels)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    loss.backward()
    t1.record()
    red.finish()
    opt.step()
torch.cuda.synchronize()
total = t0.elapsed_time(t1)
print(f"backward: {total:.2f} ms of GPU time; {len(red.ranges)} buckets of <= {os.environ.get('BUCKET_MIB', 48)} MiB "
      f"({arena.total * 4 / 2 ** 20:.0f} MiB of fp32 gradients)")
print("bucket  MiB   enqueued after (ms)   backward still to run (ms)   share of backward left")
for b, e in marks:
    lo, hi, _, _ = red.ranges[b]
    at = t0.elapsed_time(e)
    print(f"{b:5d} {(hi - lo) * 4 / 2 ** 20:5.0f} {at:14.2f} {total - at:22.2f} {100 * (total - at) / total:20.1f} %")
late = [b for b, e in marks if t0.elapsed_time(e) > total + 1e-3]
print("buckets launched from hooks during backward:", len(marks) - len(late), "of", len(red.ranges))
dist.destroy_process_group()
