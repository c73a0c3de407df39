#!/usr/bin/env python3
"""Throughput of the GPU image preparation (mv_image_prepare) against the host Pillow pipeline it replaces, and the
PCIe bytes per image of both hand-over formats.  GPU box: python tools/bench_image_prep.py"""
import os, sys, time, random
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
import numpy as np, torch
from PIL import Image
from myrtle_vision.datasets.device_transforms import DevicePlan
from myrtle_vision.datasets.transforms import build_transform

cfg = {"RandomResizedCrop": 224, "RandomHorizontalFlip": None, "Normalize": {"Mean": [0.5] * 3, "Std": [0.5] * 3}}
B = 256
rng = np.random.default_rng(0)
imgs = [Image.fromarray(rng.integers(0, 256, (256, 256, 3), dtype=np.uint8)) for _ in range(B)]
plan, cpu = DevicePlan(cfg), build_transform(cfg)
t0 = time.perf_counter(); host = [cpu(i)[0] for i in imgs]; t_host = time.perf_counter() - t0
t0 = time.perf_counter(); samples = [(plan(i), 0) for i in imgs]; t_plan = time.perf_counter() - t0
t0 = time.perf_counter(); packed, _ = DevicePlan.collate(samples); t_col = time.perf_counter() - t0
packed = {k: v.pin_memory() for k, v in packed.items()}
dev = torch.device("cuda")
plan.apply(packed, dev); torch.cuda.synchronize()
d = {k: v.to(dev) for k, v in packed.items()}
from myrtle_vision.hip import ops
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(20): out = ops.image_prepare(d["raw"], d["kh"], d["bh"], d["kv"], d["bv"], d["flip"], plan.mean, plan.std)
e.record(); torch.cuda.synchronize()
t_k = s.elapsed_time(e) / 20 * 1e-3
s.record()
for _ in range(10): plan.apply(packed, dev)
e.record(); torch.cuda.synchronize()
t_all = s.elapsed_time(e) / 10 * 1e-3
fp32 = torch.stack(host).pin_memory()
s.record()
for _ in range(10): fp32.to(dev, non_blocking=True)
e.record(); torch.cuda.synchronize()
t_fp = s.elapsed_time(e) / 10 * 1e-3
bytes_dev = sum(v.numel() * v.element_size() for v in packed.values())
print(f"host Pillow pipeline (1 core): {t_host / B * 1e3:.3f} ms/img = {B / t_host:.0f} img/s/core")
print(f"worker side of the device path: decode-free plan {t_plan / B * 1e3:.3f} ms/img, collate {t_col / B * 1e3:.3f} ms/img")
print(f"mv_image_prepare: {t_k * 1e6:.0f} us per batch of {B} = {B / t_k:.0f} img/s; H2D + kernel {t_all * 1e3:.2f} ms = {B / t_all:.0f} img/s")
print(f"PCIe per image: device path {bytes_dev / B / 1024:.0f} KiB (uint8 frame + tables), host path {fp32.numel() * 4 / B / 1024:.0f} KiB (fp32 tensor); fp32 H2D {t_fp * 1e3:.2f} ms per batch")
