"""Synthetic datasets in the reference's ON-DISK layouts (no network for the real ones; BASELINE configs[0]).

RESISC-45 layout (classification/README.md:45-91): ``<root>/images/<class>/<class>_NNN.jpg`` 256x256 RGB,
``label_map.json``, ``{train,val,test}_imagepaths.txt`` with lines ``images/<class>/<file>``.
DLRSD layout: ``<root>/images/<name>.jpg`` + ``<root>/segmaps/<name>.png`` (uint8 labels 1..17), list files with
``<image>,<segmap>`` lines.
"""
import json
import os

import numpy as np
from PIL import Image


def make_resisc45(root, classes=45, per_class=4, size=256, seed=0):
    rng = np.random.default_rng(seed)
    names = [f"class{c:02d}" for c in range(classes)]
    lines = []
    for c, name in enumerate(names):
        os.makedirs(os.path.join(root, "images", name), exist_ok=True)
        for k in range(per_class):
            rel = f"images/{name}/{name}_{k:03d}.jpg"
            Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8)).save(os.path.join(root, rel), quality=90)
            lines.append(rel)
    with open(os.path.join(root, "label_map.json"), "w") as f:
        json.dump({n: i for i, n in enumerate(names)}, f)
    rng.shuffle(lines)
    n = len(lines)
    splits = {"train": lines[: int(0.7 * n)], "val": lines[int(0.7 * n): int(0.8 * n)], "test": lines[int(0.8 * n):]}
    for k, v in splits.items():
        with open(os.path.join(root, f"{k}_imagepaths.txt"), "w") as f:
            f.write("\n".join(v) + "\n")
    return root


def make_dlrsd(root, count=16, classes=17, size=256, seed=0):
    rng = np.random.default_rng(seed)
    os.makedirs(os.path.join(root, "images"), exist_ok=True)
    os.makedirs(os.path.join(root, "segmaps"), exist_ok=True)
    lines = []
    for k in range(count):
        Image.fromarray(rng.integers(0, 256, (size, size, 3), dtype=np.uint8)).save(os.path.join(root, f"images/img{k:03d}.jpg"))
        blocks = rng.integers(1, classes + 1, (size // 32, size // 32), dtype=np.uint8)        # labels 1..17 in 32x32 blocks
        Image.fromarray(np.kron(blocks, np.ones((32, 32), dtype=np.uint8))).save(os.path.join(root, f"segmaps/img{k:03d}.png"))
        lines.append(f"images/img{k:03d}.jpg,segmaps/img{k:03d}.png")
    with open(os.path.join(root, "label_map.json"), "w") as f:
        json.dump({f"class{c:02d}": c for c in range(classes)}, f)
    n = len(lines)
    for name, part in (("train", lines[: n // 2]), ("val", lines[n // 2: 3 * n // 4]), ("test", lines[3 * n // 4:])):
        with open(os.path.join(root, f"{name}_imagepaths.txt"), "w") as f:
            f.write("\n".join(part) + "\n")
    return root
