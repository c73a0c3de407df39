#!/usr/bin/env python3
"""End-to-end throughput of the reference-style training LOOP (engine.train_worker: DataLoader workers decoding JPEGs,
GPU image preparation, model step, per-iteration loss print) on a synthetic RESISC-45 tree -- what classification/train.py
delivers, next to bench.py's resident-batch number.   usage: bench_train_loop.py [per_class=24] [batch=256] [epochs=4] [task=classification|segmentation]"""
import builtins, copy, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
per_class = int(sys.argv[1]) if len(sys.argv) > 1 else 24
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 256
epochs = int(sys.argv[3]) if len(sys.argv) > 3 else 4
task = sys.argv[4] if len(sys.argv) > 4 else "classification"
from myrtle_vision.datasets.synthetic import make_dlrsd, make_resisc45
from myrtle_vision.engine import train_worker

tmp = tempfile.mkdtemp(prefix="mv_loop_")
cfg = json.load(open(os.path.join(ROOT, task, "train_configs", "vit_base.json" if task == "classification" else "seg_base.json")))
data = json.load(open(os.path.join(ROOT, task, "data_configs", "data_config.json")))
t0 = time.time()
if task == "classification":
    data["dataset_path"] = make_resisc45(os.path.join(tmp, "NWPU-RESISC45"), classes=45, per_class=per_class)
else:
    data["dataset_path"] = make_dlrsd(os.path.join(tmp, "DLRSD_dataset"), count=45 * per_class)
print(f"dataset: {45 * per_class} images in {time.time() - t0:.1f} s", flush=True)
json.dump(data, open(os.path.join(tmp, "data_config.json"), "w"))
cfg["data_config_path"] = os.path.join(tmp, "data_config.json")
cfg["train_config"].update(output_directory=os.path.join(tmp, "ckpt"), epochs=epochs, local_batch_size=batch, global_batch_size=batch,
                           iters_per_checkpoint=10 ** 9, iters_per_val=10 ** 9, distributed=False, pretrained_backbone=None)
stamps, real_print = [], builtins.print
def stamped(*a, **k):
    if a and isinstance(a[0], str) and a[0].startswith("Iteration"):
        stamps.append(time.perf_counter())
    real_print(*a, **k)
builtins.print = stamped
train_worker(0, 1, copy.deepcopy(cfg), task)
builtins.print = real_print
if len(stamps) > 4:
    d = [b - a for a, b in zip(stamps[2:-1], stamps[3:])]
    d.sort()
    med = d[len(d) // 2]
    print(f"iterations {len(stamps)}; median {med * 1e3:.1f} ms/iteration = {batch / med:.0f} img/s; "
          f"mean {sum(d) / len(d) * 1e3:.1f} ms (epoch boundaries included)")
