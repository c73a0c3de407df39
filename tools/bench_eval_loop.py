#!/usr/bin/env python3
"""Wall time of engine.evaluate (classification/test.py, segmentation/test.py) on a synthetic tree: one short training
run to get a checkpoint, then the evaluation pass.   usage: bench_eval_loop.py [task=segmentation] [images=3000] [batch=256]"""
import copy, json, os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
task = sys.argv[1] if len(sys.argv) > 1 else "segmentation"
count = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
batch = int(sys.argv[3]) if len(sys.argv) > 3 else 256
from myrtle_vision.datasets.synthetic import make_dlrsd, make_resisc45
from myrtle_vision.engine import evaluate, train_worker

tmp = tempfile.mkdtemp(prefix="mv_eval_")
cfg = json.load(open(os.path.join(ROOT, task, "train_configs", "vit_base.json" if task == "classification" else "seg_base.json")))
data = json.load(open(os.path.join(ROOT, task, "data_configs", "data_config.json")))
if task == "classification":
    data["dataset_path"] = make_resisc45(os.path.join(tmp, "NWPU-RESISC45"), classes=45, per_class=max(count // 45, 1))
else:
    data["dataset_path"] = make_dlrsd(os.path.join(tmp, "DLRSD_dataset"), count=count)
json.dump(data, open(os.path.join(tmp, "data_config.json"), "w"))
cfg["data_config_path"] = os.path.join(tmp, "data_config.json")
cfg["train_config"].update(output_directory=os.path.join(tmp, "ckpt"), epochs=1, local_batch_size=batch, global_batch_size=batch,
                           iters_per_checkpoint=1, iters_per_val=10 ** 9, distributed=False, pretrained_backbone=None)
train_worker(0, 1, copy.deepcopy(cfg), task)
ck = sorted(os.listdir(cfg["train_config"]["output_directory"]))[-1]
cfg["train_config"]["checkpoint_path"] = os.path.join(cfg["train_config"]["output_directory"], ck)
import torch
torch.cuda.synchronize()
t0 = time.perf_counter()
res = evaluate(copy.deepcopy(cfg), task)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n = sum(1 for _ in open(os.path.join(data["dataset_path"], data["test_files"])))
print(f"evaluate({task}): {n} test images in {dt:.2f} s = {n / dt:.0f} img/s (model build + checkpoint load included)  {res}")
