"""GPU: the reference's CPU-runnable plumbing case (BASELINE configs[0]) on the HIP path -- ViT-Tiny, synthetic
RESISC-45 layout, batch 8, one process; plus the segmentation loop and the quantised evaluation script path."""
import copy
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import ROOT  # noqa: E402


def _config(tmp_path, task, size="tiny"):
    from myrtle_vision.datasets.synthetic import make_dlrsd, make_resisc45
    name = {"classification": f"vit_{size}.json", "segmentation": f"seg_{size}.json"}[task]
    cfg = json.load(open(os.path.join(ROOT, task, "train_configs", name)))
    data = json.load(open(os.path.join(ROOT, task, "data_configs", "data_config.json")))
    if task == "classification":
        data["dataset_path"] = make_resisc45(str(tmp_path / "NWPU-RESISC45"), classes=45, per_class=2)
    else:
        data["dataset_path"] = make_dlrsd(str(tmp_path / "DLRSD_dataset"), count=48)
    dpath = str(tmp_path / "data_config.json")
    json.dump(data, open(dpath, "w"))
    cfg["data_config_path"] = dpath
    t = cfg["train_config"]
    t.update(output_directory=str(tmp_path / "ckpt"), epochs=1, local_batch_size=8, global_batch_size=8, iters_per_checkpoint=2,
             iters_per_val=2, distributed=False, pretrained_backbone=None)
    cfg["vit_config"]["depth"] = 2                                          # keep the test short
    return cfg


@pytest.mark.parametrize("task", ["classification", "segmentation"])
def test_train_loop_writes_reloadable_checkpoints(tmp_path, task, capsys):
    from myrtle_vision.engine import evaluate, train_worker
    cfg = _config(tmp_path, task)
    iters = train_worker(0, 1, copy.deepcopy(cfg), task)
    out = capsys.readouterr().out
    assert iters >= 2 and "Iteration 1:" in out and "nan" not in out.lower()
    ckpts = sorted(os.listdir(cfg["train_config"]["output_directory"]))
    assert "vit_000000" in ckpts and "vit_000002" in ckpts                  # rank-0 checkpoints, reference naming
    ck = torch.load(os.path.join(cfg["train_config"]["output_directory"], "vit_000002"), map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "lr_scheduler", "iteration"} and ck["iteration"] == 2
    # resume + evaluate from the checkpoint
    cfg2 = copy.deepcopy(cfg)
    cfg2["train_config"]["checkpoint_path"] = os.path.join(cfg["train_config"]["output_directory"], "vit_000002")
    res = evaluate(cfg2, task)
    assert 0.0 <= res["accuracy"] <= 1.0


def test_quantized_evaluation_path(tmp_path):
    from myrtle_vision.engine import evaluate, train_worker
    cfg = _config(tmp_path, "classification")
    train_worker(0, 1, copy.deepcopy(cfg), "classification")
    cfg["train_config"]["checkpoint_path"] = os.path.join(cfg["train_config"]["output_directory"], "vit_000002")
    fp32 = evaluate(copy.deepcopy(cfg), "classification")
    cfg["vit_config"]["q_format"] = "FP16_32"
    q = evaluate(copy.deepcopy(cfg), "classification", quantize=True, calib_steps=1)
    assert abs(q["accuracy"] - fp32["accuracy"]) <= 0.2                     # fp16 weights barely move predictions
