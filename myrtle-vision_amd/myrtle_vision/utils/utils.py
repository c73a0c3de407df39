"""Host-side helpers of the training scripts -- counterpart of the reference ``src/myrtle_vision/utils/utils.py``
(config parsing, seeding, batch-size solver, process-group setup, dataset list files).  Same names, argument meaning
and error behaviour; the DETR helpers of the reference (``all_gather``, ``reduce_dict``) belong to the detection
task and are out of scope.
"""
import json
import os
import random

import numpy as np
import torch
import torch.distributed as dist


# ---- dataset list files (reference utils/utils.py:11-67) ------------------------------------------------------
def load_imagepaths_and_segmaps(dataset_path, imagepaths):
    """Lines ``<image path>,<segmap path>`` -> [[image, segmap], ...]"""
    pairs = []
    with open(os.path.join(dataset_path, imagepaths), encoding="utf-8") as f:
        for line in f:
            fields = line.split(",")
            pairs.append([fields[0], fields[1].strip("\n")])
    return pairs


def load_imagepaths_and_labels(dataset_path, imagepaths):
    """Lines ``images/<label>/<file>`` -> [[path, label], ...] (label = second path component)."""
    with open(os.path.join(dataset_path, imagepaths), encoding="utf-8") as f:
        return [[line.strip(), line.split("/")[1]] for line in f]


def _label_map(dataset_path, label_map_path):
    with open(os.path.join(dataset_path, label_map_path), encoding="utf-8") as f:
        return json.load(f)


def get_label_number(dataset_path, label_map_path, text_label):
    return _label_map(dataset_path, label_map_path)[text_label]


def get_label_list(dataset_path, label_map_path):
    labelmap = _label_map(dataset_path, label_map_path)
    return sorted(labelmap, key=labelmap.get)


# ---- config / seeding (reference utils/utils.py:70-83) --------------------------------------------------------
def parse_config(config_path):
    with open(config_path) as f:
        return json.loads(f.read())


def seed_everything(seed):
    random.seed(seed)
    os.environ["PYTHONHASHSEED"] = str(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed(seed)
        torch.cuda.manual_seed_all(seed)


# ---- batch-size solver (reference utils/utils.py:86-125) ------------------------------------------------------
def get_batch_sizes(target_batch, num_gpus, global_batch, verbose=False):
    """Solve global_batch = local_batch * n_batch_accum * max(num_gpus, 1)  ->  (local_batch, n_batch_accum).

    Prefers ``target_batch`` per GPU; otherwise the largest local batch below it that divides the per-GPU share;
    raises ValueError when the global batch cannot be split over the GPUs."""
    per_minibatch = num_gpus * target_batch if num_gpus > 0 else target_batch
    if global_batch % per_minibatch == 0:
        return target_batch, global_batch // per_minibatch
    if num_gpus > 0 and global_batch % num_gpus == 0:
        per_gpu = global_batch // num_gpus
        local = target_batch - 1
        while per_gpu % local != 0:
            local -= 1
        if verbose:
            print(f"WARNING: Did not select preferred max local batch size {target_batch}; "
                  f"using a local batch size of {local} instead")
        return local, per_gpu // local
    raise ValueError(
        f"WARNING: Could not fulfill the desired global batch size of {global_batch} as it is not divisible by the "
        f"number of GPUs  available ({num_gpus})\nPlease update the global_batch_size parameter in your config file "
        "or change the number of GPUs available (e.g. with CUDA_VISIBLE_DEVICES)")


# ---- process group (reference utils/utils.py:128-147) ---------------------------------------------------------
def init_distributed(rank, num_gpus, dist_backend, dist_url, group_name=None):
    """One process per GPU; ``dist_backend: "nccl"`` in the reference configs IS RCCL under PyTorch-ROCm."""
    assert torch.cuda.is_available(), "Distributed mode requires CUDA."
    if rank == 0:
        print("Initializing Distributed")
    torch.cuda.set_device(rank)
    # the container hostname may not resolve: the reference's tcp://localhost:54321 is rewritten to 127.0.0.1
    dist_url = dist_url.replace("//localhost", "//127.0.0.1")
    dist.init_process_group(dist_backend, init_method=dist_url, world_size=num_gpus, rank=rank)


def cleanup_distributed():
    dist.destroy_process_group()


def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


@torch.no_grad()
def accuracy(output, target, topk=(1,)):
    """precision@k in percent (reference utils/utils.py:244-260)."""
    if target.numel() == 0:
        return [torch.zeros([], device=output.device)]
    maxk = max(topk)
    pred = output.topk(maxk, 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    return [correct[:k].reshape(-1).float().sum(0) * (100.0 / target.size(0)) for k in topk]
