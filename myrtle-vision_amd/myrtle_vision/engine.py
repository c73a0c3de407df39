"""Training / evaluation loops shared by ``classification/`` and ``segmentation/`` scripts.

Loop semantics follow the reference scripts (classification/train.py:55-313, segmentation/train.py, */test.py):
seeding, batch-size solver, one process per GPU, rank-0 checkpoints every ``iters_per_checkpoint`` iterations named
``vit_{iteration:06}``, rank-0 validation every ``iters_per_val``, gradient accumulation without dividing the loss,
``lr_scheduler.step(epoch)`` at epoch end.  Differences, all deliberate:

* compute runs on the HIP kernels (model, loss, optimizer); no CPU fallback;
* gradients are exchanged by ``utils.ddp.GradAllReducer`` over RCCL, and only on the LAST micro-batch of an
  accumulation window (the reference's DDP all-reduces every micro-batch; the result is identical) -- except when
  ``clip_grad`` is set with ``n_batch_accum > 1``: then every micro-batch is exchanged, because the reference clips the
  running accumulated (and already averaged) gradient after every backward, and that needs the averaged gradient each time;
* the two detection-only parameters are left out of the optimizer/all-reduce (the reference's DDP dies on them);
* ``GradScaler`` is dropped (a numerical no-op without autocast);
* ``clip_grad`` is the reference's ``clip_grad_norm_`` after EVERY micro-batch backward on the running accumulated gradient
  (classification/train.py:265-270): the intermediate clips rescale the gradient arena in place
  (``AdamW.clip_accumulated``), the last one rides into the AdamW kernel as a device scalar;
* ``pretrained_backbone`` must be a local timm-format state dict (no network); a bare timm model NAME that is not a
  file means "random init" with a warning.
"""
import os
import random
import time

import torch
import torch.distributed as dist
from torch.utils.data import DataLoader, Sampler

from myrtle_vision.hip.functional import CrossEntropyLoss
from myrtle_vision.hip import ops
from myrtle_vision.utils.ddp import GradAllReducer, broadcast_parameters, exchange_dtype_from_env
from myrtle_vision.utils.miou import MIoU
from myrtle_vision.utils.models import (get_models, get_optimizer_args, prepare_model_and_load_ckpt,
                                        rename_timm_state_dict, save_checkpoint)
from myrtle_vision.utils.optim import create_optimizer, create_scheduler
from myrtle_vision.utils.utils import (cleanup_distributed, get_batch_sizes, init_distributed, parse_config,
                                       seed_everything)


class ShardSampler(Sampler):
    """Index for index what ``torch.utils.data.DistributedSampler(dataset)`` yields (classification/train.py:116,200:
    default arguments, so shuffle, seed 0, no drop_last): ``randperm(n)`` from a generator seeded ``seed + epoch``, padded
    to a multiple of the world size by repeating the head of the list, rank r takes ``indices[r::world]``
    (``tests/test_host_cpu.py::test_shard_sampler_is_torch_distributed_sampler``)."""

    def __init__(self, n, rank, world, seed=0):
        self.n, self.rank, self.world, self.seed, self.epoch = n, rank, world, seed, 0
        self.num_samples = (n + world - 1) // world

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __iter__(self):
        g = torch.Generator().manual_seed(self.seed + self.epoch)
        idx = torch.randperm(self.n, generator=g).tolist()
        pad = self.num_samples * self.world - len(idx)
        if pad > 0:
            idx += (idx * -(-pad // len(idx)))[:pad]
        return iter(idx[self.rank::self.world])

    def __len__(self):
        return self.num_samples


def _datasets(task, data_config):
    if task == "classification":
        from myrtle_vision.datasets.resisc45 import Resisc45 as DS
        collate = None
    else:
        from myrtle_vision.datasets.dlrsd import Dlrsd as DS, collate_both as collate
    mk = lambda mode, files, ops_, plan=None: DS(mode=mode, dataset_path=data_config["dataset_path"],
                                                 imagepaths=data_config[files], label_map_path=data_config["label_map"],
                                                 transform_config=data_config[ops_], device_plan=plan)
    return mk, collate


def _device_plan(data_config, ops_, enabled=True):
    """DevicePlan for a ``transform_ops_*`` section, or None (host transforms) when disabled or the chain is not one the
    GPU path covers.  MYRTLE_VISION_DEVICE_TRANSFORMS=0 forces the host path."""
    if not enabled or os.environ.get("MYRTLE_VISION_DEVICE_TRANSFORMS", "1") == "0":
        return None
    from myrtle_vision.datasets.device_transforms import DevicePlan
    try:
        return DevicePlan(data_config[ops_])
    except ValueError:
        return None


def _workers():
    """The reference uses one worker that also resizes/normalises; with those on the GPU the workers only decode, and
    there may be as many as this process's CPU share allows (capped: a 1-GPU job owns 16 cores)."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 2
    return max(1, min(8, n - 2))


class BatchFeed:
    """DataLoader -> (images fp32 [B,3,H,W], labels) ON THE DEVICE, for either path: host transforms (tensors are copied
    over) or a DevicePlan (uint8 frames + tables are copied over and mv_image_prepare / mv_mask_prepare finish the
    job).  Same batches either way (tests/test_image_prep.py)."""

    def __init__(self, dataset, plan, task, device, collate, **loader_kw):
        from myrtle_vision.datasets.device_transforms import DevicePlan
        self.plan, self.task, self.device = plan, task, device
        if plan is not None:
            loader_kw.setdefault("num_workers", _workers())
            collate = DevicePlan.collate
        else:
            loader_kw.setdefault("num_workers", 1)                          # classification/train.py:117
        # workers outlive the epoch: respawning them (fork + imports) costs seconds per epoch boundary, as long as ~90
        # iterations of ViT-B at batch 256
        loader_kw.setdefault("persistent_workers", loader_kw["num_workers"] > 0)
        self.loader = DataLoader(dataset, collate_fn=collate, pin_memory=True, **loader_kw)

    def __len__(self):
        return len(self.loader)

    def _to_device(self, batch):
        if self.plan is None:
            imgs, labels = batch
            return imgs.to(self.device, non_blocking=True), labels.to(self.device, non_blocking=True)
        packed, labels = batch
        imgs, masks = self.plan.apply(packed, self.device, mask_add=-1 if self.task == "segmentation" else 0)
        return imgs, (masks if self.task == "segmentation" else labels.to(self.device, non_blocking=True))

    def __iter__(self):
        """One batch ahead on a side stream: the host-to-device copy (50 MB of uint8 frames for 256 images, ~2 ms of PCIe)
        and the image-preparation kernels of batch i+1 run under the model step of batch i instead of in front of it."""
        if torch.device(self.device).type != "cuda":
            for batch in self.loader:
                yield self._to_device(batch)
            return
        main, side = torch.cuda.current_stream(self.device), torch.cuda.Stream(self.device)

        def stage(batch):
            with torch.cuda.stream(side):
                out = self._to_device(batch)
                ready = torch.cuda.Event()
                ready.record(side)
            for t in out:
                if torch.is_tensor(t):
                    t.record_stream(main)            # allocated on the side stream, consumed on the main one
            return out, ready

        it = iter(self.loader)
        try:
            cur = stage(next(it))
        except StopIteration:
            return
        while cur is not None:
            try:
                nxt = stage(next(it))                # queued now, so that it overlaps the step the caller runs on ``cur``
            except StopIteration:
                nxt = None
            out, ready = cur
            main.wait_event(ready)
            yield out
            cur = nxt


class _Scalars:
    """The reference's segmentation loop logs validation accuracy / loss / mIoU to TensorBoard
    (segmentation/train.py:17,33,69-71: ``SummaryWriter("runs/")``).  ``torch.utils.tensorboard`` needs the
    ``tensorboard`` package; where it is missing the same three scalars go to ``runs/scalars.jsonl`` instead (one JSON
    object per call), so the log exists either way and nothing fails at import time."""

    def __init__(self, logdir="runs/"):
        self.writer, self.path = None, None
        try:
            from torch.utils.tensorboard import SummaryWriter
            self.writer = SummaryWriter(logdir)
        except Exception:                                       # ImportError, or tensorboard present but unusable
            os.makedirs(logdir, exist_ok=True)
            self.path = os.path.join(logdir, "scalars.jsonl")

    def add_scalar(self, tag, value, step):
        if self.writer is not None:
            self.writer.add_scalar(tag, value, step)
        else:
            import json
            with open(self.path, "a") as f:
                f.write(json.dumps({"tag": tag, "value": float(value), "step": int(step)}) + "\n")

    def close(self):
        if self.writer is not None:
            self.writer.close()


def _fused_seg_tail(task, criterion):
    """The segmentation loops may replace ``criterion(vit(x), y)`` + ``argmax`` by ``vit.segmentation_loss`` only when
    the criterion is the plain mean cross entropy the reference uses (segmentation/train.py:188)."""
    from myrtle_vision.hip.functional import CrossEntropyLoss
    return task == "segmentation" and type(criterion) is CrossEntropyLoss


@torch.no_grad()
def validation(val_loader, device, criterion, vit, task, num_classes):
    total_loss, total_acc, n = 0.0, 0.0, max(len(val_loader), 1)
    miou = MIoU(num_classes, device) if task == "segmentation" else None     # counted where the predictions are
    vit.eval()
    for imgs, labels in val_loader:                            # a BatchFeed: both already on the device
        if _fused_seg_tail(task, criterion):
            loss, acc, pred = vit.segmentation_loss(imgs, labels)        # no [B,C,H,W] logits
            total_loss += float(loss) / n
            total_acc += float(acc) / n
        else:
            outputs = vit(imgs)
            total_loss += float(criterion(outputs, labels)) / n
            pred = outputs.argmax(dim=1)
            total_acc += float((pred == labels).float().mean()) / n
        if miou is not None:
            miou.add_img(pred, labels)                         # three bincounts on the device; 17 numbers leave it
    vit.train()
    return total_loss, total_acc, (miou.get_miou() if miou is not None else None)


def train_worker(rank, num_gpus, config, task="classification"):
    train_config, dist_config, vit_config = config["train_config"], config["dist_config"], config["vit_config"]
    data_config = parse_config(config["data_config_path"])
    if not torch.cuda.is_available():
        raise RuntimeError("training needs an MI355X: the myrtle_vision HIP path has no CPU fallback")
    device = torch.device("cuda", rank)
    torch.cuda.set_device(device)
    seed_everything(train_config["seed"])
    world = max(num_gpus, 1)
    batch_size, n_batch_accum = get_batch_sizes(train_config["local_batch_size"], num_gpus,
                                                train_config["global_batch_size"], verbose=(rank == 0))
    train_config["local_batch_size"] = batch_size
    train_config["global_batch_size"] = batch_size * n_batch_accum * world
    train_config["n_batch_accum"] = n_batch_accum
    if num_gpus > 1:
        init_distributed(rank, num_gpus, **dist_config)
    out_dir = train_config["output_directory"]
    if rank == 0:
        os.makedirs(out_dir, exist_ok=True)
        print("output directory:", out_dir)

    mk, collate = _datasets(task, data_config)
    use_dev = train_config.get("device_transforms", True)
    plan_t, plan_v = _device_plan(data_config, "transform_ops_train", use_dev), _device_plan(data_config, "transform_ops_val", use_dev)
    trainset = mk("train", "train_files", "transform_ops_train", plan_t)
    valset = mk("eval", "valid_files", "transform_ops_val", plan_v)
    sampler = ShardSampler(len(trainset), rank, world) if num_gpus > 1 else None     # seed 0: the reference's default
    train_loader = BatchFeed(trainset, plan_t, task, device, collate, shuffle=(sampler is None), sampler=sampler,
                             batch_size=batch_size, drop_last=train_config["drop_last_batch"])
    val_loader = BatchFeed(valset, plan_v, task, device, collate, batch_size=batch_size,
                           drop_last=train_config["drop_last_batch"])

    vit, _ = get_models(config)
    backbone = train_config.get("pretrained_backbone")
    if backbone is not None:
        if isinstance(backbone, str) and os.path.exists(backbone):
            missing = vit.load_state_dict(rename_timm_state_dict(backbone, vit_config, data_config["number_of_classes"]),
                                          strict=False)
            assert missing.unexpected_keys == []
        elif rank == 0:
            print(f"WARNING: pretrained_backbone={backbone!r} is not a local file (no network): training from random init")
    vit = vit.to(device)

    optimizer_args = get_optimizer_args(train_config)
    optimizer = create_optimizer(optimizer_args, vit)
    lr_scheduler, _ = create_scheduler(optimizer_args, optimizer)
    criterion = CrossEntropyLoss()
    iteration = prepare_model_and_load_ckpt(train_config=train_config, model=vit, optimizer=optimizer,
                                            lr_scheduler=lr_scheduler)
    optimizer.arena.bump_versions()
    reducer = GradAllReducer(optimizer.arena, exchange_dtype=exchange_dtype_from_env())   # MV_DDP_EXCHANGE=bf16: half-width, opt-in
    broadcast_parameters(optimizer.arena)
    optimizer.grad_scale = reducer.grad_scale
    # classification/train.py:265-270 (clip_grad_norm_ after EVERY backward, on the running accumulated gradient): the norm is
    # taken over the flat (all-reduced) gradient arena; on the last micro-batch of a window the coefficient is applied inside the
    # AdamW kernel, on the earlier ones the arena is rescaled in place (which needs the averaged gradient: those micro-batches are
    # exchanged too, as the reference's DDP does on every backward)
    optimizer.max_grad_norm = optimizer_args.clip_grad
    clip_every = optimizer_args.clip_grad is not None and n_batch_accum > 1
    scalars = _Scalars(train_config.get("tensorboard_dir", "runs/")) if (task == "segmentation" and rank == 0) else None

    vit.train()
    epoch_offset = max(0, int(batch_size * world * iteration / max(len(trainset), 1)))
    if num_gpus > 1:
        dist.barrier()
    n_accum, last_val = 0, (0.0, 0.0, None)
    num_classes = data_config["number_of_classes"]
    for epoch in range(epoch_offset, train_config["epochs"]):
        epoch_loss = epoch_acc = 0.0
        if sampler is not None:
            sampler.set_epoch(epoch)
        for imgs, labels in train_loader:
            if iteration % train_config["iters_per_checkpoint"] == 0 and n_accum == 0 and rank == 0:
                save_checkpoint(model=vit, optimizer=optimizer, lr_scheduler=lr_scheduler, iteration=iteration,
                                filepath=f"{out_dir}/vit_{iteration:06}")
            if iteration % train_config["iters_per_val"] == 0 and n_accum == 0 and rank == 0:
                last_val = validation(val_loader, device, criterion, vit, task, num_classes)
                if scalars is not None:                          # segmentation/train.py:69-71
                    scalars.add_scalar("accuracy", last_val[1], iteration)
                    scalars.add_scalar("loss", last_val[0], iteration)
                    scalars.add_scalar("miou", last_val[2], iteration)
            if n_accum == 0:
                optimizer.zero_grad()
            reducer.enabled = reducer.world > 1 and (clip_every or n_accum == n_batch_accum - 1)
            if _fused_seg_tail(task, criterion):
                loss, acc_t, _ = vit.segmentation_loss(imgs, labels)
            else:
                outputs = vit(imgs)
                loss = criterion(outputs, labels)
                acc_t = None
            loss.backward()
            n_accum += 1
            if clip_every and n_accum < n_batch_accum:
                reducer.finish()
                optimizer.clip_accumulated()
            if n_accum == n_batch_accum:
                n_accum = 0
                reducer.finish()
                optimizer.step()
                iteration += 1
                if rank == 0:
                    acc = float(acc_t) if acc_t is not None else float((outputs.argmax(dim=1) == labels).float().mean())
                    epoch_loss += float(loss.detach()) / len(train_loader)
                    epoch_acc += acc / len(train_loader)
                    print(f"Iteration {iteration}:\tloss={float(loss.detach()):.4f}\tacc={acc:.4f}")
        lr_scheduler.step(epoch)
        if rank == 0:
            extra = f" - val_miou: {last_val[2]:.4f}" if last_val[2] is not None else ""
            print(f"Epoch : {epoch + 1} - loss : {epoch_loss:.4f} - acc: {epoch_acc:.4f} - "
                  f"val_loss : {last_val[0]:.4f} - val_acc: {last_val[1]:.4f}{extra}\n")
    if scalars is not None:
        scalars.close()
    if num_gpus > 1:
        cleanup_distributed()
    return iteration


def launch_training(config, task):
    """``__main__`` of the reference train scripts: timestamped output dir, one process per visible GPU."""
    from datetime import datetime
    import torch.multiprocessing as mp
    config["train_config"]["output_directory"] += datetime.now().strftime("_%m_%d_%Y_%H_%M_%S")
    num_gpus = torch.cuda.device_count()
    if config["train_config"]["distributed"]:
        if num_gpus <= 1:
            print(f"WARNING: tried to enable distributed training but only found {num_gpus} GPU(s)")
    elif num_gpus > 1:
        print("INFO: you have multiple GPUs available but did not enable distributed training")
        num_gpus = 1
    try:
        if num_gpus > 1:
            mp.spawn(train_worker, args=(num_gpus, config, task), nprocs=num_gpus, join=True)
        else:
            train_worker(0, num_gpus, config, task)
    except KeyboardInterrupt:
        print("Ctrl-c pressed; cleaning up and ending training early...")


@torch.no_grad()
def evaluate(config, task, quantize=False, calib_steps=0, quantized_ckpt=False):
    """classification/test.py, segmentation/test.py and classification/test_quantize.py in one function."""
    from myrtle_vision.utils.models import load_checkpoint
    from myrtle_vision.utils.quantize import QFormat
    train_config, vit_config = config["train_config"], config["vit_config"]
    data_config = parse_config(config["data_config_path"])
    vit_config["dropout"], vit_config["emb_dropout"] = 0.0, 0.0            # classification/test.py:47-48
    if train_config["checkpoint_path"] == "":
        raise ValueError("a checkpoint is required for evaluation (train_config.checkpoint_path)")
    q_format = vit_config["q_format"]
    if quantize and not quantized_ckpt:
        vit_config["q_format"] = "FP32"                                    # test_quantize.py:90-94: build FP32, load, then prepare
    vit, _ = get_models(config)
    vit = vit.to("cuda")
    load_checkpoint(model=vit, optimizer=None, lr_scheduler=None, filepath=train_config["checkpoint_path"])
    mk, collate = _datasets(task, data_config)
    dev = torch.device("cuda")
    plan = _device_plan(data_config, "transform_ops_val", train_config.get("device_transforms", True))
    if quantize:
        if not quantized_ckpt:
            vit.quantizer.prepare_qat(q_format)
        calib = BatchFeed(mk("eval", "valid_files", "transform_ops_val", plan), plan, task, dev, collate,
                          batch_size=train_config["local_batch_size"])
        vit.train()
        for step, (imgs, _) in enumerate(calib):                           # test_quantize.py:26-34
            if step >= calib_steps:
                break
            vit(imgs)
        vit.convert()
    testset = mk("eval" if task == "classification" else "test", "test_files", "transform_ops_val", plan)
    loader = BatchFeed(testset, plan, task, dev, collate, batch_size=train_config["local_batch_size"])
    vit.eval()
    # predictions, the hit count and the mIoU histograms stay on the device (the reference concatenates int64 predictions and
    # labels of the whole test set on the host: 206 MB per batch of 256 masks); the segmentation decoder goes through the
    # fused tail, which yields the arg-max without the [B, C, H, W] logits
    miou = MIoU(data_config["number_of_classes"], dev) if task == "segmentation" else None
    correct, total = torch.zeros((), dtype=torch.float64, device=dev), 0
    preds, gts = [], []                                    # classification only (one label per image): for the report
    with torch.no_grad():
        for imgs, labels in loader:
            if task == "segmentation" and hasattr(vit, "segmentation_loss"):
                pred = vit.segmentation_loss(imgs, labels)[2]
            else:
                pred = vit(imgs).argmax(dim=1)
            correct += (pred == labels).sum()
            total += labels.numel()
            if miou is not None:
                miou.add_img(pred, labels)
            else:
                preds.append(pred.reshape(-1).cpu())
                gts.append(labels.reshape(-1).cpu())
    acc = float(correct) / max(total, 1)
    if miou is None:
        preds, gts = torch.cat(preds), torch.cat(gts)
    result = {"accuracy": acc}
    if miou is not None:
        result["miou"] = miou.get_miou()
        print(f"pixel accuracy: {acc:.4f}  mIoU: {result['miou']:.4f}")
    else:
        try:
            from sklearn.metrics import classification_report
            print(classification_report(gts.numpy(), preds.numpy(), digits=4, zero_division=0))
        except ImportError:
            print(f"accuracy: {acc:.4f}")
    return result
