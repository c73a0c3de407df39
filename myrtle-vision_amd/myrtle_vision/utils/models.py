"""Model factory, optimizer-argument adaptor and checkpoint I/O -- counterpart of the reference
``src/myrtle_vision/utils/models.py`` with the same signatures and the same checkpoint wire format
(``torch.save({"model", "optimizer", "lr_scheduler", "iteration"})``, utils/models.py:113-126).

Out of scope here: DeiT distillation (``get_teacher`` / ``DistillWrapper``; stale in the reference).
``rename_timm_state_dict`` cannot download timm weights offline; it converts a LOCAL timm-format state dict
(a ``.pth``/``.pt`` path or an in-memory dict) with the reference's renaming rules (utils/models.py:157-223).
"""
import argparse
import os
import re

import torch

from myrtle_vision.models.vit import ViT
from myrtle_vision.utils.quantize import QFormat
from myrtle_vision.utils.utils import parse_config


def get_models(config, profile=False):
    """reference utils/models.py:25-60 -> (vit, distiller=None).  Optional extension key ``vit_config["precision"]``."""
    vit_config = config["vit_config"]
    data_config = parse_config(config["data_config_path"])
    if "distiller_config" in config:
        raise NotImplementedError("DeiT distillation is outside the MI355X hot-path scope (and stale in the reference)")
    vit_kwargs = {
        "decoder": vit_config["decoder"],
        "image_size": vit_config["image_size"],
        "patch_size": vit_config["patch_size"],
        "num_classes": data_config["number_of_classes"],
        "dim": vit_config["embed_dim"],
        "depth": vit_config["depth"],
        "heads": vit_config["heads"],
        "mlp_dim": vit_config["mlp_dim"],
        "dropout": vit_config["dropout"],
        "emb_dropout": vit_config["emb_dropout"],
        "profile": profile,
        "q_format": QFormat[vit_config["q_format"]],
        "precision": vit_config.get("precision"),
    }
    return ViT(**vit_kwargs), None


def prepare_model_and_load_ckpt(train_config, model, optimizer=None, lr_scheduler=None):
    """reference utils/models.py:63-81: resume when ``checkpoint_path`` is non-empty, else iteration 0."""
    if train_config["checkpoint_path"] != "":
        return load_checkpoint(model=model, optimizer=optimizer, lr_scheduler=lr_scheduler,
                               filepath=train_config["checkpoint_path"])
    return 0


def get_optimizer_args(train_config):
    """reference utils/models.py:84-110 (returns a namespace INSTANCE; the reference mutates the class)."""
    a = argparse.Namespace()
    a.opt = train_config["optimizer"]
    a.opt_eps = train_config["opt_eps"]
    a.opt_betas = train_config["opt_betas"]
    a.clip_grad = train_config["clip_grad"]
    a.momentum = train_config["momentum"]
    a.weight_decay = train_config["weight_decay"]
    a.sched = train_config["scheduler"]
    a.lr = train_config["lr"]
    a.lr_noise = train_config.get("lr_noise")
    a.lr_noise_pct = train_config.get("lr_noise_pct")
    a.lr_noise_std = train_config.get("lr_noise_std")
    a.warmup_lr = train_config["warmup_lr"]
    a.min_lr = train_config["min_lr"]
    a.epochs = train_config["epochs"]
    a.decay_epochs = train_config["decay_epochs"]
    a.warmup_epochs = train_config["warmup_epochs"]
    a.cooldown_epochs = train_config["cooldown_epochs"]
    a.patience_epochs = train_config["patience_epochs"]
    a.decay_rate = train_config["decay_rate"]
    return a


def save_checkpoint(model, optimizer, lr_scheduler, iteration, filepath):
    ckpt = {
        "model": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
        "optimizer": optimizer.state_dict(),
        "lr_scheduler": lr_scheduler.state_dict(),
        "iteration": iteration,
    }
    torch.save(ckpt, filepath)


def load_checkpoint(model, optimizer, lr_scheduler, filepath):
    checkpoint = torch.load(filepath, map_location="cpu", weights_only=False)
    model.load_state_dict(checkpoint["model"])
    if optimizer is not None:
        optimizer.load_state_dict(checkpoint["optimizer"])
    if lr_scheduler is not None:
        lr_scheduler.load_state_dict(checkpoint["lr_scheduler"])
    return checkpoint["iteration"]


def apply_rules(name, rules):
    """Apply the first matching rule (regex substitution) to name (reference utils/models.py:144-151)."""
    for pattern, replacement in rules:
        if re.match(pattern, name) is not None:
            return re.sub(pattern, replacement, name)
    return name


_BLOCK = r"blocks\.([0-9]+)\."
_TIMM_RULES = [
    (r"pos_embed", r"pos_embedding"),
    (r"patch_embed\.proj\.(weight|bias)", r"patch_to_embedding.\1"),
    (_BLOCK + r"norm1\.(weight|bias)", r"transformer.layers.\1.0.fn.norm.\2"),
    (_BLOCK + r"attn\.qkv\.(weight|bias)", r"transformer.layers.\1.0.fn.fn.to_qkv.\2"),
    (_BLOCK + r"attn\.proj\.(weight|bias)", r"transformer.layers.\1.0.fn.fn.to_out.0.\2"),
    (_BLOCK + r"norm2\.(weight|bias)", r"transformer.layers.\1.1.fn.norm.\2"),
    (_BLOCK + r"mlp\.fc1\.(weight|bias)", r"transformer.layers.\1.1.fn.fn.net.0.\2"),
    (_BLOCK + r"mlp\.fc2\.(weight|bias)", r"transformer.layers.\1.1.fn.fn.net.3.\2"),
]
_TIMM_HEAD = [r"norm\.weight", r"norm\.bias", r"head\.weight", r"head\.bias"]


def rename_timm_state_dict(timm_model_name, vit_config, num_classes):
    """timm ViT state dict -> this package's names (reference utils/models.py:154-223).

    ``timm_model_name`` is a local file path or an already loaded dict (the reference passes a model NAME and
    downloads it; there is no network here).  The classifier head (norm/head) is dropped and the conv patch
    embedding (O, I, H, W) is permuted to the Linear layout (O, (H, W, I))."""
    if isinstance(timm_model_name, dict):
        src = timm_model_name
    elif isinstance(timm_model_name, str) and os.path.exists(timm_model_name):
        src = torch.load(timm_model_name, map_location="cpu", weights_only=False)
        src = src.get("model", src.get("state_dict", src)) if isinstance(src, dict) else src
    else:
        raise FileNotFoundError(
            f"pretrained_backbone={timm_model_name!r}: timm checkpoints cannot be downloaded here; pass a path to a "
            "local timm-format state dict (or leave pretrained_backbone unset to train from random init)")
    out = {}
    for key, value in src.items():
        if any(re.match(p, key) for p in _TIMM_HEAD):
            continue
        new_key = apply_rules(key, _TIMM_RULES)
        if new_key == "patch_to_embedding.weight" and value.dim() == 4:
            patch_dim = vit_config["patch_size"] ** 2 * value.shape[1]
            value = value.permute(0, 2, 3, 1).reshape(vit_config["embed_dim"], patch_dim)
        out[new_key] = value
    return out
