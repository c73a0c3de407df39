"""CPU restatement of the optimizer / LR-schedule semantics the reference obtains from timm==0.5.4
(classification/train.py:161-166,274-287; utils/models.py:84-110).  TEST INFRASTRUCTURE.

PARITY UNPINNED: timm is not installed here and the reference has no tests or vectors for it; this restates the
published timm 0.5.4 behaviour (``optim_factory.add_weight_decay`` + ``torch.optim.AdamW``;
``CosineLRScheduler._get_lr`` with t_in_epochs, cycle_limit=1, no noise, warmup_prefix False).
"""
import math

import torch


def timm_param_groups(named_params, weight_decay, skip_list=()):
    """timm.optim.optim_factory.add_weight_decay: 1-D tensors, '.bias' names and skip_list names get no decay."""
    decay, no_decay = [], []
    for name, p in named_params:
        if not p.requires_grad:
            continue
        (no_decay if (p.ndim <= 1 or name.endswith(".bias") or name in skip_list) else decay).append(p)
    return [{"params": no_decay, "weight_decay": 0.0}, {"params": decay, "weight_decay": weight_decay}]


def reference_adamw(named_params, lr, weight_decay, eps=1e-8, betas=(0.9, 0.999)):
    return torch.optim.AdamW(timm_param_groups(named_params, weight_decay), lr=lr, eps=eps, betas=betas)


def cosine_lr(t, *, base_lr, t_initial, lr_min, warmup_t, warmup_lr_init):
    """Learning rate timm's CosineLRScheduler sets by ``scheduler.step(t)`` (t = 0-based epoch)."""
    if t < warmup_t:
        return warmup_lr_init + t * (base_lr - warmup_lr_init) / warmup_t
    if t < t_initial:
        return lr_min + 0.5 * (base_lr - lr_min) * (1 + math.cos(math.pi * t / t_initial))
    return lr_min
