"""Training-TRAJECTORY parity (rows a16 / f1 / f4 end to end): K optimizer steps of the HIP path -- forward, loss,
backward, fused AdamW over the flat arena, cosine/warm-up schedule stepped as the reference's loop steps it, with a
checkpoint written and re-loaded half way -- against the CPU oracle run the same way: ``oracle.vit_oracle.loss_and_grads``
+ ``oracle.optim_oracle.reference_adamw`` (torch.optim.AdamW over timm's two groups) + ``cosine_lr``.

Reference: classification/train.py:161-166 (optimizer, scheduler), :239-287 (loop: zero_grad, forward, loss, backward,
step; ``lr_scheduler.step(epoch)`` with the 0-based epoch at epoch END), utils/models.py:113-141 (checkpoint).
One optimizer step per "epoch" here so that the learning rate CHANGES between the six steps (warm-up over two epochs,
then the cosine): [warmup_lr, warmup_lr, mid-warm-up, cos(2), cos(3), cos(4)].

Statistic.  Adam divides by sqrt(v): where a gradient element is ~0 the update's SIGN is decided by rounding noise, so a
handful of the 1.5 M elements move by up to 2 lr per step on either side.  Parameters are therefore compared in relative
L2 per tensor (VERDICT: <= 1e-5 of the parameter), and the UPDATE p_final - p_initial -- the far stricter statistic --
relative to its own norm."""
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle.detinit import det_images, det_labels, det_state_dict  # noqa: E402
from oracle.optim_oracle import cosine_lr, reference_adamw  # noqa: E402
from oracle.vit_oracle import ViTConfig, loss_and_grads  # noqa: E402
from test_vit_parity import report  # noqa: E402

KW = dict(decoder="classification", image_size=224, patch_size=16, num_classes=45, dim=192, depth=2, heads=3, mlp_dim=768)
SCHED = dict(base_lr=1e-3, t_initial=8, lr_min=1e-5, warmup_t=2, warmup_lr_init=1e-4)
STEPS, BATCH, WD = 6, 4, 0.05


def _batch(i):
    return det_images(f"traj{i}", BATCH, 224), det_labels(f"traj{i}", (BATCH,), 45)


def _oracle_trajectory():
    cfg = ViTConfig(**KW)
    params = {k: torch.nn.Parameter(v.clone()) for k, v in det_state_dict(cfg.param_shapes()).items()}
    opt = reference_adamw(list(params.items()), lr=SCHED["base_lr"], weight_decay=WD)
    lr, losses, lrs = SCHED["warmup_lr_init"], [], []         # timm sets warmup_lr_init at construction
    for i in range(STEPS):
        img, labels = _batch(i)
        _, loss, grads = loss_and_grads({k: v.detach() for k, v in params.items()}, img, labels, cfg)
        for k, p in params.items():
            p.grad = grads[k]                                 # None for the two detection-only parameters: AdamW skips them
        for g in opt.param_groups:
            g["lr"] = lr
        opt.step()
        losses.append(float(loss))
        lrs.append(lr)
        lr = cosine_lr(i, **SCHED)                            # lr_scheduler.step(epoch) at epoch end, 0-based (train.py:287)
    return {k: v.detach() for k, v in params.items()}, losses, lrs


def _hip_objects(precision):
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.optim import AdamW, CosineLRScheduler, ParamArena
    vit = ViT(q_format="FP32", precision=precision, **KW)
    vit.load_state_dict(det_state_dict(ViTConfig(**KW).param_shapes()))
    vit = vit.cuda().train()
    arena = ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names())
    opt = AdamW(arena, lr=SCHED["base_lr"], betas=(0.9, 0.999), eps=1e-8, weight_decay=WD)
    sched = CosineLRScheduler(opt, t_initial=SCHED["t_initial"], lr_min=SCHED["lr_min"], warmup_t=SCHED["warmup_t"],
                              warmup_lr_init=SCHED["warmup_lr_init"])
    return vit, opt, sched


def _hip_steps(vit, opt, sched, first, last, losses, lrs):
    from myrtle_vision.hip.functional import cross_entropy
    for i in range(first, last):
        img, labels = _batch(i)
        opt.zero_grad()
        loss = cross_entropy(vit(img.cuda()), labels.cuda())
        loss.backward()
        lrs.append(opt.param_groups[0]["lr"])
        opt.step()
        sched.step(i)
        losses.append(float(loss))


def _hip_trajectory(precision, tmp_path, resume_at=None):
    from myrtle_vision.utils.models import load_checkpoint, save_checkpoint
    vit, opt, sched = _hip_objects(precision)
    losses, lrs = [], []
    if resume_at is None:
        _hip_steps(vit, opt, sched, 0, STEPS, losses, lrs)
    else:
        _hip_steps(vit, opt, sched, 0, resume_at, losses, lrs)
        path = os.path.join(str(tmp_path), f"vit_{resume_at:06d}")
        save_checkpoint(vit, opt, sched, resume_at, path)
        del vit, opt, sched
        vit, opt, sched = _hip_objects(precision)              # a fresh process would start exactly here
        with torch.no_grad():
            for p in vit.parameters():
                p.add_(1.0)                                    # whatever the fresh model held must not survive the load
        assert load_checkpoint(vit, opt, sched, path) == resume_at
        opt.arena.bump_versions()
        _hip_steps(vit, opt, sched, resume_at, STEPS, losses, lrs)
    torch.cuda.synchronize()
    return {k: v.detach().float().cpu() for k, v in vit.state_dict().items()}, losses, lrs


def _compare(tag, got, want, init):
    worst_p = worst_u = 0.0
    for k, w in want.items():
        if k in ("pos_embedding_det", "det_tokens"):
            assert torch.equal(got[k], init[k])               # never touched by either optimizer
            continue
        e_p = float((got[k] - w).norm() / w.norm())
        e_u = float((got[k] - w).norm() / (w - init[k]).norm())
        worst_p, worst_u = max(worst_p, e_p), max(worst_u, e_u)
    report(f"trajectory/{tag} parameters rel-L2", worst_p)
    report(f"trajectory/{tag} update rel-L2", worst_u)
    return worst_p, worst_u


def test_fp32_trajectory_matches_oracle_and_resume_is_bit_exact(tmp_path):
    torch.set_num_threads(8)
    want, want_losses, want_lrs = _oracle_trajectory()
    init = det_state_dict(ViTConfig(**KW).param_shapes())
    assert len(set(want_lrs)) >= 4                             # the schedule really changes along the way
    got, losses, lrs = _hip_trajectory("fp32", tmp_path)
    assert lrs == pytest.approx(want_lrs, rel=1e-12)
    for a, b in zip(losses, want_losses):
        assert abs(a - b) < 1e-4 * abs(b)
    worst_p, worst_u = _compare("fp32 vs oracle", got, want, init)
    assert worst_p < 1e-5
    assert worst_u < 2e-2                                      # see the module docstring: near-zero gradient elements under Adam
    # save -> reload at step 3 -> continue == uninterrupted, bit for bit (model, optimizer moments, step count, schedule)
    res, res_losses, res_lrs = _hip_trajectory("fp32", tmp_path, resume_at=3)
    assert res_lrs == lrs and res_losses == losses
    for k in got:
        assert torch.equal(res[k], got[k]), k


def test_bf16_trajectory_stays_within_its_drift_bound(tmp_path):
    """The benchmarked arithmetic over the same six steps: losses within 2e-2, parameters within 1e-3 (relative L2) of the
    fp32 oracle trajectory -- bf16 gradients are 1e-2 accurate, Adam turns that into a few percent of each update -- and
    its own resume is bit-exact."""
    torch.set_num_threads(8)
    want, want_losses, _ = _oracle_trajectory()
    init = det_state_dict(ViTConfig(**KW).param_shapes())
    got, losses, lrs = _hip_trajectory("bf16", tmp_path)
    for a, b in zip(losses, want_losses):
        assert abs(a - b) < 2e-2 * abs(b)
    worst_p, worst_u = _compare("bf16 vs oracle", got, want, init)
    assert worst_p < 1e-3 and worst_u < 0.25
    res, res_losses, _ = _hip_trajectory("bf16", tmp_path, resume_at=3)
    assert res_losses == losses
    for k in got:
        assert torch.equal(res[k], got[k]), k
