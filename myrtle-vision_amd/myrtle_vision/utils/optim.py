"""Optimizer and LR schedule of the reference training loop on the HIP AdamW kernel.

The reference builds them with ``timm.optim.create_optimizer`` / ``timm.scheduler.create_scheduler``
(classification/train.py:161-166) from the namespace ``get_optimizer_args`` fills (utils/models.py:84-110).
timm==0.5.4 is not available offline and the reference has no tests for it: the semantics below are a
restatement (PARITY UNPINNED, see oracle/optim_oracle.py for the CPU restatement the tests compare with):

* ``create_optimizer`` with opt="adamw": ``torch.optim.AdamW`` over two groups -- parameters with
  ``ndim <= 1`` or a name ending in ``.bias`` get weight_decay 0, the rest ``weight_decay``; ViT defines no
  ``no_weight_decay()`` so pos_embedding / cls_token ARE decayed.
* ``create_scheduler`` with sched="cosine": ``CosineLRScheduler(t_initial=epochs, lr_min=min_lr,
  warmup_lr_init=warmup_lr, warmup_t=warmup_epochs, cycle_limit=1, t_in_epochs=True)``, stepped with the
  0-based epoch at epoch END (classification/train.py:287).

MI355X-native layout: all trainable, *used* parameters live in ONE flat fp32 arena per weight-decay group
(``ParamArena``); ``param.data`` and ``param.grad`` are views into it.  The optimizer step is then one kernel
launch per group over 86 M contiguous elements (HBM-bound: 7 x 4 B per element), ``zero_grad`` is one memset, and
the DDP gradient all-reduce (``utils/ddp.py``) works on contiguous slices of the same buffer with no packing.
"""
import math
from typing import Iterable, List, Tuple

import torch

from myrtle_vision.hip import ops


class ParamArena:
    """Flat fp32 storage for parameters and their gradients, split into (decay, no_decay) regions."""

    def __init__(self, named_params: Iterable[Tuple[str, torch.nn.Parameter]], skip=()):
        every = [(n, p) for n, p in named_params if p.requires_grad]
        named = [(n, p) for n, p in every if n not in set(skip)]
        if not named:
            raise ValueError("no trainable parameters")
        # parameter order of the torch.optim.AdamW the reference builds through timm (add_weight_decay: the no-decay group
        # first, then the decayed one, each in named_parameters order, the never-used detection parameters included):
        # the integer keys of the optimizer part of a reference checkpoint index into this list
        nd = lambda n, p: p.ndim <= 1 or n.endswith(".bias")
        self.torch_groups = [[n for n, p in every if nd(n, p)], [n for n, p in every if not nd(n, p)]]
        dev = named[0][1].device
        decay = [(n, p) for n, p in named if not (p.ndim <= 1 or n.endswith(".bias"))]
        no_decay = [(n, p) for n, p in named if (p.ndim <= 1 or n.endswith(".bias"))]
        self.names: List[str] = [n for n, _ in decay + no_decay]
        self.params: List[torch.nn.Parameter] = [p for _, p in decay + no_decay]
        # each tensor starts on a 32-byte boundary: kernels can use 16-byte vector access on any slice of the fp32 arena AND
        # of its bf16 staging copy (the half-width gradient exchange casts bucket slices, utils/ddp.py)
        offs, total = [], 0
        for _, p in decay + no_decay:
            offs.append(total)
            total += (p.numel() + 7) & ~7
        self.n_decay = offs[len(decay)] if no_decay and decay else (total if decay else 0)
        self.offsets, self.total = offs, total
        self.flat_param = torch.zeros(total, dtype=torch.float32, device=dev)
        self.flat_grad = torch.zeros(total, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for p, o in zip(self.params, offs):
                n = p.numel()
                self.flat_param[o:o + n].copy_(p.detach().reshape(-1).float())
                p.data = self.flat_param[o:o + n].view(p.shape)
                p.grad = None
                ops.register_grad_slot(p, self.flat_grad[o:o + n])      # backward kernels write d/dp here directly
        # any torch-side in-place write to the flat buffer (loading into the arena, an EMA, a custom collective) moves the
        # weight caches' epoch by itself; raw-pointer writes (the AdamW kernel) announce themselves through bump_versions()
        ops.register_arena(self.flat_param)

    def slot(self, j):
        o = self.offsets[j]
        return self.flat_grad[o:o + self.params[j].numel()]

    def zero_grad(self):
        """``set_to_none`` semantics: with no gradient attached, the backward kernels write each parameter's gradient
        straight into its arena slot and autograd adopts that view (``ops.grad_out``) -- no zero-fill, no adds."""
        for p in self.params:
            p.grad = None

    def sync_grad(self, j):
        """Make slot j hold parameter j's gradient: a no-op on the fast path (the gradient already IS the slot);
        copies a gradient autograd materialised elsewhere; zero-fills the slot of a parameter that received none."""
        p, slot = self.params[j], self.slot(j)
        g = p.grad
        if g is None:
            slot.zero_()
        elif g.data_ptr() != slot.data_ptr():
            slot.copy_(g.detach().reshape(-1))
            p.grad = slot.view(p.shape)

    def sync_grads(self):
        for j in range(len(self.params)):
            self.sync_grad(j)

    def bump_versions(self):
        """The AdamW kernel writes through raw pointers: tell autograd / the bf16 weight cache the data changed."""
        for p in self.params:
            torch.autograd.graph.increment_version(p)
        ops.invalidate_weight_caches(self.flat_param)


class AdamW:
    """torch.optim.AdamW semantics on the fused HIP kernel over a ``ParamArena`` (one launch per decay group)."""

    def __init__(self, arena: ParamArena, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.01):
        self.arena = arena
        self.defaults = dict(lr=lr, betas=tuple(betas), eps=eps, weight_decay=weight_decay)
        nd = arena.n_decay
        self.param_groups = [
            dict(lr=lr, initial_lr=lr, weight_decay=weight_decay, range=(0, nd), name="decay"),
            dict(lr=lr, initial_lr=lr, weight_decay=0.0, range=(nd, arena.total), name="no_decay"),
        ]
        self.exp_avg = torch.zeros_like(arena.flat_param)
        self.exp_avg_sq = torch.zeros_like(arena.flat_param)
        self.step_count = 0
        self.grad_scale = 1.0            # e.g. 1/world_size after a SUM all-reduce
        self.max_grad_norm = None        # train_config.clip_grad: clip_grad_norm_ folded into the step (train.py:265-270)
        self.last_grad_norm = None       # device tensor [total_norm, clip coefficient] of the last clipped step
        self._hyper = None               # use_device_scalars(): per-group fp32 [3] (lr, bias_corr1, bias_corr2) on the device

    # -- per-step scalars on the device (HIP-graph capture: utils/graph.py) ---------------------------------------------
    def use_device_scalars(self, ring: int = 32):
        """From now on the kernels read lr and the two bias corrections from device memory instead of launch arguments, so a
        captured ``step()`` stays valid while the schedule and the step count advance.  ``advance()`` (called by ``step()``
        itself outside a capture, by the graph's replayer otherwise) bumps the step count and sends the new values: pinned
        staging slots in a ring, each guarded by an event, so the host may run many steps ahead of the device."""
        dev = self.arena.flat_param.device
        self._hyper = {g["name"]: torch.zeros(3, dtype=torch.float32, device=dev) for g in self.param_groups}
        self._stage = [torch.zeros(len(self.param_groups), 3, dtype=torch.float32).pin_memory() for _ in range(ring)]
        self._stage_done = [None] * ring
        self._stage_i = 0

    def advance(self):
        """step_count += 1 and (lr, 1 - beta1^t, 1 - beta2^t) of every group -> the device, on the current stream."""
        self.step_count += 1
        b1, b2 = self.defaults["betas"]
        bc1, bc2 = 1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count
        i = self._stage_i
        self._stage_i = (i + 1) % len(self._stage)
        if self._stage_done[i] is not None:
            self._stage_done[i].synchronize()           # the copy that last read this slot has run (normally long ago)
        host = self._stage[i]
        for r, g in enumerate(self.param_groups):
            host[r, 0], host[r, 1], host[r, 2] = g["lr"], bc1, bc2
        for r, g in enumerate(self.param_groups):
            self._hyper[g["name"]].copy_(host[r], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self._stage_done[i] = ev

    def zero_grad(self, set_to_none: bool = False):
        self.arena.zero_grad()

    @torch.no_grad()
    def clip_accumulated(self):
        """``clip_grad_norm_`` on the running accumulated gradient BETWEEN the micro-batches of an accumulation window
        (classification/train.py:265-270 runs it after every backward): the arena becomes grad_scale * coefficient * itself, in
        place -- the pending 1/world of a SUM all-reduce is applied here, so the next micro-batch adds to the averaged, clipped
        gradient exactly as the reference's does.  No host synchronisation (norm and coefficient stay on the device)."""
        if self.max_grad_norm is None:
            return
        self.arena.sync_grads()
        a = self.arena
        self.last_grad_norm = ops.grad_norm_clip(a.flat_grad, self.max_grad_norm, self.grad_scale)
        a.flat_grad.mul_(self.last_grad_norm[1] * self.grad_scale)

    @torch.no_grad()
    def step(self):
        self.arena.sync_grads()
        if self._hyper is None:
            self.step_count += 1
        elif not torch.cuda.is_current_stream_capturing():
            self.advance()                # inside a capture the replayer advances (utils/graph.py)
        b1, b2 = self.defaults["betas"]
        a = self.arena
        coef = None
        if self.max_grad_norm is not None:
            # torch.nn.utils.clip_grad_norm_ over every parameter that has a gradient = the whole arena (alignment gaps
            # hold zeros): one norm reduction, and the coefficient rides into the AdamW kernel as a device scalar
            self.last_grad_norm = ops.grad_norm_clip(a.flat_grad, self.max_grad_norm, self.grad_scale)
            coef = self.last_grad_norm[1:2]
        for g in self.param_groups:
            lo, hi = g["range"]
            if hi > lo:
                ops.adamw_step(a.flat_param[lo:hi], a.flat_grad[lo:hi], self.exp_avg[lo:hi], self.exp_avg_sq[lo:hi],
                               lr=g["lr"], beta1=b1, beta2=b2, eps=self.defaults["eps"],
                               weight_decay=g["weight_decay"], step=self.step_count, grad_scale=self.grad_scale,
                               clip_coef=coef, hyper=None if self._hyper is None else self._hyper[g["name"]])
        a.bump_versions()

    # checkpoint format = torch.optim.AdamW.state_dict() of the optimizer the reference builds (classification/train.py:
    # 161-166, utils/models.py:113-126): ``state`` keyed by the parameter's index in timm's group order with a per-parameter
    # ``step``, ``param_groups`` = [no_decay, decay] carrying those indices.  ``param_names`` (index -> name) is an extra
    # key torch ignores.  Parameters that never received a gradient (the detection tokens) have no state, as in torch.
    def _torch_index(self):
        names = self.arena.torch_groups[0] + self.arena.torch_groups[1]
        return {n: i for i, n in enumerate(names)}, names

    def state_dict(self):
        a = self.arena
        index, names = self._torch_index()
        state = {}
        if self.step_count > 0:
            for n, p, o in zip(a.names, a.params, a.offsets):
                k = p.numel()
                state[index[n]] = {"step": self.step_count,
                                   "exp_avg": self.exp_avg[o:o + k].view(p.shape).clone(),
                                   "exp_avg_sq": self.exp_avg_sq[o:o + k].view(p.shape).clone()}
        by_name = {g["name"]: g for g in self.param_groups}
        groups, first = [], 0
        for gname, members in zip(("no_decay", "decay"), a.torch_groups):
            g = by_name[gname]
            groups.append({"lr": g["lr"], "betas": self.defaults["betas"], "eps": self.defaults["eps"],
                           "weight_decay": g["weight_decay"], "amsgrad": False, "maximize": False,
                           "initial_lr": g["initial_lr"], "params": list(range(first, first + len(members)))})
            first += len(members)
        return {"state": state, "param_groups": groups, "param_names": names}

    def load_state_dict(self, sd):
        a = self.arena
        index, names = self._torch_index()
        st = sd["state"]
        if "param_names" in sd and list(sd["param_names"]) != names:
            raise ValueError("optimizer checkpoint was written for a different parameter list")
        legacy = any(isinstance(k, str) for k in st)            # round-1 layout: keyed by name, one top-level step
        steps = set()
        for n, p, o in zip(a.names, a.params, a.offsets):
            ent = st.get(n) if legacy else st.get(index[n])
            if ent is None:
                continue
            k = p.numel()
            if tuple(ent["exp_avg"].shape) != tuple(p.shape):
                raise ValueError(f"optimizer state of {n}: shape {tuple(ent['exp_avg'].shape)} != {tuple(p.shape)}")
            self.exp_avg[o:o + k].copy_(ent["exp_avg"].reshape(-1))
            self.exp_avg_sq[o:o + k].copy_(ent["exp_avg_sq"].reshape(-1))
            if "step" in ent:
                steps.add(int(ent["step"]))
        if legacy:
            self.step_count = int(sd["step"])
        else:
            if len(steps) > 1:
                raise ValueError(f"per-parameter steps differ ({sorted(steps)}): one fused step counter cannot hold them")
            self.step_count = steps.pop() if steps else 0
        by_name = {g["name"]: g for g in self.param_groups}
        order = ("decay", "no_decay") if legacy else ("no_decay", "decay")
        for gname, s in zip(order, sd["param_groups"]):
            by_name[gname].update({k: v for k, v in s.items() if k in ("lr", "initial_lr", "weight_decay")})


class CosineLRScheduler:
    """timm 0.5.4 ``CosineLRScheduler`` restated for cycle_limit=1, t_in_epochs=True, no noise, warmup_prefix False."""

    def __init__(self, optimizer, t_initial, lr_min=0.0, warmup_t=0, warmup_lr_init=0.0):
        self.optimizer = optimizer
        self.t_initial, self.lr_min, self.warmup_t, self.warmup_lr_init = t_initial, lr_min, warmup_t, warmup_lr_init
        self.base_values = [g["initial_lr"] for g in optimizer.param_groups]
        if warmup_t:
            self.warmup_steps = [(v - warmup_lr_init) / warmup_t for v in self.base_values]
            self._update([warmup_lr_init for _ in self.base_values])   # timm sets warmup_lr_init at construction
        else:
            self.warmup_steps = [1 for _ in self.base_values]

    def _get_lr(self, t):
        if t < self.warmup_t:
            return [self.warmup_lr_init + t * s for s in self.warmup_steps]
        if t < self.t_initial:
            return [self.lr_min + 0.5 * (v - self.lr_min) * (1 + math.cos(math.pi * t / self.t_initial))
                    for v in self.base_values]
        return [self.lr_min for _ in self.base_values]

    def _update(self, values):
        for g, v in zip(self.optimizer.param_groups, values):
            g["lr"] = v

    def step(self, epoch, metric=None):
        self._update(self._get_lr(epoch))

    def state_dict(self):
        return {k: v for k, v in self.__dict__.items() if k != "optimizer"}

    def load_state_dict(self, sd):
        self.__dict__.update(sd)


def create_optimizer(args, model, skip=None):
    """Counterpart of ``timm.optim.create_optimizer(args, model)`` for the options the reference configs use."""
    opt = str(args.opt).lower()
    if opt != "adamw":
        raise NotImplementedError(f"optimizer {args.opt!r}: only 'adamw' (every reference train_config) is implemented")
    if skip is None:
        skip = model.unused_parameter_names() if hasattr(model, "unused_parameter_names") else ()
    arena = ParamArena(model.named_parameters(), skip=skip)
    betas = tuple(args.opt_betas) if getattr(args, "opt_betas", None) else (0.9, 0.999)
    eps = args.opt_eps if getattr(args, "opt_eps", None) is not None else 1e-8
    return AdamW(arena, lr=args.lr, betas=betas, eps=eps, weight_decay=args.weight_decay)


def create_scheduler(args, optimizer):
    """Counterpart of ``timm.scheduler.create_scheduler(args, optimizer)`` -> (scheduler, num_epochs)."""
    if str(args.sched).lower() != "cosine":
        raise NotImplementedError(f"scheduler {args.sched!r}: only 'cosine' (every reference train_config) is implemented")
    sched = CosineLRScheduler(optimizer, t_initial=args.epochs, lr_min=args.min_lr, warmup_t=args.warmup_epochs,
                              warmup_lr_init=args.warmup_lr)
    return sched, args.epochs + (getattr(args, "cooldown_epochs", 0) or 0)
