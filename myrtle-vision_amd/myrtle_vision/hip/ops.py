"""Tensor-level wrappers over the C ABI (no autograd here; see ``functional.py``).

Every wrapper takes CUDA (ROCm) tensors owned by PyTorch, passes raw device pointers plus the current stream to
``libmyrtle_vision_hip.so`` and returns immediately (asynchronous).  PyTorch is used only for device memory and
streams.  CPU tensors are rejected: there is no CPU compute path in this package.

Precision: ``"bf16"`` = bf16 activations and weights on MFMA with fp32 accumulation and an fp32 residual stream
(the benchmark configuration); ``"fp32"`` = every contraction to fp32 accuracy (parity mode: bf16x6 Linear products, f32-MFMA
attention; MV_F32_GEMM=mfma puts every product on the bit-exact fmaf-chain kernels); ``"bf16x3"`` = the fp32 mode's data flow with
every nn.Linear product to 2^-16 relative instead of 2^-25 (two bf16 pieces per operand, three pairings: half the matrix-core work
of bf16x6; attention core, LayerNorm, GELU, softmax and the residual stream exactly as in ``"fp32"``) -- the cheapest arithmetic
inside BASELINE's 1e-3 end to end; ``"bf16x3h"`` = bf16x3 with the attention core on IEEE-half operands (the fused bf16 kernels
templated on the element type, fp32 sums / softmax / outputs): logits still 7x inside 1e-3 and every arg-max exact, gradients to
1.6e-3 (twelve layers of 2^-12 roundings of q and k under the exponential), 20 % faster than bf16x3.
"""
import ctypes
import os
import threading
import weakref

import torch

from . import lib as _l
from .lib import (EPI_DGELU, EPI_EMBED, EPI_GELU, EPI_GELU_GRAD, EPI_GELU_GRAD8, EPI_MUL, EPI_MUL8, EPI_NONE, EPI_RESIDUAL,
                  EPI_SPLIT_DGELU, EPI_SPLIT_GELU, MV_BF16, MV_F32, check,
                  lib)

__all__ = ["EPI_NONE", "EPI_GELU", "EPI_RESIDUAL", "EPI_DGELU", "EPI_EMBED", "EPI_GELU_GRAD", "EPI_MUL", "EPI_GELU_GRAD8", "EPI_MUL8",
           "EPI_SPLIT_DGELU", "EPI_SPLIT_GELU"]

_DT = {torch.float32: MV_F32, torch.bfloat16: MV_BF16}


def act_dtype(prec: str):
    if prec == "bf16":
        return torch.bfloat16
    if prec in ("fp32", "bf16x3", "bf16x3h"):
        return torch.float32
    raise ValueError(f"unknown precision {prec!r} (expected 'bf16', 'bf16x3', 'bf16x3h' or 'fp32')")


def require_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError(
                "myrtle_vision computes only through its HIP kernels on an MI355X: got a CPU tensor "
                "(there is no CPU fallback; move the model and inputs to 'cuda')")


def _s():
    return torch.cuda.current_stream().cuda_stream


def _p(t):
    return None if t is None else t.data_ptr()


def pad8(n: int) -> int:
    return (n + 7) & ~7


class KernelTimer:
    """Optional per-launch timing of the MFMA kernels with events recorded on the launch stream (the stream handed
    to the C ABI is torch's current stream).  Used by bench.py for the live roofline figure; off by default."""

    def __init__(self, sample_every: int = 1, sample_steps=None):
        """``sample_every`` = k: only the launches of every k-th step (``next_step()``) are bracketed by events -- the
        event packets between kernels cost 1.4 ms (4 %) of a ViT-B step in which every GEMM launch is timed (rocprofv3 trace: ~300
        event records of ~5 us of idle each).  ``sample_steps``: an explicit set of step indices instead."""
        self.records = {}          # kernel name -> list of (start_event, end_event, algorithmic_flops)
        self.by_shape = {}         # shape label -> the same tuples
        self.sample_every = max(int(sample_every), 1)
        self.sample_steps = None if sample_steps is None else set(int(s) for s in sample_steps)
        self.step = -1
        self.active = True

    def next_step(self):
        self.step += 1
        self.active = (self.step in self.sample_steps) if self.sample_steps is not None else self.step % self.sample_every == 0

    def begin(self):
        if not self.active:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def end(self, name, start, flops, shape=None):
        if start is None:
            return
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        self.records.setdefault(name, []).append((start, e, flops))
        if shape is not None:
            self.by_shape.setdefault(shape, []).append((start, e, flops))

    def shape_summary(self):
        """-> {shape label: dict(launches, avg_us, tflops)}: the same launches as ``summary`` split by product shape."""
        torch.cuda.synchronize()
        out = {}
        for shape, recs in self.by_shape.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            out[shape] = dict(launches=len(recs), avg_us=1e3 * ms / len(recs),
                              tflops=sum(f for _, _, f in recs) / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
        return out

    def summary(self):
        """-> {name: dict(launches, total_ms, avg_us, flops_per_launch, tflops)} (synchronises)."""
        torch.cuda.synchronize()
        out = {}
        for name, recs in self.records.items():
            ms = sum(s.elapsed_time(e) for s, e, _ in recs)
            fl = sum(f for _, _, f in recs)
            out[name] = dict(launches=len(recs), total_ms=ms, avg_us=1e3 * ms / len(recs), flops_per_launch=fl / len(recs),
                             tflops=fl / (ms * 1e-3) / 1e12 if ms > 0 else 0.0)
        return out


_timer = None


def set_kernel_timer(timer):
    global _timer
    _timer = timer


_workspaces = {}


def workspace(nbytes: int, device) -> torch.Tensor:
    """Grow-only scratch buffer per (device, stream): ops on one stream execute in order, so they can share it."""
    key = (device.index, _s())
    w = _workspaces.get(key)
    if w is None or w.numel() < nbytes:
        w = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _workspaces[key] = w
    return w


# ------------------------------------------------------------------------------------------------------------
# gradient destinations
# ------------------------------------------------------------------------------------------------------------
# ``optim.ParamArena`` keeps every parameter gradient in one flat buffer (the all-reduce / AdamW operand).  The backward
# kernels write a parameter's gradient STRAIGHT into its arena slot when the parameter has no gradient yet: autograd
# then adopts the returned view as ``param.grad`` (no ``grad += new`` launch, no zero-fill of 344 MB per step).  If the
# parameter already holds a gradient (micro-batch accumulation) a fresh tensor is returned and autograd adds as usual.
_grad_slots = {}


def register_grad_slot(param, slot):
    """``slot``: flat fp32 view of ``param.numel()`` elements that must receive d(loss)/d(param)."""
    key = id(param)
    _grad_slots[key] = slot
    weakref.finalize(param, _grad_slots.pop, key, None)


def grad_slot(param):
    return _grad_slots.get(id(param)) if param is not None else None


def grad_out(param, shape, device):
    """fp32 tensor of ``shape`` for the kernels to write the gradient of ``param`` into (``param`` may be None)."""
    slot = grad_slot(param)
    if slot is not None and param.grad is None:
        n = 1
        for d in shape:
            n *= d
        if n == slot.numel():
            return slot.view(shape)
    return torch.empty(shape, dtype=torch.float32, device=device)


# ------------------------------------------------------------------------------------------------------------
# weights prepared for the MFMA path
# ------------------------------------------------------------------------------------------------------------
class PreparedWeight:
    """bf16 copies of one nn.Linear weight [N_out, K_in]: ``w`` [N_out, pad8(K_in)] for the forward product and
    ``wt`` [K_in, pad8(N_out)] (transposed) for the input-gradient product; pads are zero."""

    __slots__ = ("w", "wt", "ldw", "ldt", "n_out", "k_in", "version", "ptr", "ref", "epoch", "owner")


class _WeightPrepItem(ctypes.Structure):          # mv_weight_prep_item (include/myrtle_vision_hip.h)
    _fields_ = [("w", ctypes.c_void_p), ("w_bf16", ctypes.c_void_p), ("wt_bf16", ctypes.c_void_p), ("ldw", ctypes.c_int),
                ("ldt", ctypes.c_int), ("R", ctypes.c_int), ("C", ctypes.c_int), ("tiles_x", ctypes.c_int),
                ("first_block", ctypes.c_int)]


# Cache epoch: the prepared copies of a weight are valid while (its _version, its owner's epoch) are unchanged.  A weight's
# OWNER is the registered parameter arena whose flat buffer holds it (None for a free-standing parameter).  An owner's epoch
# moves when (a) ``invalidate_weight_caches(flat)`` names it -- ParamArena.bump_versions() after a raw-pointer write (the AdamW
# kernel), a replayed HIP graph (utils.graph) -- (b) ITS flat buffer is written through torch (its ``_version`` counts in-place
# ops on the flat tensor: loading into the arena, an EMA, a custom collective), which a parameter's own version counter never
# sees, or (c) ``invalidate_weight_caches()`` without an argument (everything).  Epochs only ever grow: an arena that dies folds
# its last version into its generation, so a value can never repeat, and one model's optimizer step does not make another
# (frozen, EMA, evaluation) model's copies stale.  One integer compare per lookup; nothing is refreshed until a weight is used.
_generation = [0]


class _Owner:
    """One registered arena: weak reference to the flat tensor, its address range, a generation counter."""
    __slots__ = ("ref", "lo", "hi", "gen", "last")

    def __init__(self, flat):
        self.ref, self.lo, self.gen, self.last = weakref.ref(flat), flat.data_ptr(), 0, flat._version
        self.hi = self.lo + flat.numel() * flat.element_size()

    def epoch(self) -> int:
        t = self.ref()
        if t is not None:
            self.last = t._version
        return self.gen + self.last


_owners = []


def register_arena(flat: torch.Tensor):
    for o in _owners:
        if o.ref() is None:
            _prep_table.pop(id(o), None)
    _owners[:] = [o for o in _owners if o.ref() is not None]
    _owners.append(_Owner(flat))


def _owner_of(weight):
    ptr = weight.data_ptr()
    for o in reversed(_owners):                     # the newest registration wins if an address range was recycled
        if o.lo <= ptr < o.hi and o.ref() is not None:
            return o
    return None


def invalidate_weight_caches(flat=None):
    """Move the epoch of the arena ``flat`` (every prepared copy of ITS parameters becomes stale), or of everything."""
    if flat is None:
        _generation[0] += 1
        return
    ptr = flat.data_ptr()
    for o in _owners:
        if o.lo == ptr and o.ref() is flat:
            o.gen += 1
            return
    _generation[0] += 1                             # not registered: be safe


def cache_epoch(owner=None) -> int:
    return _generation[0] + (owner.epoch() if owner is not None else 0)


_prepared = {}   # id(parameter) -> PreparedWeight (identity-keyed: tensors define == elementwise); entries die with the tensor
_prep_table = {}  # per owner: id(owner) -> (key (ids, pointers), uint8 item table on the device, count, total_blocks)
_capture_keep = None   # utils.graph: while a step is being captured, every device table / weight copy the capture reads is
#                        appended here, and the GraphedTrainStep keeps the list for as long as its graph exists


def _refresh_stale(device, owner, only=None):
    """After an optimizer step EVERY prepared weight of that optimizer's arena is stale: refresh them all in one launch
    (mv_weight_prep_batch) instead of one ~7 us launch per nn.Linear at its first use.  Only weights of ``owner`` are
    touched (``only``: a single weight, for free-standing parameters inside a graph capture)."""
    stale = []
    for pw in list(_prepared.values()):                      # a finalizer may pop entries while we look
        w = pw.ref() if pw.ref is not None else None
        if w is None or pw.owner is not owner or pw.w.device != device or not w.is_contiguous():
            continue
        if only is not None and w is not only:
            continue
        if pw.version != w._version or pw.ptr != w.data_ptr() or pw.epoch != cache_epoch(owner):
            stale.append((pw, w))
    if not stale:
        return
    epoch = cache_epoch(owner)
    key = tuple((id(w), w.data_ptr(), pw.w.data_ptr()) for pw, w in stale)
    ent = _prep_table.get(id(owner))
    if ent is None or ent[0] != key:
        items = (_WeightPrepItem * len(stale))()
        first = 0
        for it, (pw, w) in zip(items, stale):
            it.w, it.w_bf16, it.wt_bf16 = w.data_ptr(), pw.w.data_ptr(), pw.wt.data_ptr()
            it.ldw, it.ldt, it.R, it.C = pw.ldw, pw.ldt, pw.n_out, pw.k_in
            it.tiles_x, it.first_block = (max(pw.k_in, pw.ldw) + 63) // 64, first
            first += it.tiles_x * ((max(pw.n_out, pw.ldt) + 63) // 64)
        table = torch.frombuffer(bytearray(bytes(items)), dtype=torch.uint8).to(device)
        ent = _prep_table[id(owner)] = (key, table, len(stale), first)
    _, table, count, total = ent
    if _capture_keep is not None:                            # the graph replays this launch: it owns what the launch reads
        _capture_keep.append((table, [(pw.w, pw.wt) for pw, _ in stale]))
    check(lib().mv_weight_prep_batch(_p(table), count, total, _s()), "weight_prep_batch", count=count)
    for pw, w in stale:
        pw.version, pw.ptr, pw.epoch = w._version, w.data_ptr(), epoch


def prepared_weight(weight: torch.Tensor) -> PreparedWeight:
    """Cached bf16 / transposed-bf16 copies, refreshed when the parameter changed (``_version``, storage, owner epoch)."""
    key = id(weight)
    pw = _prepared.get(key)
    if pw is not None and pw.version == weight._version and pw.ptr == weight.data_ptr() and pw.epoch == cache_epoch(pw.owner):
        return pw
    require_cuda(weight)
    n_out, k_in = weight.shape
    owner = _owner_of(weight)
    if pw is not None and pw.owner is not owner:
        pw.owner, pw.epoch = owner, -1                      # the parameter moved into (or out of) an arena
    if pw is not None and pw.n_out == n_out and pw.k_in == k_in and pw.w.device == weight.device and weight.is_contiguous():
        # this weight and every other stale one of the same arena (a free-standing weight inside a capture: itself only)
        _refresh_stale(weight.device, owner, only=weight if (owner is None and _capture_keep is not None) else None)
        if pw.version == weight._version and pw.ptr == weight.data_ptr() and pw.epoch == cache_epoch(owner):
            return pw
    if pw is None or pw.n_out != n_out or pw.k_in != k_in or pw.w.device != weight.device:
        pw = PreparedWeight()
        pw.n_out, pw.k_in, pw.owner = n_out, k_in, owner
        pw.ldw, pw.ldt = pad8(k_in), pad8(n_out)
        pw.w = torch.empty(n_out, pw.ldw, dtype=torch.bfloat16, device=weight.device)
        pw.wt = torch.empty(k_in, pw.ldt, dtype=torch.bfloat16, device=weight.device)
        pw.ref = weakref.ref(weight)
        if key not in _prepared:
            weakref.finalize(weight, _prepared.pop, key, None)
        _prepared[key] = pw
    wd = weight.detach()
    if not wd.is_contiguous():
        wd = wd.contiguous()
    if _capture_keep is not None:
        _capture_keep.append((wd, [(pw.w, pw.wt)]))
    check(lib().mv_weight_prep(_p(wd), _p(pw.w), pw.ldw, _p(pw.wt), pw.ldt, n_out, k_in, _s()), "weight_prep",
          R=n_out, C=k_in)
    pw.version, pw.ptr, pw.epoch = weight._version, weight.data_ptr(), cache_epoch(owner)
    return pw


# ------------------------------------------------------------------------------------------------------------
# LayerNorm
# ------------------------------------------------------------------------------------------------------------
def layernorm_fwd(x, ldx, rows, dim, gamma, beta, out_dtype, eps=1e-5):
    """x: fp32 tensor whose data_ptr is row 0; rows are ``ldx`` elements apart.  -> (y [rows, dim], mean, rstd)"""
    require_cuda(x, gamma, beta)
    y = torch.empty(rows, dim, dtype=out_dtype, device=x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().mv_layernorm_fwd(_p(x), ldx, _p(gamma), _p(beta), _p(y), _DT[out_dtype], _p(mean), _p(rstd), rows, dim,
                                 eps, _s()), "layernorm_fwd", rows=rows, dim=dim, ldx=ldx)
    return y, mean, rstd


def layernorm_fwd_split(x, ldx, rows, dim, gamma, beta, eps=1e-5):
    """LayerNorm whose output leaves as the bf16 pieces of the split-operand products (``split_ex(layernorm_fwd(...))`` without the
    fp32 tensor in between, bit-identical): -> (y_split [rows, nseg * dim], mean, rstd)."""
    require_cuda(x, gamma, beta)
    y = _split_buffer(rows, dim, x.device)
    mean = torch.empty(rows, dtype=torch.float32, device=x.device)
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device)
    check(lib().mv_layernorm_fwd_split(_p(x), ldx, _p(gamma), _p(beta), _p(y), current_segments(), _p(mean), _p(rstd), rows, dim,
                                       eps, _s()), "layernorm_fwd_split", rows=rows, dim=dim)
    return y, mean, rstd


def layernorm_bwd(dy, x, ldx, gamma, mean, rstd, dx_add, dx, lddx, rows, dim, dx16=None, dx_colsum=None, beta=None, dx_split=None):
    """Writes dx (fp32 rows ``lddx`` apart; ``dx_add`` added if given, same layout) -> (dgamma, dbeta).
    Optional fused by-products: ``dx16`` (bf16 [rows, dim] copy of dx) or ``dx_split`` (the bf16 pieces of dx, a ``_split_buffer`` of
    the current segment count) and ``dx_colsum`` (fp32 [dim] column sums).
    ``gamma`` / ``beta`` (the parameters) select the gradient destinations (``grad_out``)."""
    require_cuda(dy, x, dx)
    dgamma = grad_out(gamma, (dim,), x.device)
    dbeta = grad_out(beta, (dim,), x.device)
    nbytes = lib().mv_layernorm_bwd_workspace_bytes(rows, dim)
    ws = workspace(nbytes, x.device)
    if dx_split is not None:
        check(lib().mv_layernorm_bwd_split(_p(dy), _DT[dy.dtype], _p(x), ldx, _p(gamma), _p(mean), _p(rstd), _p(dx_add), _p(dx),
                                           lddx, _p(dgamma), _p(dbeta), 0, _p(ws), ws.numel(), rows, dim, _p(dx_split),
                                           current_segments(), _p(dx_colsum), _s()), "layernorm_bwd_split", rows=rows, dim=dim)
        return dgamma, dbeta
    check(lib().mv_layernorm_bwd(_p(dy), _DT[dy.dtype], _p(x), ldx, _p(gamma), _p(mean), _p(rstd), _p(dx_add), _p(dx),
                                 lddx, _p(dgamma), _p(dbeta), 0, _p(ws), ws.numel(), rows, dim, _p(dx16), _p(dx_colsum), _s()),
          "layernorm_bwd", rows=rows, dim=dim)
    return dgamma, dbeta


# ------------------------------------------------------------------------------------------------------------
# fp32 Linear products on the bf16 matrix cores ("bf16x6"): each fp32 operand is split into three bf16 pieces
# (mv_split3_bf16) and ONE bf16 MFMA product contracts over the six piece pairings that matter; fp32-accurate (2^-25
# relative per product, fp32 accumulation) at ~2x the f32-MFMA kernel's rate.  mv_gemm_f32 (bit-exact k-ordered fmaf
# chain) stays the path for everything the shape rules below exclude, and for MV_F32_GEMM=mfma.
# ------------------------------------------------------------------------------------------------------------
_f32_gemm_mode = os.environ.get("MV_F32_GEMM", "bf16x6")


def set_f32_gemm(mode: str) -> str:
    """'bf16x6' (default) or 'mfma' (every fp32 product on mv_gemm_f32).  Returns the previous mode."""
    global _f32_gemm_mode
    if mode not in ("bf16x6", "mfma"):
        raise ValueError(f"f32 gemm mode {mode!r}")
    prev, _f32_gemm_mode = _f32_gemm_mode, mode
    return prev


def pad32(n: int) -> int:
    return (n + 31) & ~31


# How many SEGMENTS a split operand has: 6 = bf16x6 (three pieces per operand, six pairings: fp32-accurate, ``precision="fp32"``),
# 3 = bf16x3 (two pieces, the pairings a0 b0 + a0 b1 + a1 b0: 2^-16 relative per product, half the matrix-core work,
# ``precision="bf16x3"``).  The count is a property of the running model, carried per THREAD (forward runs on the caller's thread,
# backward on autograd's workers): ``with segments(prec_segments(prec))`` in the forward of every autograd function that holds a
# Linear product, which records it for its backward (functional._scoped).
_seg_tls = threading.local()


# ``precision="bf16x3h"`` additionally runs the attention core on IEEE-half operands (mv_attention_fwd_f16 / _bwd_f16): the scope
# value 4 = three segments + half attention (``current_segments()`` still answers 3).
def prec_segments(prec: str) -> int:
    return 4 if prec == "bf16x3h" else 3 if prec == "bf16x3" else 6


def current_segments() -> int:
    n = getattr(_seg_tls, "n", 6)
    return 3 if n == 4 else n


def half_attention() -> bool:
    return getattr(_seg_tls, "n", 6) == 4


class segments:
    def __init__(self, n: int):
        if n not in (3, 4, 6):
            raise ValueError(f"segments {n}: 3 (bf16x3), 4 (bf16x3 + half attention) or 6 (bf16x6)")
        self.n = n

    def __enter__(self):
        self.prev = getattr(_seg_tls, "n", 6)
        _seg_tls.n = self.n
        return self

    def __exit__(self, *exc):
        _seg_tls.n = self.prev
        return False


def _split_buffer(rows, cols, device):
    """[rows, nseg * cols] bf16 whose STORAGE runs on to the next multiple of 32 rows, zero-filled: the dW product walks its
    contraction (the rows) in stages of 32, and zero rows add nothing -- any token count takes the bf16x6 / bf16x3 path."""
    full = torch.empty(pad32(rows), current_segments() * cols, dtype=torch.bfloat16, device=device)
    if full.shape[0] != rows:
        full[rows:].zero_()
    return full[:rows]


def split3(x, rows, cols, ldx, role, stack=False):
    """fp32 [rows, cols] (row stride ldx) -> its bf16 segments (six, or three in a ``segments(3)`` scope): [rows, nseg * cols]
    side by side, or stacked [nseg * rows, cols] (``stack``).  role 0 = left operand of the product, 1 = right operand."""
    require_cuda(x)
    nseg = current_segments()
    if stack:
        out = torch.empty(nseg * rows, cols, dtype=torch.bfloat16, device=x.device)
        ldo, seg = cols, rows * cols
    else:
        out = _split_buffer(rows, cols, x.device)
        ldo, seg = nseg * cols, cols
    fn = lib().mv_split3_bf16 if nseg == 6 else lib().mv_split2_bf16
    check(fn(_p(x), ldx, _p(out), ldo, seg, rows, cols, role, _s()), "split3_bf16" if nseg == 6 else "split2_bf16", rows=rows,
          cols=cols)
    return out


# Activation splits shared between the dX and dW products of one backward function: inside ``split_scope()`` the last
# two role-0 side-by-side splits are kept (with a reference to their source tensor) and reused when the same tensor and
# geometry come again.  Nothing is remembered outside a scope, and a scope must not write into a tensor it has already
# split (the kernels write through raw pointers: no version counter would notice).
_split_tls = threading.local()        # backward runs on autograd's per-device worker threads: one memo per thread


class split_scope:
    def __enter__(self):
        self.prev = getattr(_split_tls, "memo", None)
        _split_tls.memo = []
        return self

    def __exit__(self, *exc):
        _split_tls.memo = self.prev
        return False


def split_act(x, rows, cols, ldx, colsum_out=None):
    """Role-0 side-by-side split [rows, 6 * cols] of an activation / gradient, memoised inside a ``split_scope``.
    ``colsum_out`` (fp32 [cols]): also leave the column sums of x there (the bias gradient when x is a dY) -- from the split
    pass itself when the split is made here, by a separate pass when it comes from the memo."""
    memo = getattr(_split_tls, "memo", None)
    make = (lambda: split3(x, rows, cols, ldx, 0)) if colsum_out is None else \
           (lambda: split_ex(x, rows, cols, ldx=ldx, colsum_out=colsum_out))
    if memo is None:
        return make()
    key = (rows, cols, ldx, x.data_ptr(), current_segments())
    for ent in memo:
        if ent[1] == key:                                    # ent[0] keeps that storage alive: the address cannot be reused
            if colsum_out is not None:
                colsum(x, rows, cols, ldx, colsum_out)
            return ent[2]
    out = make()
    memo.append((x, key, out))
    if len(memo) > 2:
        memo.pop(0)
    return out


class _SplitWeight:
    __slots__ = ("version", "ptr", "fwd", "dx", "epoch")


_split_weights = {}


def split_weight(weight: torch.Tensor, which: str) -> torch.Tensor:
    """Cached right-operand splits of an nn.Linear weight [N, K]: 'fwd' -> [N, nseg K] (y = x W^T), 'dx' -> [K, nseg N] (the
    transposed weight, dx = dy W); refreshed when the parameter changed.  One cache entry per segment count."""
    key = id(weight)
    sw = _split_weights.get(key)
    epoch = cache_epoch(_owner_of(weight))
    if sw is None or sw.version != weight._version or sw.ptr != weight.data_ptr() or sw.epoch != epoch:
        if sw is None:
            weakref.finalize(weight, _split_weights.pop, key, None)
        sw = _split_weights[key] = _SplitWeight()
        sw.version, sw.ptr, sw.fwd, sw.dx, sw.epoch = weight._version, weight.data_ptr(), {}, {}, epoch
    n_out, k_in = weight.shape
    nseg = current_segments()
    have = sw.fwd if which == "fwd" else sw.dx
    if nseg in have:
        return have[nseg]
    if weight.is_contiguous() and n_out % 2 == 0 and k_in % 2 == 0:
        # one read of w gives both layouts (mv_weight_split); a weight that takes gradients will be asked for the other one in
        # this step's backward, so both are made now -- one launch per weight and optimizer step, no transposed fp32 copy
        both = weight.requires_grad and torch.is_grad_enabled()
        want_fwd, want_dx = which == "fwd" or both, which == "dx" or both
        fwd = _split_buffer(n_out, k_in, weight.device) if want_fwd and nseg not in sw.fwd else None
        dx = _split_buffer(k_in, n_out, weight.device) if want_dx and nseg not in sw.dx else None
        check(lib().mv_weight_split(_p(weight.detach()), _p(fwd), _p(dx), n_out, k_in, nseg, _s()), "weight_split", R=n_out, C=k_in)
        if fwd is not None:
            sw.fwd[nseg] = fwd
        if dx is not None:
            sw.dx[nseg] = dx
        return have[nseg]
    if which == "fwd":
        sw.fwd[nseg] = split3(weight.detach().contiguous(), n_out, k_in, k_in, 1)
    else:
        sw.dx[nseg] = split3(weight.detach().t().contiguous(), k_in, n_out, n_out, 1)
    return have[nseg]


# ------------------------------------------------------------------------------------------------------------
# Building block of a faster tolerance-meeting mode (round 4; NOT used by the model yet, profiles/r04_fp8_correction_study.txt):
# the bf16x3 product with its two correction terms on e4m3 operands and the 8-bit matrix instruction.
# ------------------------------------------------------------------------------------------------------------
def f8c_exponent(t: torch.Tensor) -> int:
    """The power of two that puts max |t| into [128, 256) (e4m3's largest finite value is 448).  Synchronises: tests / tools."""
    import math
    amax = float(t.detach().abs().max())
    return 7 - math.floor(math.log2(amax)) if amax > 0 else 0


def split_f8c(x, rows, cols, role, exp_hi, ldx=None):
    """fp32 [rows, cols] -> uint8 [rows, 4 * cols]: bf16 p0 | e4m3 segment 1 | e4m3 segment 2 (include/myrtle_vision_hip.h)."""
    require_cuda(x)
    out = torch.empty(rows, 4 * cols, dtype=torch.uint8, device=x.device)
    check(lib().mv_split_f8c(_p(x), cols if ldx is None else ldx, _p(out), 4 * cols, rows, cols, role, int(exp_hi), _s()),
          "split_f8c", rows=rows, cols=cols)
    return out


def gemm_nt_f8c(a8, b8, M, N, K, exp_a, exp_b, out, ldc, bias=None, epi=0, aux=None, ld_aux=0):
    """out[M, N] = A B^T from two ``split_f8c`` operands (roles 0 and 1, exponents exp_a / exp_b) (+ bias, + aux for EPI_RESIDUAL)."""
    require_cuda(a8, b8, out)
    check(lib().mv_gemm_nt_f8c(_p(a8), 4 * K, _p(b8), 4 * K, _p(out), ldc, _DT[out.dtype], M, N, K, -(int(exp_a) + int(exp_b) + 8),
                               _p(bias), epi, _p(aux), ld_aux, _s()), "gemm_nt_f8c", M=M, N=N, K=K)
    return out


def _x6_nt_ok(M, N, Kc):
    """[M, N] = A[M, Kc] B[N, Kc]^T through the 8-phase kernel with a 6 * Kc contraction."""
    return _f32_gemm_mode == "bf16x6" and M >= 128 and N % 16 == 0 and N >= 64 and Kc % 64 == 0


def _x6_tn_ok(M, N, K):
    """dW[N, K] = dY[M, N]^T X[M, K] through the segmented TN ring kernel with a 6 * M contraction."""
    return _f32_gemm_mode == "bf16x6" and M >= 128 and N % 8 == 0 and K % 8 == 0


def x6_block_ok(M, *dims):
    """Whether every Linear product of a transformer block with M token rows and these feature widths takes the bf16x6
    path (forward, dX and dW): the block functions then keep the SPLITS of their activations instead of the activations."""
    return _f32_gemm_mode == "bf16x6" and M >= 128 and all(d % 64 == 0 for d in dims)


def split_ex(x, rows, cols, *, ldx=None, op=0, h=None, ldh=None, colsum_out=None):
    """Role-0 side-by-side split [rows, 6 * cols] of v = x (op 0), gelu(x) (op 1) or x * gelu'(h) (op 2); ``colsum_out``
    (fp32 [cols]) receives the column sums of v."""
    require_cuda(x)
    out = _split_buffer(rows, cols, x.device)
    ws = workspace(lib().mv_split3_ex_workspace_bytes(rows, cols), x.device) if colsum_out is not None else None
    fn = lib().mv_split3_bf16_ex if current_segments() == 6 else lib().mv_split2_bf16_ex
    check(fn(_p(x), cols if ldx is None else ldx, _p(h), cols if ldh is None else ldh, op, _p(out), rows,
                                  cols, _p(colsum_out), _p(ws), ws.numel() if ws is not None else 0, _s()),
          "split3_bf16_ex", rows=rows, cols=cols, mode=op)
    return out


def nt_x6(a6, weight, which, M, out, *, bias=None, residual=None):
    """out[M, N] = a (given as its split a6) times the Linear weight: which = 'fwd' -> a W^T (+bias) (+residual),
    'dx' -> a W.  out / residual dense fp32 [M, N]."""
    n_out, k_in = weight.shape
    N, Kc = (n_out, k_in) if which == "fwd" else (k_in, n_out)
    _nt_x6(a6, split_weight(weight, which), out, N, M, N, Kc, bias, EPI_RESIDUAL if residual is not None else EPI_NONE,
           residual, N, tag=which)
    return out


def nt_split_ok(M, N, Kc):
    """Whether the split-output epilogues of the NT kernels take [M, N] = A[M, Kc] B[N, Kc]^T (whole 256 x 256 tiles only)."""
    return M % 256 == 0 and N % 256 == 0 and (current_segments() * Kc) % 128 == 0


def nt_x6_gelu_split(y6, weight, M, h, *, bias):
    """fc1 of the split-operand modes in ONE launch: h[M, N] (fp32 pre-activation, kept for the backward) = y W^T + b and the
    bf16 pieces of gelu(h) -> [M, nseg * N] (what fc2's forward and dW products read); no split pass over h."""
    N, Kc = weight.shape
    nseg = current_segments()
    a6 = _split_buffer(M, N, h.device)
    t0 = _timer.begin() if _timer is not None else None
    check(lib().mv_gemm_nt_bf16(_p(y6), nseg * Kc, _p(split_weight(weight, "fwd")), nseg * Kc, _p(a6), nseg * N, MV_BF16, M, N,
                                nseg * Kc, _p(bias), EPI_SPLIT_GELU, None, 0, nseg, _p(h), N, _s()),
          "gemm_nt_bf16(split gelu)", M=M, N=N, K=nseg * Kc)
    if t0 is not None:
        _timer.end(f"gemm_nt_bf16x{nseg}", t0, 2.0 * M * N * Kc, shape=f"fwd N{N} K{Kc} split-gelu")
    return a6


def nt_x6_dgelu_split(d6, weight, M, h, colsum_out):
    """fc2's input gradient of the split-operand modes in ONE launch: the bf16 pieces of (dY W2) * gelu'(h) -> [M, nseg * K] and its
    column sums (fc1's bias gradient) -> ``colsum_out``; no fp32 dh, no split pass."""
    N, K = weight.shape                                  # nn.Linear [out = N, in = K]: dX[M, K] = dY[M, N] W
    nseg = current_segments()
    dh6 = _split_buffer(M, K, h.device)
    part = torch.empty((M + 63) // 64, K, dtype=torch.float32, device=h.device)
    t0 = _timer.begin() if _timer is not None else None
    check(lib().mv_gemm_nt_bf16(_p(d6), nseg * N, _p(split_weight(weight, "dx")), nseg * N, _p(dh6), nseg * K, MV_BF16, M, K,
                                nseg * N, None, EPI_SPLIT_DGELU, _p(h), K, nseg, _p(part), K, _s()),
          "gemm_nt_bf16(split dgelu)", M=M, N=K, K=nseg * N)
    if t0 is not None:
        _timer.end(f"gemm_nt_bf16x{nseg}", t0, 2.0 * M * N * K, shape=f"dx N{K} K{N} split-dgelu")
    colsum(part, part.shape[0], K, K, colsum_out)
    return dh6


def tn_x6(dy6, x6, M, weight):
    """dW[N, K] = dY^T X from the two splits, into the weight's gradient slot."""
    N, K = weight.shape
    dw = grad_out(weight, (N, K), dy6.device)
    _tn_segments(dy6, x6, dw, M, N, K)
    return dw


def _tn_segments(dy6, x6, dw, M, N, K):
    nseg = current_segments()
    Mp = pad32(M)                                  # the splits' storage is zero-padded to whole stages (_split_buffer)
    ws = workspace(lib().mv_gemm_tn_workspace_bytes(N, K, nseg * Mp), dy6.device)
    t0 = _timer.begin() if _timer is not None else None
    fn = lib().mv_gemm_tn_bf16_x6 if nseg == 6 else lib().mv_gemm_tn_bf16_x3
    check(fn(_p(dy6), _p(x6), _p(dw), K, N, K, Mp, _p(ws), ws.numel(), _s()), f"gemm_tn_bf16_x{nseg}", M=N, N=K, rows=Mp)
    if t0 is not None:
        _timer.end(f"gemm_tn_bf16x{nseg}", t0, 2.0 * M * N * K, shape=f"dw N{N} K{K}")


def _x6_ksplits(M, N):
    """K-split count of a bf16x6 NT product: 3 (two of the six segments per slice) when that shortens the schedule on
    the 256 CUs by a quarter or more -- 150 tiles: one round of 6 segments -> two rounds of 2."""
    tiles = ((M + 255) // 256) * ((N + 255) // 256)
    r1, r3 = -(-tiles // 256), -(-(3 * tiles) // 256) / 3.0
    return 3 if r3 <= 0.75 * r1 else 1


def _nt_x6(a6, b6, out, ldc, M, N, Kc, bias, epi, aux=None, ld_aux=0, aux_i=0, tag="fwd"):
    nseg = current_segments()
    Kx = nseg * Kc
    t0 = _timer.begin() if _timer is not None else None
    S = _x6_ksplits(M, N) if epi in (EPI_NONE, EPI_RESIDUAL) and ldc == N and (aux is None or ld_aux == N) else 1
    if S > 1 and (Kx % (128 * S) != 0 or out.dtype != torch.float32):
        S = 1
    c_dt = MV_F32 if out.dtype == torch.float32 else _l.MV_F16       # half output: q / k / v of precision "bf16x3h" (EPI_NONE only)
    if S > 1:
        slabs = workspace(S * M * N * 4, out.device)
        check(lib().mv_gemm_nt_bf16_ksplit(_p(a6), Kx, _p(b6), Kx, _p(slabs), M, N, Kx, S, _p(bias), _s()),
              f"gemm_nt_bf16_ksplit(x{nseg})", M=M, N=N, K=Kx, S=S)
        check(lib().mv_sum_slabs_add(_p(slabs), M * N, S, _p(aux) if epi == EPI_RESIDUAL else None, _p(out), M * N, _s()),
              "sum_slabs_add", n=M * N)
        if t0 is not None:
            _timer.end(f"gemm_nt_bf16x{nseg}", t0, 2.0 * M * N * Kc, shape=f"{tag} N{N} K{Kc} epi{epi} S{S}")
        return
    check(lib().mv_gemm_nt_bf16(_p(a6), Kx, _p(b6), Kx, _p(out), ldc, c_dt, M, N, Kx, _p(bias), epi, _p(aux),
                                ld_aux, aux_i, None, 0, _s()), f"gemm_nt_bf16(x{nseg})", M=M, N=N, K=Kx, epi=epi)
    if t0 is not None:
        _timer.end(f"gemm_nt_bf16x{nseg}", t0, 2.0 * M * N * Kc, shape=f"{tag} N{N} K{Kc} epi{epi}")


# ------------------------------------------------------------------------------------------------------------
# contractions.  Logical shapes: x [M, K], W [N, K] (nn.Linear layout), y [M, N]
# ------------------------------------------------------------------------------------------------------------
def linear_fwd(x, M, K, weight, bias, out, ldc, *, lda=None, epi=EPI_NONE, aux=None, ld_aux=0, aux_i=0, out2=None,
               ld_out2=0):
    """out[M, N] (+epilogue) = x[M, K] @ weight[N, K]^T + bias.  Dispatches on x.dtype: bf16 -> the bf16 MFMA kernels;
    fp32 -> bf16x6 (fp32-accurate, on the bf16 MFMA) where the shape rules allow, else mv_gemm_f32 (f32 MFMA, fmaf-chain bits)."""
    N = weight.shape[0]
    lda = K if lda is None else lda
    if x.dtype == torch.bfloat16:
        pw = prepared_weight(weight)
        t0 = _timer.begin() if _timer is not None else None
        check(lib().mv_gemm_nt_bf16(_p(x), lda, _p(pw.w), pw.ldw, _p(out), ldc, _DT[out.dtype], M, N, K, _p(bias), epi,
                                    _p(aux), ld_aux, aux_i, _p(out2), ld_out2, _s()),
              "gemm_nt_bf16", M=M, N=N, K=K, epi=epi)
        if t0 is not None:
            _timer.end("gemm_nt_bf16", t0, 2.0 * M * N * K, shape=f"fwd N{N} K{K} epi{epi}")
    elif (_x6_nt_ok(M, N, K) and out.dtype == torch.float32 and weight.dtype == torch.float32
          and (epi in (EPI_NONE, EPI_RESIDUAL, EPI_EMBED) and out2 is None
               or epi == EPI_GELU and ldc == N and (out2 is None or ld_out2 == N))):
        a6, b6 = split_act(x, M, K, lda), split_weight(weight, "fwd")
        if epi == EPI_GELU:
            # pre-activation (out2 when the caller keeps it) then the exact erf GELU as its own pass
            pre = out if out2 is None else out2
            _nt_x6(a6, b6, pre, N, M, N, K, bias, EPI_NONE)
            check(lib().mv_gelu_fwd(_p(pre), _p(out), MV_F32, M * N, _s()), "gelu_fwd", n=M * N)
        else:
            _nt_x6(a6, b6, out, ldc, M, N, K, bias, epi, aux, ld_aux, aux_i)
    else:
        w = weight.detach()
        check(lib().mv_gemm_f32(_p(x), lda, 1, 0, 0, _p(w), 1, w.stride(0), 0, 0, _p(out), ldc, 1, 0, 0, M, N, K, 1, 1,
                                1.0, 0, _p(bias), epi, _p(aux), ld_aux, aux_i, _p(out2), ld_out2, _s()),
              "gemm_f32", M=M, N=N, K=K, epi=epi)
    return out


def linear_dx(dy, M, N, weight, out, ldc, *, ld_dy=None, epi=EPI_NONE, aux=None, ld_aux=0, colsum_partial=None):
    """out[M, K] = dy[M, N] @ weight[N, K]  (optionally * gelu'(aux)).  ``colsum_partial`` (bf16 path, EPI_DGELU only):
    fp32 [ceil(M/64), K] receiving per-64-row column sums of ``out`` (bias-gradient partials)."""
    K = weight.shape[1]
    ld_dy = N if ld_dy is None else ld_dy
    if dy.dtype == torch.bfloat16:
        pw = prepared_weight(weight)
        t0 = _timer.begin() if _timer is not None else None
        check(lib().mv_gemm_nt_bf16(_p(dy), ld_dy, _p(pw.wt), pw.ldt, _p(out), ldc, _DT[out.dtype], M, K, N, None, epi,
                                    _p(aux), ld_aux, 0, _p(colsum_partial), K if colsum_partial is not None else 0, _s()),
              "gemm_nt_bf16(dx)", M=M, N=K, K=N, epi=epi)
        if t0 is not None:
            _timer.end("gemm_nt_bf16", t0, 2.0 * M * N * K, shape=f"dx N{K} K{N} epi{epi}")
    elif (_x6_nt_ok(M, K, N) and out.dtype == torch.float32 and weight.dtype == torch.float32
          and (epi == EPI_NONE or epi == EPI_DGELU and ldc == K and ld_aux == K)):
        _nt_x6(split_act(dy, M, N, ld_dy), split_weight(weight, "dx"), out, ldc, M, K, N, None, EPI_NONE, tag="dx")
        if epi == EPI_DGELU:
            check(lib().mv_gelu_bwd(_p(aux), _p(out), _p(out), MV_F32, M * K, _s()), "gelu_bwd", n=M * K)
    else:
        w = weight.detach()
        check(lib().mv_gemm_f32(_p(dy), ld_dy, 1, 0, 0, _p(w), w.stride(0), 1, 0, 0, _p(out), ldc, 1, 0, 0, M, K, N, 1,
                                1, 1.0, 0, None, epi, _p(aux), ld_aux, 0, None, 0, _s()),
              "gemm_f32(dx)", M=M, N=K, K=N, epi=epi)
    return out


def linear_dw(dy, x, M, N, K, *, ld_dy=None, ldx=None, want_bias=True, weight=None, bias=None):
    """dW[N, K] = dy[M, N]^T @ x[M, K] (fp32), db[N] = column sums of dy.  ``weight`` / ``bias`` (the parameters, optional)
    select the gradient destinations (``grad_out``)."""
    ld_dy = N if ld_dy is None else ld_dy
    ldx = K if ldx is None else ldx
    dw = grad_out(weight, (N, K), x.device)
    db = grad_out(bias, (N,), x.device) if want_bias else None
    if dy.dtype == torch.bfloat16:
        nbytes = lib().mv_gemm_tn_workspace_bytes(N, K, M)
        ws = workspace(nbytes, x.device)
        t0 = _timer.begin() if _timer is not None else None
        check(lib().mv_gemm_tn_bf16(_p(dy), ld_dy, _p(x), ldx, _p(dw), K, N, K, M, 0, _p(db), _p(ws), ws.numel(), _s()),
              "gemm_tn_bf16", M=N, N=K, Kc=M)
        if t0 is not None:
            _timer.end("gemm_tn_bf16(+reduce+colsum)", t0, 2.0 * M * N * K, shape=f"dw N{N} K{K} bias{int(want_bias)}")
    elif _x6_tn_ok(M, N, K) and dy.dtype == torch.float32 and x.dtype == torch.float32:
        a6, b6 = split_act(dy, M, N, ld_dy, colsum_out=db if want_bias else None), split_act(x, M, K, ldx)
        _tn_segments(a6, b6, dw, M, N, K)
    else:
        S = f32_dw_splits(M, N, K)
        if S == 1:
            check(lib().mv_gemm_f32(_p(dy), 1, ld_dy, 0, 0, _p(x), ldx, 1, 0, 0, _p(dw), K, 1, 0, 0, N, K, M, 1, 1, 1.0, 0,
                                    None, EPI_NONE, None, 0, 0, None, 0, _s()), "gemm_f32(dw)", M=N, N=K, K=M)
        else:
            # the token dimension (50 432 rows at batch 256) split over the kernel's batch axis: S x as many workgroups for a
            # product with only 36-144 output tiles; slabs summed in a fixed order (deterministic)
            Mc = M // S
            slabs = workspace(S * N * K * 4, x.device).view(torch.float32)[:S * N * K]
            check(lib().mv_gemm_f32(_p(dy), 1, ld_dy, Mc * ld_dy, 0, _p(x), ldx, 1, Mc * ldx, 0, _p(slabs), K, 1, N * K, 0,
                                    N, K, Mc, S, 1, 1.0, 0, None, EPI_NONE, None, 0, 0, None, 0, _s()),
                  "gemm_f32(dw, split)", M=N, N=K, K=Mc, S=S)
            check(lib().mv_sum_slabs(_p(slabs), N * K, S, _p(dw), N * K, 0, _s()), "sum_slabs", S=S, n=N * K)
        if want_bias:
            colsum(dy, M, N, ld_dy, db)
    return dw, db


def f32_dw_splits(M, N, K):
    """Number of equal token-dimension splits of the fp32 dW product [N, K] = dY[M, N]^T X[M, K]: enough for >= 2 workgroups
    per CU, each split a multiple of 16 rows and at least 512 rows long."""
    tiles = ((N + 127) // 128) * ((K + 127) // 128)
    best = 1
    for s in (2, 4, 8, 16, 32):
        if M % (16 * s) == 0 and M // s >= 512 and tiles * best < 512 and (N * K) % 4 == 0:
            best = s
    return best


def colsum(x, rows, cols, ld, out):
    ws = workspace(256 * cols * 4 + 1024, x.device)
    check(lib().mv_colsum(_p(x), _DT[x.dtype], ld, _p(out), 0, rows, cols, _p(ws), ws.numel(), _s()), "colsum",
          rows=rows, cols=cols)
    return out


# ------------------------------------------------------------------------------------------------------------
# attention
# ------------------------------------------------------------------------------------------------------------
def attention_fused_supported(qkv_dtype, N, dim_head):
    return qkv_dtype == torch.bfloat16 and dim_head == 64 and N <= 320


def attention_fwd(qkv, B, N, H, scale):
    """qkv bf16 [B, N, 3*H*64] -> (out bf16 [B, N, H*64], lse fp32 [B, H, N])"""
    require_cuda(qkv)
    out = torch.empty(B, N, H * 64, dtype=torch.bfloat16, device=qkv.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    check(lib().mv_attention_fwd(_p(qkv), _p(out), _p(lse), B, N, H, scale, _s()), "attention_fwd", B=B, N=N, H=H)
    return out, lse


def attention_bwd(qkv, out, dout, lse, B, N, H, scale, colsum=None):
    """-> dqkv.  ``colsum`` (optional fp32 [B, 3*H*64]) receives per-image column sums of dqkv (bias-gradient partials)."""
    dqkv = torch.empty_like(qkv)
    check(lib().mv_attention_bwd(_p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), _p(colsum), B, N, H, scale, _s()),
          "attention_bwd", B=B, N=N, H=H)
    return dqkv


def attention_f32_fused_supported(qkv_dtype, N, dim_head):
    return qkv_dtype == torch.float32 and dim_head == 64 and N <= 272


def attention_f16_supported(qkv_dtype, N, dim_head):
    """The half-operand attention core of precision "bf16x3h" (the bf16 kernels instantiated on IEEE half): fp32 q/k/v, N <= 288."""
    return qkv_dtype == torch.float32 and dim_head == 64 and N <= 288 and half_attention()


def cast_f16(src):
    """fp32 -> IEEE half (stored in a torch.float16 tensor)."""
    require_cuda(src)
    out = torch.empty(src.shape, dtype=torch.float16, device=src.device)
    check(lib().mv_cast(_p(src), MV_F32, _p(out), _l.MV_F16, src.numel(), _s()), "cast(f16)", n=src.numel())
    return out


def attention_fwd_f16(qkv16, B, N, H, scale):
    """-> (out fp32 [B, N, H*64], lse fp32 [B, H, N]) from half q/k/v [B, N, 3, H, 64]."""
    out = torch.empty(B, N, H * 64, dtype=torch.float32, device=qkv16.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv16.device)
    check(lib().mv_attention_fwd_f16(_p(qkv16), _p(out), _p(lse), B, N, H, scale, _s()), "attention_fwd_f16", B=B, N=N, H=H)
    return out, lse


def attention_bwd_f16(qkv16, out, dout, lse, B, N, H, scale, split=False, colsum=None):
    """-> dqkv [B, N, 3*H*64]: fp32, or (``split``) its bf16 pieces [B * N, nseg * 3*H*64] for the to_qkv dW / dX products.  ``out``
    / ``dout`` fp32 [B, N, H*64]: dout is scaled into half's range per (image, head) by a power of two taken from its own largest
    magnitude (one pass that also leaves delta); the kernel divides the factor out of its outputs.  ``colsum``: fp32 [B, 3*H*64]
    receiving per-image column sums of dqkv (to_qkv's bias gradient after a sum over images)."""
    dev = qkv16.device
    dout16 = torch.empty(B, N, H * 64, dtype=torch.float16, device=dev)
    delta = torch.empty(B, H, N, dtype=torch.float32, device=dev)
    gscale = torch.empty(B * H, dtype=torch.float32, device=dev)
    check(lib().mv_attention_bwd_prep_f16(_p(dout), _p(out), _p(dout16), _p(delta), _p(gscale), B, N, H, _s()),
          "attention_bwd_prep_f16", B=B, N=N, H=H)
    nseg = current_segments() if split else 0
    dqkv = _split_buffer(B * N, 3 * H * 64, dev) if split else torch.empty(B, N, 3 * H * 64, dtype=torch.float32, device=dev)
    check(lib().mv_attention_bwd_f16(_p(qkv16), _p(dout16), _p(delta), _p(lse), _p(gscale), _p(dqkv), nseg, _p(colsum), B, N, H,
                                     scale, _s()), "attention_bwd_f16", B=B, N=N, H=H)
    return dqkv


def attention_fwd_f32(qkv, B, N, H, scale):
    """Exact fp32 attention core, forward only (no probabilities kept): qkv fp32 [B, N, 3*H*64] -> out fp32 [B, N, H*64]."""
    require_cuda(qkv)
    out = torch.empty(B, N, H * 64, dtype=torch.float32, device=qkv.device)
    check(lib().mv_attention_fwd_f32(_p(qkv), _p(out), B, N, H, scale, _s()), "attention_fwd_f32", B=B, N=N, H=H)
    return out


def attention_fwd_f32_lse(qkv, B, N, H, scale):
    """fp32 attention core for training: -> (out fp32 [B, N, H*64], lse fp32 [B, H, N]); no probabilities are kept."""
    require_cuda(qkv)
    out = torch.empty(B, N, H * 64, dtype=torch.float32, device=qkv.device)
    lse = torch.empty(B, H, N, dtype=torch.float32, device=qkv.device)
    check(lib().mv_attention_fwd_f32_lse(_p(qkv), _p(out), _p(lse), B, N, H, scale, _s()), "attention_fwd_f32_lse",
          B=B, N=N, H=H)
    return out, lse


def attention_bwd_f32_fused(qkv, out, dout, lse, B, N, H, scale):
    """-> dqkv fp32 (qkv's layout) from the saved output and log-sum-exp."""
    dqkv = torch.empty_like(qkv)
    check(lib().mv_attention_bwd_f32(_p(qkv), _p(out), _p(dout), _p(lse), _p(dqkv), B, N, H, scale, _s()),
          "attention_bwd_f32", B=B, N=N, H=H)
    return dqkv


def attention_probs_fp32(qkv, B, N, H, dh, scale):
    """Materialised path (fp32): probs[B, H, N, N] = softmax(q k^T * scale).  qkv fp32 [B, N, 3, H, dh]."""
    D = H * dh
    scores = torch.empty(B, H, N, N, dtype=torch.float32, device=qkv.device)
    # A = q: [m=n, k=d] strides (3D, 1), batches b: N*3D, h: dh.  B[k=d][n=j] = k[j][d]: strides (1, 3D), offset D
    k_view = qkv.view(-1)[D:]
    check(lib().mv_gemm_f32(_p(qkv), 3 * D, 1, N * 3 * D, dh, _p(k_view), 1, 3 * D, N * 3 * D, dh, _p(scores), N, 1,
                            H * N * N, N * N, N, N, dh, B, H, 1.0, 0, None, EPI_NONE, None, 0, 0, None, 0, _s()),
          "gemm_f32(qk^T)", B=B, H=H, N=N)
    probs = torch.empty_like(scores)
    check(lib().mv_softmax_fwd(_p(scores), _p(probs), B * H * N, N, scale, _s()), "softmax_fwd", rows=B * H * N, cols=N)
    return probs


def attention_pv_fp32(probs, qkv, B, N, H, dh):
    D = H * dh
    out = torch.empty(B, N, D, dtype=torch.float32, device=qkv.device)
    v_view = qkv.view(-1)[2 * D:]
    check(lib().mv_gemm_f32(_p(probs), N, 1, H * N * N, N * N, _p(v_view), 3 * D, 1, N * 3 * D, dh, _p(out), D, 1,
                            N * D, dh, N, dh, N, B, H, 1.0, 0, None, EPI_NONE, None, 0, 0, None, 0, _s()),
          "gemm_f32(pv)", B=B, H=H, N=N)
    return out


def attention_bwd_fp32(probs, qkv, dout, B, N, H, dh, scale):
    """-> dqkv fp32 [B, N, 3, H, dh]"""
    D = H * dh
    dev = qkv.device
    L = lib()
    dqkv = torch.empty_like(qkv)
    flat, dflat = qkv.view(-1), dqkv.view(-1)
    k_view, v_view = flat[D:], flat[2 * D:]
    dk_view, dv_view = dflat[D:], dflat[2 * D:]
    bq, hq = N * 3 * D, dh            # batch strides inside qkv-shaped tensors
    bp, hp = H * N * N, N * N         # batch strides inside [B,H,N,N]
    bo, ho = N * D, dh                # batch strides inside [B,N,D]
    # dP[n][j] = sum_d dO[n][d] V[j][d]
    dP = torch.empty(B, H, N, N, dtype=torch.float32, device=dev)
    check(L.mv_gemm_f32(_p(dout), D, 1, bo, ho, _p(v_view), 1, 3 * D, bq, hq, _p(dP), N, 1, bp, hp, N, N, dh, B, H, 1.0, 0,
                        None, EPI_NONE, None, 0, 0, None, 0, _s()), "gemm_f32(dP)")
    # dV[j][d] = sum_n P[n][j] dO[n][d]
    check(L.mv_gemm_f32(_p(probs), 1, N, bp, hp, _p(dout), D, 1, bo, ho, _p(dv_view), 3 * D, 1, bq, hq, N, dh, N, B, H, 1.0,
                        0, None, EPI_NONE, None, 0, 0, None, 0, _s()), "gemm_f32(dV)")
    dS = torch.empty_like(dP)
    check(L.mv_softmax_bwd(_p(probs), _p(dP), _p(dS), B * H * N, N, scale, _s()), "softmax_bwd")
    # dQ[n][d] = sum_j dS[n][j] K[j][d]
    check(L.mv_gemm_f32(_p(dS), N, 1, bp, hp, _p(k_view), 3 * D, 1, bq, hq, _p(dqkv), 3 * D, 1, bq, hq, N, dh, N, B, H, 1.0,
                        0, None, EPI_NONE, None, 0, 0, None, 0, _s()), "gemm_f32(dQ)")
    # dK[j][d] = sum_n dS[n][j] Q[n][d]
    check(L.mv_gemm_f32(_p(dS), 1, N, bp, hp, _p(qkv), 3 * D, 1, bq, hq, _p(dk_view), 3 * D, 1, bq, hq, N, dh, N, B, H, 1.0,
                        0, None, EPI_NONE, None, 0, 0, None, 0, _s()), "gemm_f32(dK)")
    return dqkv


# ------------------------------------------------------------------------------------------------------------
# embedding assembly, casts, elementwise
# ------------------------------------------------------------------------------------------------------------
def patchify(img, p, out_dtype):
    require_cuda(img)
    B, C, H, W = img.shape
    if img.dtype != torch.float32 or not img.is_contiguous():
        img = img.float().contiguous()
    out = torch.empty(B * (H // p) * (W // p), p * p * C, dtype=out_dtype, device=img.device)
    check(lib().mv_patchify(_p(img), _p(out), _DT[out_dtype], B, C, H, W, p, _s()), "patchify", B=B, C=C, H=H, W=W, p=p)
    return out


def embed_cls(cls, pos, x, B, T, D):
    check(lib().mv_embed_cls(_p(cls), _p(pos), _p(x), B, T, D, _s()), "embed_cls", B=B, T=T, D=D)


def embed_bwd(dx, B, T, D, pos=None, cls_token=None):
    dpos = grad_out(pos, (T, D), dx.device)
    dcls = grad_out(cls_token, (D,), dx.device)
    check(lib().mv_embed_bwd(_p(dx), _p(dpos), _p(dcls), 0, B, T, D, _s()), "embed_bwd", B=B, T=T, D=D)
    return dpos, dcls


def embed_bwd_gather(dx, B, T, D, out_dtype, pos=None, cls_token=None):
    """One pass over dx [B, T, D]: (dpos [T, D], dcls [D], dy [B * (T - 1), D] in ``out_dtype``)."""
    dpos = grad_out(pos, (T, D), dx.device)
    dcls = grad_out(cls_token, (D,), dx.device)
    dy = torch.empty(B * (T - 1), D, dtype=out_dtype, device=dx.device)
    check(lib().mv_embed_bwd_gather(_p(dx), _p(dy), _DT[out_dtype], _p(dpos), _p(dcls), B, T, D, _s()), "embed_bwd_gather",
          B=B, T=T, D=D)
    return dpos, dcls, dy


def gather_patch_rows(src, B, T, D, out_dtype):
    out = torch.empty(B * (T - 1), D, dtype=out_dtype, device=src.device)
    check(lib().mv_gather_patch_rows(_p(src), _p(out), _DT[out_dtype], B, T, D, _s()), "gather_patch_rows", B=B, T=T, D=D)
    return out


def cast(src, out_dtype):
    require_cuda(src)
    if src.dtype == out_dtype:
        return src
    out = torch.empty(src.shape, dtype=out_dtype, device=src.device)
    check(lib().mv_cast(_p(src), _DT[src.dtype], _p(out), _DT[out_dtype], src.numel(), _s()), "cast", n=src.numel())
    return out


def cast_into(src, dst):
    """dst <- src with a dtype conversion, into caller-owned storage (the DDP exchange's bf16 staging copies)."""
    require_cuda(src, dst)
    if src.numel() != dst.numel() or not (src.is_contiguous() and dst.is_contiguous()):
        raise ValueError("cast_into: contiguous tensors of equal size")
    check(lib().mv_cast(_p(src), _DT[src.dtype], _p(dst), _DT[dst.dtype], src.numel(), _s()), "cast", n=src.numel())
    return dst


def gelu_fwd(x):
    y = torch.empty_like(x)
    check(lib().mv_gelu_fwd(_p(x), _p(y), _DT[x.dtype], x.numel(), _s()), "gelu_fwd", n=x.numel())
    return y


def gelu_bwd(x, dy):
    dx = torch.empty_like(x)
    check(lib().mv_gelu_bwd(_p(x), _p(dy), _p(dx), _DT[x.dtype], x.numel(), _s()), "gelu_bwd", n=x.numel())
    return dx


def add_f32(a, b):
    out = torch.empty_like(a)
    check(lib().mv_add_f32(_p(a), _p(b), _p(out), a.numel(), _s()), "add_f32", n=a.numel())
    return out


# ------------------------------------------------------------------------------------------------------------
# fake quantisation
# ------------------------------------------------------------------------------------------------------------
def quant_float(x, exp_bits, man_bits):
    require_cuda(x)
    xf = x.detach().float().contiguous()
    y = torch.empty_like(xf)
    check(lib().mv_quant_float(_p(xf), _p(y), xf.numel(), exp_bits, man_bits, _s()), "quant_float", n=xf.numel())
    return y


def quant_float_f16(x):
    """float_quantize(x, exp=5, man=10) stored as IEEE half (exact) -- an operand of ``linear_f16``."""
    require_cuda(x)
    xf = x.detach().float().contiguous()
    y = torch.empty(xf.shape, dtype=torch.float16, device=xf.device)
    check(lib().mv_quant_float_f16(_p(xf), _p(y), xf.numel(), _s()), "quant_float_f16", n=xf.numel())
    return y


def linear_f16_supported(M, N, K):
    """Shapes the f16 MFMA GEMM takes (the 8-phase kernel: K % 128 == 0 and a grid worth a launch)."""
    return K % 128 == 0 and M >= 256 and N >= 256


def linear_f16(x16, w16, M, N, K, bias, out, residual=None):
    """out fp32 [M, N] = x16 [M, K] . w16 [N, K]^T + bias (+ residual): half operands on v_mfma_f32_16x16x32_f16, fp32
    accumulation."""
    t0 = _timer.begin() if _timer is not None else None
    epi, aux, ld_aux = (EPI_RESIDUAL, _p(residual), N) if residual is not None else (EPI_NONE, None, 0)
    check(lib().mv_gemm_nt_f16(_p(x16), K, _p(w16), K, _p(out), N, M, N, K, _p(bias), epi, aux, ld_aux, None, 0, _s()),
          "gemm_nt_f16", M=M, N=N, K=K)
    if t0 is not None:
        _timer.end("gemm_nt_f16", t0, 2.0 * M * N * K, shape=f"f16 N{N} K{K} epi{epi}")
    return out


def quant_fixed(x, wl, fl, clamp=True, symmetric=False):
    require_cuda(x)
    xf = x.detach().float().contiguous()
    y = torch.empty_like(xf)
    check(lib().mv_quant_fixed(_p(xf), _p(y), xf.numel(), wl, fl, int(clamp), int(symmetric), _s()), "quant_fixed",
          n=xf.numel())
    return y


def quant_affine(x, scale, zero_point, qmin, qmax):
    require_cuda(x)
    xf = x.detach().float().contiguous()
    y = torch.empty_like(xf)
    check(lib().mv_quant_affine(_p(xf), _p(y), xf.numel(), float(scale), int(zero_point), qmin, qmax, _s()),
          "quant_affine", n=xf.numel())
    return y


def quant_affine_codes(x, rows, cols, scale, zero_point, qmin, qmax, pre_gelu=False):
    """fp32 or bf16 [rows, cols] -> bf16 [rows, pad8(cols)] integer codes (q - zero_point) of the affine quantiser
    (exact); ``pre_gelu``: of gelu(x)."""
    require_cuda(x)
    xf = x.detach()
    if xf.dtype not in (torch.float32, torch.bfloat16):
        xf = xf.float()
    xf = xf.contiguous()
    ld = pad8(cols)
    codes = torch.empty(rows, ld, dtype=torch.bfloat16, device=x.device)
    check(lib().mv_quant_affine_codes(_p(xf), _DT[xf.dtype], _p(codes), rows, cols, ld, float(scale), int(zero_point), qmin,
                                      qmax, 1 if pre_gelu else 0, _s()),
          "quant_affine_codes", rows=rows, cols=cols)
    return codes


def linear_codes(xc, wc, M, N, K, alpha, bias, out, residual=None):
    """out (fp32 or bf16) [M, N] = alpha * (xc [M, pad8(K)] . wc [N, pad8(K)]^T) + bias (+ residual fp32 [M, N]):
    integer-code operands on the bf16 MFMA path = exact int8 arithmetic (|codes| <= 256, fp32 accumulation)."""
    t0 = _timer.begin() if _timer is not None else None
    epi, aux, ld_aux = (EPI_RESIDUAL, _p(residual), N) if residual is not None else (EPI_NONE, None, 0)
    check(lib().mv_gemm_nt_bf16_scaled(_p(xc), xc.shape[1], _p(wc), wc.shape[1], _p(out), N, _DT[out.dtype], M, N, K,
                                       float(alpha), _p(bias), epi, aux, ld_aux, 0, None, 0, _s()),
          "gemm_nt_bf16_scaled", M=M, N=N, K=K)
    if t0 is not None:
        _timer.end("gemm_nt_bf16", t0, 2.0 * M * N * K)
    return out


def pad16(n: int) -> int:
    return (n + 15) & ~15


def quant_affine_i8(x, rows, cols, scale, zero_point, pre_gelu=False):
    """fp32 or bf16 [rows, cols] -> int8 [rows, pad16(cols)] codes q - 128 of the quint8 affine quantiser (exact);
    ``pre_gelu``: of gelu(x).  Feeds ``linear_i8``."""
    require_cuda(x)
    xf = x.detach()
    if xf.dtype not in (torch.float32, torch.bfloat16):
        xf = xf.float()
    xf = xf.contiguous()
    ld = pad16(cols)
    codes = torch.empty(rows, ld, dtype=torch.int8, device=x.device)
    check(lib().mv_quant_affine_i8(_p(xf), _DT[xf.dtype], _p(codes), rows, cols, ld, float(scale), int(zero_point),
                                   1 if pre_gelu else 0, _s()), "quant_affine_i8", rows=rows, cols=cols)
    return codes


def linear_i8_supported(M, N, K):
    """Shapes the int8 MFMA GEMM takes (8-phase kernel: two whole 128-byte K-tiles per iteration, a grid worth a launch)."""
    return K % 256 == 0 and M >= 256 and N >= 256


def linear_i8(x8, w8, M, N, K, alpha, bias, icorr, out, residual=None, gelu_q8=None):
    """out (fp32 or bf16) [M, N] = alpha * (x8 [M, pad16(K)] . w8 [N, pad16(K)]^T + icorr[N]) + bias (+ residual fp32):
    int8 operands on v_mfma_i32_16x16x64_i8, int32 accumulation.  ``gelu_q8 = (scale, zero_point)`` of the NEXT layer's
    quint8 quantiser: ``out`` is int8 [M, N] and receives that quantiser's codes (q - 128) of gelu(the product)."""
    t0 = _timer.begin() if _timer is not None else None
    epi, aux, ld_aux = (EPI_RESIDUAL, _p(residual), N) if residual is not None else (EPI_NONE, None, 0)
    cdt, qs, qz = _DT.get(out.dtype), 0.0, 0
    if gelu_q8 is not None:
        if residual is not None or out.dtype != torch.int8:
            raise ValueError("gelu_q8 needs an int8 output and no residual")
        epi, cdt, qs, qz = _l.EPI_GELU_Q8, _l.MV_I8, float(gelu_q8[0]), int(gelu_q8[1])
    check(lib().mv_gemm_nt_i8(_p(x8), x8.shape[1], _p(w8), w8.shape[1], _p(out), N, cdt, M, N, K, float(alpha),
                              _p(bias), _p(icorr), epi, aux, ld_aux, qs, qz, _s()), "gemm_nt_i8", M=M, N=N, K=K, epi=epi)
    if t0 is not None:
        _timer.end("gemm_nt_i8", t0, 2.0 * M * N * K, shape=f"i8 N{N} K{K} epi{epi}")
    return out


def layernorm_q8(x, ldx, rows, dim, gamma, beta, eps, scale, zero_point):
    """LayerNorm whose output goes straight into a quint8 quantiser: -> int8 codes (q - 128) [rows, dim]."""
    require_cuda(x, gamma, beta)
    codes = torch.empty(rows, dim, dtype=torch.int8, device=x.device)
    check(lib().mv_layernorm_fwd_q8(_p(x), ldx, _p(gamma), _p(beta), _p(codes), rows, dim, float(eps), float(scale),
                                    int(zero_point), _s()), "layernorm_fwd_q8", rows=rows, dim=dim)
    return codes


def attention_fwd_f32_q8(qkv, B, N, H, scale, q_scale, q_zero_point):
    """Exact fp32 attention core whose output goes straight into a quint8 quantiser: -> int8 codes [B, N, H*64]."""
    require_cuda(qkv)
    codes = torch.empty(B, N, H * 64, dtype=torch.int8, device=qkv.device)
    check(lib().mv_attention_fwd_f32_q8(_p(qkv), _p(codes), B, N, H, scale, float(q_scale), int(q_zero_point), _s()),
          "attention_fwd_f32_q8", B=B, N=N, H=H)
    return codes


def minmax_update(x, state):
    """state: fp32 [4] on device, [0]=running min, [1]=running max (init +inf/-inf)."""
    require_cuda(x, state)                     # the kernel updates ``state`` with device atomics: never a host pointer
    if state.device != x.device or state.dtype != torch.float32 or not state.is_contiguous():
        raise RuntimeError("minmax_update: state must be a contiguous fp32 tensor on the input's device")
    xf = x.detach().float().contiguous()
    check(lib().mv_minmax(_p(xf), xf.numel(), _p(state), _s()), "minmax", n=xf.numel())


# ------------------------------------------------------------------------------------------------------------
# loss, upsample, optimizer
# ------------------------------------------------------------------------------------------------------------
def cross_entropy(logits, labels, *, want_grad, grad_dtype=torch.float32, ld_dl=None, want_argmax=False):
    """Mean CE.  logits fp32 [B, C] or [B, C, H, W]; labels int64.  -> (loss[1], dlogits|None, argmax|None)"""
    require_cuda(logits, labels)
    logits = logits.contiguous()
    labels = labels.contiguous()
    if labels.dtype != torch.int64:
        labels = labels.long()
    C = logits.shape[1]
    outer = logits.shape[0]
    inner = 1
    for s in logits.shape[2:]:
        inner *= s
    stat = torch.empty(4, dtype=torch.float32, device=logits.device)   # mean loss, counted labels, bad labels, unused
    loss = stat[:1]
    dl = None
    if want_grad:
        if inner == 1:
            ld_dl = C if ld_dl is None else ld_dl
            dl = torch.empty(outer, ld_dl, dtype=grad_dtype, device=logits.device)
        else:
            ld_dl = C
            dl = torch.empty(logits.shape, dtype=grad_dtype, device=logits.device)
    am = torch.empty(labels.shape, dtype=torch.int64, device=logits.device) if want_argmax else None
    check(lib().mv_cross_entropy(_p(logits), _p(labels), _p(stat), _p(dl), _DT[grad_dtype], ld_dl or C, _p(am), outer, C,
                                 inner, 1.0, _s()), "cross_entropy", outer=outer, C=C, inner=inner)
    return loss, dl, am


def upsample_bilinear_fwd(small, sb, sc, sp, B, C, h, w, H, W):
    big = torch.empty(B, C, H, W, dtype=torch.float32, device=small.device)
    check(lib().mv_upsample_bilinear_fwd(_p(small), sb, sc, sp, _p(big), B, C, h, w, H, W, _s()), "upsample_fwd",
          B=B, C=C, h=h, w=w, H=H, W=W)
    return big


def upsample_bilinear_bwd(dbig, dsmall, sb, sc, sp, B, C, h, w, H, W):
    check(lib().mv_upsample_bilinear_bwd(_p(dbig), _p(dsmall), sb, sc, sp, B, C, h, w, H, W, _s()), "upsample_bwd",
          B=B, C=C, h=h, w=w, H=H, W=W)
    return dsmall


SEG_CE_MAX_CLASSES = 32
SEG_CE_LDS_LIMIT = 64 * 1024


def seg_ce_supported(C, h, w, W):
    """Shapes the fused segmentation tail covers (mv_seg_ce_* return MV_ERR_UNSUPPORTED outside them)."""
    return C <= SEG_CE_MAX_CLASSES and (h * w * C + W * C) * 4 <= SEG_CE_LDS_LIMIT


def seg_ce_fwd(small, labels, B, C, h, w, H, W):
    """small fp32 [B*h*w, C], labels int64 [B,H,W] -> (stats[2] = mean loss, pixel accuracy; lse [B,H,W]; pred u8)."""
    require_cuda(small, labels)
    labels = labels.contiguous()
    if labels.dtype != torch.int64:
        labels = labels.long()
    dev = small.device
    lse = torch.empty(B, H, W, dtype=torch.float32, device=dev)
    pred = torch.empty(B, H, W, dtype=torch.uint8, device=dev)
    nblk = lib().mv_seg_ce_partials(B, H, W)
    partials = torch.empty(max(4 * nblk, 4), dtype=torch.float32, device=dev)
    stats = torch.empty(4, dtype=torch.float32, device=dev)      # mean loss, pixel accuracy, counted labels, bad labels
    check(lib().mv_seg_ce_fwd(_p(small), _p(labels), _p(lse), _p(pred), _p(partials), _p(stats), B, C, h, w, H, W, _s()),
          "seg_ce_fwd", B=B, C=C, h=h, w=w, H=H, W=W)
    return stats, lse, pred, labels


def seg_ce_bwd(small, labels, lse, B, C, h, w, H, W, *, grad_dtype=torch.float32, ld=None, grad_scale=1.0, stats=None):
    """``stats``: the forward's (its label count is the mean's denominator); None = every pixel counts."""
    ld = C if ld is None else ld
    ds = torch.empty(B * h * w, ld, dtype=grad_dtype, device=small.device)
    check(lib().mv_seg_ce_bwd(_p(small), _p(labels), _p(lse), _p(stats), _p(ds), _DT[grad_dtype], ld, grad_scale, B, C, h, w,
                              H, W, _s()), "seg_ce_bwd", B=B, C=C, h=h, w=w, H=H, W=W, ld=ld)
    return ds


def image_prepare(raw, kh, bh, kv, bv, flip, mean, std):
    """uint8 [B, Hs, Ws, 3] + Pillow resampling tables -> fp32 [B, 3, oh, ow] (crop/resize/flip/ToTensor/Normalize)."""
    require_cuda(raw, kh, bh, kv, bv, flip)
    B, Hs, Ws, _ = raw.shape
    oh, ow, ks = kv.shape[1], kh.shape[1], kh.shape[2]
    out = torch.empty(B, 3, oh, ow, dtype=torch.float32, device=raw.device)
    check(lib().mv_image_prepare(_p(raw), Hs * Ws * 3, Hs, Ws, _p(kh), _p(bh), _p(kv), _p(bv), ks, _p(flip),
                                 mean[0], mean[1], mean[2], std[0], std[1], std[2], _p(out), B, oh, ow, _s()),
          "image_prepare", B=B, Hs=Hs, Ws=Ws, oh=oh, ow=ow, ks=ks)
    return out


def image_resize_u8(raw, kh, bh, kv, bv):
    """uint8 [B, Hs, Ws, 3] + Pillow resampling tables -> uint8 [B, oh, ow, 3]: a plain Image.resize(BILINEAR), kept as
    the uint8 image a following RandomResizedCrop resamples again."""
    require_cuda(raw, kh, bh, kv, bv)
    B, Hs, Ws, _ = raw.shape
    oh, ow, ks = kv.shape[1], kh.shape[1], kh.shape[2]
    out = torch.empty(B, oh, ow, 3, dtype=torch.uint8, device=raw.device)
    check(lib().mv_image_resize_u8(_p(raw), Hs * Ws * 3, Hs, Ws, _p(kh), _p(bh), _p(kv), _p(bv), ks, _p(out), B, oh, ow, _s()),
          "image_resize_u8", B=B, Hs=Hs, Ws=Ws, oh=oh, ow=ow, ks=ks)
    return out


def mask_resize_u8(mask, yi, xi):
    """uint8 [B, Hs, Ws] + NEAREST index tables -> uint8 [B, oh, ow]."""
    require_cuda(mask, yi, xi)
    B, Hs, Ws = mask.shape
    oh, ow = yi.shape[1], xi.shape[1]
    out = torch.empty(B, oh, ow, dtype=torch.uint8, device=mask.device)
    check(lib().mv_mask_resize_u8(_p(mask), Hs * Ws, Hs, Ws, _p(yi), _p(xi), _p(out), B, oh, ow, _s()),
          "mask_resize_u8", B=B, Hs=Hs, Ws=Ws, oh=oh, ow=ow)
    return out


def mask_prepare(mask, yi, xi, flip, add=0):
    """uint8 [B, Hs, Ws] + NEAREST index tables -> int64 [B, oh, ow] (+ add)."""
    require_cuda(mask, yi, xi, flip)
    B, Hs, Ws = mask.shape
    oh, ow = yi.shape[1], xi.shape[1]
    out = torch.empty(B, oh, ow, dtype=torch.int64, device=mask.device)
    check(lib().mv_mask_prepare(_p(mask), Hs * Ws, Hs, Ws, _p(yi), _p(xi), _p(flip), add, _p(out), B, oh, ow, _s()),
          "mask_prepare", B=B, Hs=Hs, Ws=Ws, oh=oh, ow=ow)
    return out


def adamw_step(p, g, m, v, *, lr, beta1, beta2, eps, weight_decay, step, grad_scale=1.0, clip_coef=None, hyper=None):
    """In-place AdamW on flat fp32 tensors (torch.optim.AdamW semantics).  ``clip_coef``: fp32 device scalar multiplied
    into every gradient (``grad_norm_clip``).  ``hyper`` (fp32 [3] on the device: lr, 1 - beta1^step, 1 - beta2^step)
    replaces ``lr`` / ``step``: what a launch captured in a HIP graph reads, see ``utils.optim.AdamW.use_device_scalars``."""
    require_cuda(p, g, m, v, clip_coef, hyper)
    if hyper is not None:
        check(lib().mv_adamw_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(hyper), beta1, beta2, eps, weight_decay, grad_scale,
                                 _p(clip_coef), _s()), "adamw_dev", n=p.numel())
        return
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    check(lib().mv_adamw(_p(p), _p(g), _p(m), _p(v), p.numel(), lr, beta1, beta2, eps, weight_decay, bc1, bc2,
                         grad_scale, _p(clip_coef), _s()), "adamw", n=p.numel())


def grad_norm_clip(g, max_norm, grad_scale=1.0):
    """-> fp32 [2] on the device: (total_norm of g * grad_scale, min(1, max_norm / (total_norm + 1e-6))) -- the two numbers
    of torch.nn.utils.clip_grad_norm_ (classification/train.py:265-270); no host synchronisation."""
    require_cuda(g)
    out = torch.empty(2, dtype=torch.float32, device=g.device)
    ws = workspace(lib().mv_grad_norm_workspace_bytes(), g.device)
    check(lib().mv_grad_norm_clip(_p(g), g.numel(), float(grad_scale), float(max_norm), _p(out), _p(ws), ws.numel(), _s()),
          "grad_norm_clip", n=g.numel())
    return out


def dropout(x, p, seed, offset, out=None):
    """x * keep / (1 - p) with the Philox mask of (seed, offset); fp32 or bf16, any shape (contiguous)."""
    require_cuda(x)
    x = x.contiguous()
    y = torch.empty_like(x) if out is None else out
    check(lib().mv_dropout(_p(x), _p(y), _DT[x.dtype], x.numel(), float(p), int(seed), int(offset), _s()), "dropout",
          n=x.numel(), p=p)
    return y
