"""Data-parallel gradient exchange for one process per GPU (reference: DistributedDataParallel(vit, device_ids=[rank]),
classification/train.py:156; DDP's bucketed all-reduce overlapped with backward).

MI355X-native shape: gradients already live in ONE flat fp32 buffer (``optim.ParamArena``), so a bucket is a
contiguous slice -- no packing/unpacking copies.  A bucket is all-reduced (SUM, RCCL over xGMI via
``torch.distributed`` backend "nccl") as soon as the last gradient in it has been accumulated, from autograd's
post-accumulate hooks, i.e. overlapped with the rest of backward on RCCL's own stream.  The 1/world_size of DDP's
mean is folded into the optimizer kernel (``AdamW.grad_scale``) instead of a separate pass over 344 MB.

Parameters that never receive a gradient (``pos_embedding_det``, ``det_tokens``: SURVEY 9.1, the reason the
reference's own DDP fails on its 2nd iteration) are simply not in the arena; any bucket still incomplete when
backward ends is reduced in ``finish()``, so a missing gradient can never deadlock the ranks.

xGMI is point-to-point (7 links x ~153 GB/s per GPU), a ring all-reduce is per-link bound, so buckets are large
(default 48 MiB): 86 M fp32 gradients = 8 collectives per step rather than torch-DDP's 14 x 25 MiB, and the bucket that can
only complete at the end of backward (the first layers') stays below 7 % of the gradient bytes.

``exchange_dtype=torch.bfloat16`` (opt-in; ``MV_DDP_EXCHANGE=bf16`` in the training loops and bench.py) halves the bytes on
the links: a bucket is rounded to bf16 into a staging buffer when it completes, summed in bf16 by the collective, and
widened back into the fp32 arena in ``finish()``.  The fp32 default is bit-for-bit DDP's mean; the bf16 form adds one
rounding of each rank's gradient and the collective's bf16 partial sums (<= ~1e-2 relative per element at 8 ranks: the size
of the bf16 backward's own error, tests/test_host_cpu.py::test_gradient_allreduce_bf16_exchange_world2_gloo).

``measure=True`` brackets ``finish()`` with events on the compute stream: the time between the end of backward and the
last collective's completion as the GPU sees it -- the EXPOSED part of the exchange (``exposed_ms()``); bench.py reports it.
"""
from typing import List

import torch
import torch.distributed as dist


class GradAllReducer:
    def __init__(self, arena, process_group=None, bucket_bytes: int = 48 << 20, tail_bytes: int = 12 << 20,
                 exchange_dtype=torch.float32, measure: bool = False):
        if exchange_dtype not in (torch.float32, torch.bfloat16):
            raise ValueError(f"exchange_dtype {exchange_dtype}: float32 or bfloat16")
        self.arena = arena
        self.exchange_dtype = exchange_dtype
        self.stage = (torch.empty(arena.total, dtype=torch.bfloat16, device=arena.flat_grad.device)
                      if exchange_dtype == torch.bfloat16 else None)
        self.measure = measure and arena.flat_grad.is_cuda
        self._spans = []                                       # (event at finish() entry, event after the last wait)
        self.group = process_group
        self.world = dist.get_world_size(process_group) if dist.is_initialized() else 1
        # bucket boundaries fall on parameter boundaries; buckets are filled from the END of the arena, because
        # backward produces the last layers' gradients first
        # A bucket never straddles the boundary between the two weight-decay regions: the no-decay region (every bias and
        # LayerNorm parameter of EVERY layer, < 1 MB) is complete only when the first layer's gradients are, i.e. at the very
        # end of backward -- sharing a bucket with it held the last layers' 57 MiB of weight gradients back until then
        # (tools/ddp_overlap_timeline.py: 23 % of the gradient bytes were enqueued after backward; now < 7 %).
        # The LAST buckets to complete (the first layers' weights: the low end of the decay region, which starts the arena)
        # taper to ``tail_bytes``:
        # whatever is enqueued in the final instants of backward is the exposed part of the exchange, and with uniform
        # 48 MiB buckets that was 47 MiB at 0.3 ms before the end (timeline tool); tapered it is one or two parameters.
        cap_full, cap_tail = max(bucket_bytes // 4, 1), max(min(tail_bytes, bucket_bytes) // 4, 1)
        n_decay_params = sum(1 for o in arena.offsets if o < arena.n_decay)
        self.ranges: List[tuple] = []
        hi = arena.total
        lo_idx = len(arena.params)
        while lo_idx > 0:
            j = lo_idx
            lo = hi
            floor = n_decay_params if lo_idx > n_decay_params else 0          # first parameter index this bucket may reach
            cap = cap_tail if lo_idx <= n_decay_params and hi <= 2 * cap_full else cap_full   # decay region = [0, n_decay)
            while j > floor and (hi - arena.offsets[j - 1]) <= cap or j == lo_idx:
                j -= 1
                lo = arena.offsets[j]
                if j == floor:
                    break
            self.ranges.append((lo, hi, j, lo_idx))
            hi, lo_idx = lo, j
        self.bucket_of = {}
        self.sizes = []
        for b, (lo, hi, j0, j1) in enumerate(self.ranges):
            self.sizes.append(j1 - j0)
            for j in range(j0, j1):
                self.bucket_of[j] = b
        self.pending = list(self.sizes)
        self.launched = [False] * len(self.ranges)
        self.handles = []
        self.enabled = self.world > 1
        self._hooks = []
        for j, p in enumerate(arena.params):
            self._hooks.append(p.register_post_accumulate_grad_hook(self._make_hook(j)))

    def _make_hook(self, j):
        def hook(param):
            if not self.enabled:
                return
            self.arena.sync_grad(j)              # no-op when the gradient was written in place (the normal case)
            b = self.bucket_of[j]
            self.pending[b] -= 1
            if self.pending[b] == 0 and not self.launched[b]:
                self._launch(b)
        return hook

    def _launch(self, b):
        lo, hi, _, _ = self.ranges[b]
        self.launched[b] = True
        buf = self.arena.flat_grad[lo:hi]
        if self.stage is not None:
            _convert(buf, self.stage[lo:hi])                   # on the compute stream, behind the kernels that wrote the bucket
            buf = self.stage[lo:hi]
        self.handles.append(dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self):
        """Call after backward: reduce whatever has not been reduced, wait for every collective, re-arm."""
        if self.enabled:
            span = None
            if self.measure:
                span = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                span[0].record()                              # everything backward enqueued is in front of this
            for b in range(len(self.ranges)):
                if not self.launched[b]:
                    for j in range(self.ranges[b][2], self.ranges[b][3]):
                        if self.arena.params[j].grad is None:
                            self.arena.sync_grad(j)          # no gradient this step: the slot must hold zeros
                    self._launch(b)
            for h in self.handles:
                h.wait()                                      # the compute stream waits for the collective's stream
            if self.stage is not None:                        # the buckets tile [0, total): widen back in one launch
                _convert(self.stage, self.arena.flat_grad)
            if span is not None:
                span[1].record()
                self._spans.append(span)
        self.handles = []
        self.pending = list(self.sizes)
        self.launched = [False] * len(self.ranges)

    def exposed_ms(self, reset: bool = True):
        """Mean GPU time per step between the end of backward and the completion of the exchange (``measure=True``), or
        None.  Synchronises the device: call it outside the timed region."""
        if not self._spans:
            return None
        torch.cuda.synchronize()
        ms = sum(a.elapsed_time(b) for a, b in self._spans) / len(self._spans)
        if reset:
            self._spans = []
        return ms

    def describe(self) -> dict:
        """What one step puts on the links (for the benchmark's ``dist`` record)."""
        esz = 2 if self.stage is not None else 4
        sizes = [(hi - lo) * esz for lo, hi, _, _ in self.ranges]
        return {"buckets": len(self.ranges), "bytes": int(sum(sizes)), "largest_bucket_bytes": int(max(sizes)),
                "exchange_dtype": "bf16" if self.stage is not None else "fp32"}

    @property
    def grad_scale(self) -> float:
        """Multiply summed gradients by this to get DDP's mean."""
        return 1.0 / self.world

    def remove(self):
        for h in self._hooks:
            h.remove()


def _convert(src, dst):
    """dst <- src across fp32 / bf16 (the staging copies of the bf16 exchange): the HIP cast kernel on the device, torch on
    the host (the gloo tests)."""
    if src.is_cuda:
        from myrtle_vision.hip import ops
        ops.cast_into(src, dst)
    else:
        dst.copy_(src)


def exchange_dtype_from_env():
    """``MV_DDP_EXCHANGE`` = fp32 (default) | bf16."""
    import os
    v = os.environ.get("MV_DDP_EXCHANGE", "fp32").lower()
    if v not in ("fp32", "bf16"):
        raise ValueError(f"MV_DDP_EXCHANGE={v!r}: fp32 or bf16")
    return torch.bfloat16 if v == "bf16" else torch.float32


def broadcast_parameters(arena, src: int = 0, process_group=None):
    """DDP's constructor broadcast (classification/train.py:156): one collective over the flat arena."""
    if dist.is_initialized() and dist.get_world_size(process_group) > 1:
        dist.broadcast(arena.flat_param, src=src, group=process_group)
        arena.bump_versions()
