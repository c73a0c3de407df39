"""The whole training step as ONE HIP graph (hipGraph through ``torch.cuda.CUDAGraph``).

The reference's shipped configs train at ``local_batch_size`` 32 (classification/train_configs/vit_base.json): a ViT-B step
is then ~1 200 kernel launches of a few microseconds each, and the time between kernels (launch latency, Python) is a
visible share of it.  A static-shape step -- zero_grad, forward, loss, backward, AdamW (classification/train.py:239-279) --
is a fixed sequence of launches on fixed addresses, so it is captured once and replayed: one host call per step.

What makes the step capturable here:
* every kernel is launched on the stream the C ABI is handed (= torch's current stream, the capturing one);
* nothing in the step synchronises with the host, allocates outside torch's graph pool, or reads a host scalar that changes:
  the learning rate and AdamW's bias corrections come from device memory (``AdamW.use_device_scalars``), the clipping
  coefficient already did; the loss stays on the device;
* gradients are written straight into the flat arena (no accumulation across micro-batches: ``n_batch_accum == 1``);
* dropout must be off (its mask seed is a host draw per call: a replay would repeat one mask) -- checked.
Single process only: the RCCL exchange is not captured (world size 1); a multi-GPU run keeps the eager step.

After every replay the epoch of THIS optimizer's arena moves (``ParamArena.bump_versions``): the graph refreshed its bf16
weight copies BEFORE the optimizer step it contains, so an eager forward afterwards (validation) must prepare them again.
The graph keeps alive every device table and weight copy its captured preparation launches read or write.
"""
import torch

from myrtle_vision.hip import ops


class GraphedTrainStep:
    """``step = GraphedTrainStep(model, optimizer, loss_fn, example_inputs, example_labels)``; ``loss = step(inputs, labels)``.

    ``loss_fn(model, inputs, labels) -> scalar loss tensor`` (e.g. ``lambda m, x, y: cross_entropy(m(x), y)``).  Inputs of
    the example's shape and dtype are copied into the graph's static buffers; the returned loss is the graph's static
    output (read it before the next call, or ``.clone()`` it)."""

    def __init__(self, model, optimizer, loss_fn, example_inputs, example_labels, warmup: int = 3):
        if not example_inputs.is_cuda:
            raise RuntimeError("GraphedTrainStep needs device tensors (the HIP path has no CPU fallback)")
        for m in model.modules():
            if type(m).__name__ == "Dropout" and getattr(m, "p", 0.0) > 0.0 and m.training:
                raise ValueError("a HIP-graph step cannot contain dropout (p > 0): its mask seed is drawn on the host per call")
        if torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1:
            raise ValueError("GraphedTrainStep is single-process: the gradient exchange is not captured")
        if ops._timer is not None:
            raise RuntimeError("switch the KernelTimer off before capturing (it records events between launches)")
        self.model, self.optimizer, self.loss_fn = model, optimizer, loss_fn
        self.inputs = example_inputs.clone()
        self.labels = example_labels.clone()
        if optimizer._hyper is None:
            optimizer.use_device_scalars()
        # warm-up on a side stream (torch's capture recipe): every kernel's one-time setup (function attributes, workspaces,
        # weight-cache entries) happens here, not inside the capture.  These are REAL optimizer steps on the example batch.
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._eager_step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        optimizer.zero_grad()
        optimizer.arena.bump_versions()                 # the capture must CONTAIN the weight preparation: every copy stale
        # The graph OWNS what it captured: the device table of the batched weight preparation and the bf16 copies it writes
        # (ops._capture_keep).  The preparation inside the capture covers THIS optimizer's arena only (ops._refresh_stale is
        # per owner), so other models may come and go between replays without the graph reading a recycled table.
        self._keep = []
        ops._capture_keep = self._keep
        try:
            with torch.cuda.graph(self.graph):
                self.loss = self._eager_step(advance=False)
        finally:
            ops._capture_keep = None
        self.warmup_steps = warmup

    def _eager_step(self, advance=True):
        self.optimizer.zero_grad()
        loss = self.loss_fn(self.model, self.inputs, self.labels)
        loss.backward()
        self.optimizer.step()                           # outside a capture it advances the device scalars itself
        return loss.detach()

    def __call__(self, inputs, labels):
        self.inputs.copy_(inputs, non_blocking=True)
        self.labels.copy_(labels, non_blocking=True)
        self.optimizer.advance()                        # step count, lr, bias corrections -> device (three floats per group)
        self.graph.replay()
        self.optimizer.arena.bump_versions()            # parameters changed behind the host-side caches' back (this arena only)
        return self.loss
