#!/usr/bin/env python3
"""Headline benchmark: ViT-B/16 224^2 bf16 training step, images/sec (BASELINE.json `metric`).

    python bench.py --gpus N --steps K --warmup W          (N > 1: starts N fresh children itself, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU.  A "step" = zero_grad + forward + cross-entropy + backward (+ gradient all-reduce over
RCCL/xGMI overlapped with backward when N > 1) + AdamW, on a synthetic batch that is already resident in HBM
(per-GPU batch 256, weak scaling: global batch = 256 * N).  Rank 0 prints ONE JSON line.

`roofline`: the dominant kernel family is gemm_nt_8phase_kernel (mv_gemm_nt_bf16: every nn.Linear forward and input-gradient
product, 2/3 of all FLOPs).  achieved = sum of algorithmic FLOPs (2*M*N*K per launch) / sum of launch durations,
measured with events recorded on the launch stream around every launch of two steps of the timed region (--timer-every).
`cpu_baseline`: the CPU oracle (oracle/vit_oracle.py, kind "port") timed on this host's cores, rank 0, N=1 only.
`within_tolerance` (N=1, default workload): the same step in precision "bf16x3h" -- the fastest arithmetic that meets the reference
tolerance (1e-3, exact arg-max) -- measured after the timed region; `value` stays the bf16 step BASELINE.json names.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "myrtle-vision_amd"))
sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

TRAIN_GFLOP_PER_IMG = 105.38        # fwd 35.13 GFLOP x 3
BF16_DENSE_PEAK_TFLOPS = 2500.0     # MI355X_MICROARCH.md: ~2.5 PF dense bf16
METRIC = {"cls": "images/sec (train fwd+bwd) ViT-B/16 224^2 bf16",
          "seg": "images/sec (train fwd+bwd) ViT-B/16 segmentation 224^2",
          "seg256": "images/sec (train fwd+bwd) ViT-B/16 segmentation 256^2",
          "infer-int8": "images/sec (inference fwd) ViT-B/16 224^2 affine-int8 fake-quant"}
WORKLOAD = {"cls": "ViT-B/16 224^2 classification training step: zero_grad + fwd + CE + bwd + grad all-reduce + AdamW; "
                   "1000 classes; random-init weights; batch resident in HBM",
            "seg": "ViT-B/16 224^2 segmentation training step (17 classes, per-pixel CE over the bilinear-upsampled "
                   "14x14 map): zero_grad + fwd + CE + bwd + grad all-reduce + AdamW; batch resident in HBM",
            "seg256": "ViT-B/16 256^2 segmentation training step (17 classes, N=257 tokens, bicubic pos-emb resize)",
            "infer-int8": "ViT-B/16 224^2 forward only, min/max-calibrated per-tensor affine fake-quant (Q8 sites)"}
PRECISION_DTYPE = {
    "bf16x3": "bf16x3 (fp32 tensors; every nn.Linear product from two bf16 pieces per operand, three pairings on the bf16 MFMA "
              "with fp32 accumulation = 2^-16 relative; attention core on the f32 MFMA; inside 1e-3 of the reference end to end)",
    "bf16x3h": "bf16x3h (bf16x3 with the attention core on IEEE-half operands, fp32 sums / softmax / outputs; logits inside 1e-3 of "
               "the reference, arg-max exact, gradients to 1.6e-3)",
    "fp32": "fp32 (every nn.Linear product as bf16x6 on the bf16 MFMA = fp32-accurate; attention core on the f32 MFMA)"}
VIT_B = dict(decoder="classification", image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12,
             mlp_dim=3072, dropout=0.0, emb_dropout=0.0)


def _lib_digest16():
    """First 16 hex digits of the source digest the loaded library was built from (myrtle_vision/hip/build.py)."""
    try:
        with open(os.path.join(ROOT, "myrtle-vision_amd", "lib", "build.sha256")) as f:
            return f.read().strip()[:16]
    except OSError:
        return None


def _latest_profile(suffix):
    """Newest committed profiles/rNN_<suffix> (rounds sort lexicographically) -> (parsed, file name, refusal reason or None).
    A counter summary is quoted only for the library it was measured on: its ``lib_source_digest16`` (or its evidence set's
    ``<tag>_stamp.json``) must equal the digest of the library this process loaded -- a kernel change without a refreshed
    evidence set reports ``null`` and says why instead of stale counters."""
    import glob
    # rNN_<suffix> or rNN_<tag>_<suffix> (tools/evidence.sh names its outputs <tag>_<suffix>, e.g. r03_b_pmc_traffic.json)
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]_*{suffix}")))
    if not files:
        return None, None, "no committed counter summary"
    name = os.path.basename(files[-1])
    try:
        with open(files[-1]) as f:
            d = json.load(f)
    except (OSError, ValueError) as e:
        return None, name, f"unreadable: {e}"
    measured = d.get("lib_source_digest16")
    if measured is None:
        try:
            with open(files[-1][:-len(suffix)] + "stamp.json") as f:
                measured = json.load(f).get("lib_source_digest16")
        except (OSError, ValueError):
            measured = None
    loaded = _lib_digest16()
    if measured is None or loaded is None or measured != loaded:
        return None, name, (f"profiles/{name} was measured on library source digest {measured}, this process loaded {loaded}: "
                            "counters not quoted (re-run tools/evidence.sh on this tree)")
    return d, name, None


def pmc_traffic(kernel_family):
    """HBM-side bytes per launch of the dominant kernel family, from the committed rocprofv3 PMC passes
    (FETCH_SIZE and WRITE_SIZE in separate runs, FETCH_SIZE doubled per the gfx950 correction of
    MI355X_MICROARCH.md section HBM; tools/pmc_traffic.py -> profiles/rNN_pmc_traffic.json).  Counters cannot be read from
    inside this process, so this is the last measured value for the same command ON THE SAME LIBRARY, else None + the reason."""
    d, name, why = _latest_profile("pmc_traffic.json")
    if d is None:
        return None, {"refused": why}
    try:
        return round(d["kernels"][kernel_family]["hbm_bytes_per_launch"]), {"source": f"profiles/{name}", "commit": d.get("commit"),
                                                                             "lib_source_digest16": _lib_digest16()}
    except (TypeError, KeyError, ValueError) as e:
        return None, {"refused": f"profiles/{name}: {e!r}"}


def pmc_mfma_util():
    """MFMA utilisation (matrix-pipe busy cycles / SIMD-cycles) from the committed SQ counter pass
    (tools/pmc_mfma_util.py -> profiles/rNN_mfma_util.json): the dominant kernel family and the attention + MLP blocks."""
    d, name, why = _latest_profile("mfma_util.json")
    if d is None:
        return {"refused": why}
    try:
        return {"gemm_nt": round(d["kernels"]["gemm_nt"]["mfma_util"], 4), "gemm_tn": round(d["kernels"]["gemm_tn"]["mfma_util"], 4),
                "attention": round(d["kernels"]["attention"]["mfma_util"], 4),
                "attention_mlp_block": round(d["block"]["mfma_util"], 4), "source": f"profiles/{name}",
                "commit": d.get("commit"), "lib_source_digest16": _lib_digest16()}
    except (TypeError, KeyError, ValueError) as e:
        return {"refused": f"profiles/{name}: {e!r}"}


def tolerance_mode_line(cfg, batch, dev, img, labels, precision="bf16x3h", steps=6, warmup=2):
    """The training step of the headline workload in ``precision`` (default: bf16x3h -- every nn.Linear product from two bf16 pieces
    per operand, attention on IEEE-half operands; logits <= 1.6e-4 of the reference, arg-max exact, gradients <= 5e-4): img/s."""
    import torch
    from myrtle_vision.hip.functional import cross_entropy
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.optim import AdamW, ParamArena
    vit = ViT(precision=precision, q_format="FP32", **cfg).to(dev)
    opt = AdamW(ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names()), lr=6.25e-5, betas=(0.9, 0.999), eps=1e-8,
                weight_decay=0.05)
    vit.train()

    def step():
        opt.zero_grad()
        cross_entropy(vit(img), labels).backward()
        opt.step()

    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"precision": precision, "value": round(batch / ms * 1e3, 1), "unit": "images/sec", "ms_per_step": round(ms, 3),
            "steps": steps, "warmup": warmup,
            "parity": "logits within 1e-3 of the reference (measured <= 1.6e-4), bit-exact arg-max, gradients within 1e-3 "
                      "(tests/test_vit_parity.py, profiles/r04_parity_measured.txt)"}


def cpu_baseline(seconds_budget=25.0):
    """Reference-equivalent CPU path (the oracle) on the host cores: ViT-B/16 B=8 fp32 fwd+bwd."""
    from oracle.detinit import det_state_dict
    from oracle.vit_oracle import ViTConfig, loss_and_grads
    # the GPU box exposes every host cpu but a 1-GPU job owns a 16-core share: oversubscribing 256 threads made this
    # 200x slower.  Use the affinity mask, capped at 16.
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    torch.set_num_threads(max(1, min(ncores, 16)))
    cfg = ViTConfig(decoder="classification", image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12,
                    heads=12, mlp_dim=3072)
    params = det_state_dict(cfg.param_shapes())
    g = torch.Generator().manual_seed(1234)
    img = torch.randn(8, 3, 224, 224, generator=g)
    labels = torch.randint(0, 1000, (8,), generator=g)
    loss_and_grads(params, img, labels, cfg)            # warm-up
    times = []
    t_start = time.perf_counter()
    while len(times) < 5 and (time.perf_counter() - t_start) < seconds_budget:
        t0 = time.perf_counter()
        loss_and_grads(params, img, labels, cfg)
        times.append(time.perf_counter() - t0)
    times.sort()
    med = times[len(times) // 2]
    return {"value": round(8.0 / med, 3), "unit": "images/sec", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"ViT-B/16 224^2 fp32 fwd+bwd, batch 8, median of {len(times)} after 1 warm-up "
                      f"(oracle/vit_oracle.py; {torch.get_num_threads()}-thread affinity share of a {os.cpu_count()}-cpu host)"}


def self_launch(n_gpus):
    """``python bench.py --gpus N`` started plainly (no WORLD_SIZE): become the launcher, as the reference scripts do with
    mp.spawn (classification/train.py:349-356).  N fresh children are started through torch.distributed.run BEFORE this
    process touches the GPU (nothing here has: no HIP call, no torch.cuda.is_available()), this process only waits and
    passes on their exit code -- a failed rank fails the command, and no GPU-initialised process is ever re-exec'd."""
    import socket
    import subprocess
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")           # dmabuf IPC: what RCCL needs on this host driver
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr",
           "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.run(cmd, env=env).returncode


def set_host_threads(local_world):
    """N ranks on one node share its host cores: each rank's intra-op pool is its share of the affinity mask (torch's default
    -- every visible cpu per process -- oversubscribes an 8-rank launch N-fold; torchrun would pin it to 1 instead, which
    starves rank 0's cpu_baseline leg at N = 1).  The HIP path itself needs one host thread per rank."""
    try:
        ncores = len(os.sched_getaffinity(0))
    except AttributeError:
        ncores = os.cpu_count() or 1
    n = max(1, min(16, ncores // max(local_world, 1)))
    if local_world > 1 or "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(n)
    return n


def dry_run(args, rank, world):
    """``--dry-run``: the launch / rendezvous / barrier / max-over-ranks / JSON plumbing with NO compute (a sleep stands in
    for the step), runnable without a GPU (gloo): what the CPU test of the multi-process path exercises.  Its JSON line
    says so and carries no throughput."""
    backend = os.environ.get("MV_DIST_BACKEND", "gloo")
    if args.dry_run_fail_rank is not None and rank == args.dry_run_fail_rank:
        # rehearsal of a rank that dies before the rendezvous completes: the launcher must fail the command, not hang
        raise SystemExit(f"rank {rank}: --dry-run-fail-rank")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo" if backend != "nccl" or not torch.cuda.is_available() else "nccl", rank=rank,
                                world_size=world)
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(0.001 * (1 + rank))
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    if rank == 0:
        print(json.dumps({"metric": METRIC[args.workload], "value": None, "unit": "images/sec", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * float(t) / args.steps, 3),
                          "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision,
                          "data": "dry-run: no compute, launch and collective plumbing only",
                          "config": {"workload": "dry-run", "global_batch": args.batch * world, "parallelism": f"dp{world}",
                                     "host_threads_per_rank": torch.get_num_threads()}}))
    if world > 1:
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=None,
                    help="per-GPU batch (default 256; 1024 for infer-int8: BASELINE config 5)")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--workload", default="cls", choices=["cls", "seg", "seg256", "infer-int8"],
                    help="cls = the headline (SURVEY section 8d config 2/3); seg/seg256 = config 4 (segmentation "
                         "decoder, 17 classes, 224^2 / 256^2); infer-int8 = config 5 (forward only, per-tensor "
                         "affine fake-quant at the Q8 sites after a 10-batch min/max calibration)")
    ap.add_argument("--seg-unfused", action="store_true", help="seg: vit(img) + cross_entropy instead of the fused tail")
    ap.add_argument("--no-optimizer", action="store_true", help="fwd+bwd only (section 8d: report with and without)")
    ap.add_argument("--int8-bf16-attention", action="store_true",
                    help="infer-int8: ViT.convert(bf16_attention=True) -- fused bf16 attention core instead of exact fp32")
    ap.add_argument("--q-format", default=None, choices=["FP16_16", "FP16_32", "TF32"],
                    help="cls workload with a fake-quantised model (runs in fp32 precision by definition; the forward "
                         "products of the FP16 formats go to the f16 matrix cores)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--per-shape", action="store_true", help="add the event-timed GEMM launches split by product shape")
    ap.add_argument("--no-kernel-timer", action="store_true")
    ap.add_argument("--timer-every", type=int, default=0,
                    help="bracket the GEMM launches with events on every k-th timed step only; 0 (default) = on the first and the "
                         "middle step of the timed region (the first only below 16 steps): a step whose ~150 GEMM launches are "
                         "bracketed runs 1.4 ms (4 %%) longer, idle time around ~300 event packets (rocprofv3 trace, round 4)")
    ap.add_argument("--graph", action="store_true",
                    help="capture the training step in ONE HIP graph and replay it (utils/graph.py): single GPU, static shapes; "
                         "the event timer is off in this mode, so the line carries no per-kernel roofline section")
    ap.add_argument("--prune-dead-tokens", action="store_true",
                    help="cls workload: ViT(prune_dead_tokens=True) -- the last block's FeedForward on the cls rows only (what the "
                         "classification decoder reads); identical logits / loss / gradients, 5.5 GFLOP per image less.  Off in "
                         "the headline line, which computes every token of every block as the reference does")
    ap.add_argument("--dry-run", action="store_true",
                    help="no compute: only the multi-process launch, rendezvous, barriers and the JSON line (CPU test)")
    ap.add_argument("--dry-run-fail-rank", type=int, default=None,
                    help="--dry-run only: this rank exits non-zero before the rendezvous (the launcher must fail, not hang)")
    args = ap.parse_args()

    if args.batch is None:
        args.batch = 1024 if args.workload == "infer-int8" else 256
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args.gpus))         # one fresh process per GPU; this one never initialises a GPU
    set_host_threads(int(os.environ.get("LOCAL_WORLD_SIZE", os.environ.get("WORLD_SIZE", "1"))))
    if args.dry_run:
        return dry_run(args, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if os.environ.get("MV_DIST_BACKEND", "nccl") != "nccl":
        local_rank = 0                                         # rehearsal: every rank shares GPU 0
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus} (one process per GPU)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    force_dist = world == 1 and os.environ.get("MV_FORCE_DIST") == "1"   # rehearsal: the RCCL code path with one rank
    if force_dist:
        os.environ.setdefault("MASTER_PORT", "29533")
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("MV_DIST_BACKEND", "nccl")     # "nccl" = RCCL; "gloo" only to rehearse N ranks on one GPU
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
        if not force_dist and dist.get_world_size() != args.gpus:      # every rank checks: one process per GPU, all of them here
            raise SystemExit(f"rank {rank}: process group has {dist.get_world_size()} ranks, --gpus says {args.gpus}")

    from myrtle_vision.hip import ops
    from myrtle_vision.hip.functional import cross_entropy
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.ddp import GradAllReducer, broadcast_parameters, exchange_dtype_from_env
    from myrtle_vision.utils.optim import AdamW, ParamArena
    from myrtle_vision.utils.utils import seed_everything

    seed_everything(1234)                                   # same initial weights on every rank
    cfg = dict(VIT_B)
    size = 224
    if args.workload in ("seg", "seg256"):
        size = 256 if args.workload == "seg256" else 224
        cfg.update(decoder="segmentation", num_classes=17, image_size=size)
    q_format = "PyTorchINT8" if args.workload == "infer-int8" else (args.q_format or "FP32")
    if args.prune_dead_tokens and args.workload != "cls":
        raise SystemExit("--prune-dead-tokens: classification workload only (the segmentation decoder reads every patch token)")
    vit = ViT(precision=args.precision, q_format=q_format, prune_dead_tokens=args.prune_dead_tokens, **cfg).to(dev)
    arena = ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names())
    opt = AdamW(arena, lr=6.25e-5, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.05)      # vit_base.json
    reducer = GradAllReducer(arena, exchange_dtype=exchange_dtype_from_env(), measure=True)    # MV_DDP_EXCHANGE=bf16: opt-in
    if force_dist:
        reducer.enabled = True                                  # launch the bucketed all-reduces although world == 1
        dist.broadcast(arena.flat_param, src=0)
    broadcast_parameters(arena)
    opt.grad_scale = reducer.grad_scale

    g = torch.Generator().manual_seed(1234 + rank)          # per-rank shard of the synthetic global batch
    img = torch.randn(args.batch, 3, size, size, generator=g).to(dev)
    if args.workload in ("seg", "seg256"):
        labels = torch.randint(0, 17, (args.batch, size, size), generator=g).to(dev)
    else:
        labels = torch.randint(0, 1000, (args.batch,), generator=g).to(dev)
    vit.train()

    if args.workload == "infer-int8":
        vit.eval()
        with torch.no_grad():
            for i in range(10):                                  # min/max calibration (test_quantize.py:26-34)
                vit(torch.randn(64, 3, size, size, generator=g).to(dev))
        vit.convert(bf16_attention=args.int8_bf16_attention)

        def step():
            with torch.no_grad():
                return vit(img).float().mean()
    else:
        def step():
            opt.zero_grad()
            if args.workload in ("seg", "seg256") and not args.seg_unfused:
                loss = vit.segmentation_loss(img, labels)[0]     # what segmentation/train.py runs here (engine.py)
            else:
                loss = cross_entropy(vit(img), labels)
            loss.backward()
            reducer.finish()
            if not args.no_optimizer:
                opt.step()
            return loss

    if args.graph:
        if args.workload == "infer-int8" or args.no_optimizer or world > 1 or force_dist:
            raise SystemExit("--graph: single-GPU training workloads with the optimizer step")
        from myrtle_vision.utils.graph import GraphedTrainStep
        seg_fused = args.workload in ("seg", "seg256") and not args.seg_unfused
        loss_fn = (lambda m, x, y: m.segmentation_loss(x, y)[0]) if seg_fused else (lambda m, x, y: cross_entropy(m(x), y))
        graphed = GraphedTrainStep(vit, opt, loss_fn, img, labels)

        def step():                                          # noqa: F811  (replaces the eager step)
            return graphed(img, labels)
        args.no_kernel_timer = True
    for _ in range(args.warmup):
        loss = step()
    reducer.exposed_ms(reset=True)                            # drop the warm-up steps' spans (lazy RCCL / communicator setup)
    timed_steps = (sorted({0, args.steps // 2}) if args.steps >= 16 else [0]) if args.timer_every <= 0 else \
        list(range(0, args.steps, args.timer_every))
    timer = None if args.no_kernel_timer else ops.KernelTimer(sample_steps=timed_steps)
    if world > 1 or force_dist:
        dist.barrier()
    torch.cuda.synchronize()
    ops.set_kernel_timer(timer)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if timer is not None:
            timer.next_step()
        loss = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    ops.set_kernel_timer(None)
    final_loss = float(loss.detach())
    exposed_ms = reducer.exposed_ms()                           # per rank; the slowest rank's is reported
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t)
        if exposed_ms is not None:
            t = torch.tensor([exposed_ms], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            exposed_ms = float(t)

    if rank == 0:
        img_s = args.batch * world * args.steps / elapsed
        out = {
            "metric": METRIC[args.workload],
            "value": round(img_s, 2), "unit": "images/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(1e3 * elapsed / args.steps, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,
            "dtype": ("int8 (v_mfma_i32_16x16x64_i8, int32 accumulate; attention core "
                      + ("bf16" if args.int8_bf16_attention else "fp32 on the f32 MFMA") + ")")
            if args.workload == "infer-int8" else (f"fp32 values fake-quantised to {args.q_format}" if args.q_format
                                                   else PRECISION_DTYPE.get(args.precision, args.precision)),
            "data": "synthetic",
            "config": {"workload": WORKLOAD[args.workload] + ("" if not args.no_optimizer else " [optimizer step OFF]")
                       + (f" [q_format {args.q_format}]" if args.q_format else "")
                       + (" [attention core: fused bf16 kernel]" if args.int8_bf16_attention else "")
                       + (" [prune_dead_tokens: last block's FeedForward on the cls rows only; 99.83 GFLOP/img executed]"
                          if args.prune_dead_tokens else "")
                       + (" [activations saved for backward in bf16; gelu'(h) as an 8-bit code]"
                          if args.precision == "bf16" and args.workload != "infer-int8" and not args.q_format else ""),
                       "step_launch": "one HIP graph replay per step" if args.graph else "eager (one launch per kernel)",
                       # what the bf16 mode stores between forward and backward (hip/functional.py): not a precision claim
                       **({"saved_activations": "bf16, except gelu'(h): an 8-bit code on a fixed grid of step 0.005 holding 0 and 1 "
                                                "exactly (|error| <= 0.0025), written by fc1's epilogue and multiplied into dX of fc2"}
                          if args.precision == "bf16" and args.workload != "infer-int8" and not args.q_format else {}),
                       "per_gpu_batch": args.batch, "global_batch": args.batch * world,
                       "parallelism": f"dp{world}", "final_loss": round(final_loss, 4)},
        }
        if world > 1 or force_dist:
            # what the exchange looked like, so a scaling run explains itself: the collective library and
            # its version, what one step puts on the links, and how long the GPU waited for the exchange after backward's
            # last kernel (events on the compute stream around GradAllReducer.finish(), mean over the timed steps, max over ranks)
            try:
                rccl = ".".join(str(v) for v in torch.cuda.nccl.version())
            except Exception:
                rccl = None
            out["dist"] = {"backend": dist.get_backend(), "world": dist.get_world_size(), "rccl_version": rccl,
                           **reducer.describe(),
                           "allreduce_exposed_ms": None if exposed_ms is None else round(exposed_ms, 3),
                           "rehearsal_one_rank": bool(force_dist),
                           "env": {k: os.environ[k] for k in ("NCCL_MAX_NCHANNELS", "NCCL_MIN_NCHANNELS", "NCCL_ALGO", "NCCL_PROTO",
                                                               "MV_DDP_EXCHANGE") if k in os.environ}}
        if args.workload == "cls":
            gflop = TRAIN_GFLOP_PER_IMG - (5.55 if args.prune_dead_tokens else 0.0)      # FLOPs actually executed per image
            out["step_mfma_frac"] = round(img_s / world * gflop * 1e9 / (BF16_DENSE_PEAK_TFLOPS * 1e12), 4)
        if timer is not None:
            summ = timer.summary()
            k = summ.get("gemm_nt_bf16")
            traffic, traffic_src = pmc_traffic("gemm_nt")
            kx = summ.get("gemm_nt_bf16x3") or summ.get("gemm_nt_bf16x6")
            if kx and not k:
                # the split-operand modes: the same NT kernel over a 3 K (6 K) contraction; the timer counts the fp32 product's
                # 2 M N K, the matrix cores execute nseg times that
                nseg = 3 if "gemm_nt_bf16x3" in summ else 6
                out["roofline"] = {"bound": "mfma", "kernel": f"gemm_nt_8phase_kernel family over the {nseg} K contraction of bf16x{nseg}",
                                   "achieved": round(nseg * kx["tflops"], 1), "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(nseg * kx["tflops"] / BF16_DENSE_PEAK_TFLOPS, 4), "traffic": None,
                                   "fp32_equivalent_tflops": round(kx["tflops"], 1), "launches": kx["launches"],
                                   "avg_launch_us": round(kx["avg_us"], 1)}
            if k:
                out["roofline"] = {"bound": "mfma", "kernel": "gemm_nt_8phase_kernel family (mv_gemm_nt_bf16)",
                                   "achieved": round(k["tflops"], 1), "peak": BF16_DENSE_PEAK_TFLOPS, "unit": "TFLOP/s",
                                   "frac": round(k["tflops"] / BF16_DENSE_PEAK_TFLOPS, 4), "traffic": traffic,
                                   "traffic_source": traffic_src,
                                   "launches": k["launches"], "timed_steps": f"steps {timed_steps} of {args.steps}",
                                   "avg_launch_us": round(k["avg_us"], 1),
                                   "gflop_per_launch": round(k["flops_per_launch"] / 1e9, 2),
                                   # SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), separate PMC pass
                                   "mfma_util": pmc_mfma_util(),
                                   # measured on this board class (profiles/r01_clock_power.md): what the 1400 W cap lets
                                   # an MFMA stream on random bf16 operands sustain -- context for frac, not its denominator
                                   "power_capped_tflops": {"mfma_from_registers": 1930, "mfma_fed_from_lds": 1600}}
            nt_steps = len(timed_steps)
            out["kernels"] = {n: {"launches": v["launches"], "avg_us": round(v["avg_us"], 1), "tflops": round(v["tflops"], 1),
                                  "ms_per_step": round(v["total_ms"] / nt_steps, 3)} for n, v in summ.items()}
            if args.per_shape:
                out["per_shape"] = {k: {"launches": v["launches"], "avg_us": round(v["avg_us"], 1), "tflops": round(v["tflops"], 1)}
                                    for k, v in sorted(timer.shape_summary().items())}
        if os.environ.get("MV_COMMIT"):
            out["commit"] = os.environ["MV_COMMIT"]              # set by tools/evidence.sh: which tree this line was measured on
        if (world == 1 and not force_dist and not args.no_cpu_baseline and args.workload == "cls" and args.precision == "bf16"
                and not args.q_format and not args.no_optimizer and not args.prune_dead_tokens):
            # next to the benchmarked arithmetic (logits 5-8e-3 of the reference): the SAME step in the fastest arithmetic that meets
            # north_star's 1e-3 / exact arg-max (tests/test_vit_parity.py), measured in this process after the timed region
            ops.set_kernel_timer(None)
            del vit, opt, arena, reducer
            torch.cuda.empty_cache()
            try:
                out["within_tolerance"] = tolerance_mode_line(cfg, args.batch, dev, img, labels)
            except Exception as e:                               # a secondary figure must never cost the headline line
                out["within_tolerance"] = {"error": f"{type(e).__name__}: {e}"[:300]}
        if world == 1 and not args.no_cpu_baseline and args.workload == "cls":
            try:
                out["cpu_baseline"] = cpu_baseline()
            except Exception as e:                               # likewise: report the failure, keep the line
                out["cpu_baseline"] = {"value": None, "error": f"{type(e).__name__}: {e}"[:300]}
        print(json.dumps(out))
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
