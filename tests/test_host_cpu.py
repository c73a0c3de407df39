"""CPU tests (no GPU): the C ABI loads and exports every symbol the header declares, host logic mirrors the
reference's semantics, checkpoints keep the reference wire format, quantiser plumbing renames keys like the
reference, and the N > 1 gradient exchange is correct with world_size 2 over gloo."""
import json
import math
import os
import re
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT, load_golden

HEADER = os.path.join(ROOT, "include", "myrtle_vision_hip.h")


# ---------------------------------------------------------------- C ABI
def _parse_header():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    decls = {}
    for m in re.finditer(r"\b(int|long|size_t|const char\*)\s+(mv_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        ret, name, args = m.group(1), m.group(2), m.group(3).strip()
        kinds = ""
        if args and args != "void":
            for a in args.split(","):
                a = " ".join(a.split())
                if "*" in a or "mv_stream_t" in a:
                    kinds += "p"
                elif a.startswith("size_t"):
                    kinds += "z"
                elif a.startswith("long"):
                    kinds += "l"
                elif a.startswith("float"):
                    kinds += "f"
                elif a.startswith("int"):
                    kinds += "i"
                elif a.startswith("uint64_t"):
                    kinds += "Q"
                else:
                    raise AssertionError(f"unparsed argument {a!r} in {name}")
        decls[name] = (kinds, ret)
    return decls


def test_library_exports_every_declared_symbol_with_matching_signature():
    from myrtle_vision.hip import lib
    decls = _parse_header()
    assert len(decls) >= 34
    assert set(decls) == set(lib.SIGNATURES), set(decls) ^ set(lib.SIGNATURES)
    for name, (kinds, _) in decls.items():
        assert lib.SIGNATURES[name][0] == kinds, (name, kinds, lib.SIGNATURES[name][0])
    handle = lib.lib()                      # loads the .so on a machine without a GPU; binds every symbol
    assert handle.mv_version() >= 100
    assert handle.mv_seg_ce_partials(256, 224, 224) == 256 * 49 and handle.mv_seg_ce_partials(0, 224, 224) == 0
    assert b"aligned" in handle.mv_error_string(-2)
    assert handle.mv_gemm_tn_workspace_bytes(768, 768, 50432) > 0
    assert handle.mv_layernorm_bwd_workspace_bytes(50432, 768) == 1024 * 3 * 768 * 4


def test_missing_library_fails_loudly(tmp_path, monkeypatch):
    from myrtle_vision.hip import lib
    monkeypatch.setattr(lib, "_lib", None)
    monkeypatch.setattr(lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(lib.HipLibraryMissing, match="no CPU"):
        lib.lib()


def test_cpu_forward_fails_loudly():
    from myrtle_vision.models.vit import ViT
    vit = ViT(decoder="classification", image_size=224, patch_size=16, num_classes=5, dim=64, depth=1, heads=1, mlp_dim=64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vit(torch.randn(1, 3, 224, 224))


# ---------------------------------------------------------------- model class / state dict
@pytest.mark.parametrize("name", ["micro_cls", "micro_seg", "micro_seg_256", "base_cls", "base_seg"])
def test_state_dict_matches_reference(name):
    from myrtle_vision.models.vit import ViT
    _, meta = load_golden(name)
    vit = ViT(patch_size=16, q_format="FP32", **meta["kwargs"])
    sd = vit.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == meta["param_shapes"]
    assert list(sd.keys()) == meta["state_keys"]              # the reference's own key ORDER (recorded as a list)
    assert set(vit.unused_parameter_names()) == set(meta["unused_params"])


@pytest.mark.parametrize("name,fmt", [("micro_cls_fp16_32", "FP16_32"), ("micro_cls_tf32", "TF32"),
                                      ("micro_cls_fp16_16", "FP16_16"), ("micro_seg_fp16_32", "FP16_32")])
def test_prepared_state_dict_keys_match_reference(name, fmt):
    """prepare_qat wraps modules as Sequential(QuantStub, module): keys gain the same '.1.' as the reference's, in the
    same order (why the reference has --quantized_ckpt)."""
    from myrtle_vision.models.vit import ViT
    _, meta = load_golden(name)
    vit = ViT(patch_size=16, q_format=fmt, **meta["kwargs"])
    assert list(vit.state_dict().keys()) == meta["state_keys_prepared"]


def test_constructor_asserts_like_reference():
    from myrtle_vision.models.vit import ViT
    base = dict(decoder="classification", image_size=224, patch_size=16, num_classes=5, dim=64, depth=1, heads=1, mlp_dim=64)
    with pytest.raises(AssertionError, match="divisible by the patch size"):
        ViT(**{**base, "image_size": 225})
    with pytest.raises(AssertionError, match="way too small"):
        ViT(**{**base, "image_size": 64})
    with pytest.raises(AssertionError, match="decoder must be"):
        ViT(**{**base, "decoder": "foo"})


def test_quant_prepare_renames_keys_like_reference():
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.quantize import QFormat
    _, meta = load_golden("micro_cls_fp16_32_conv")
    vit = ViT(patch_size=16, q_format="FP32", **meta["kwargs"])
    vit.quantizer.prepare_qat("FP16_32")
    assert list(vit.state_dict().keys()) == meta["state_keys_after_convert"]
    assert vit.precision == "fp32" and vit.quantizer.q_format == QFormat.FP16_32
    with pytest.raises(ValueError, match="already quantized"):
        vit.quantizer.prepare_qat("TF32")


# ---------------------------------------------------------------- utils
def test_get_batch_sizes_reference_semantics():
    from myrtle_vision.utils.utils import get_batch_sizes
    assert get_batch_sizes(32, 2, 64) == (32, 1)            # vit_base.json on 2 GPUs
    assert get_batch_sizes(32, 1, 64) == (32, 2)
    assert get_batch_sizes(32, 0, 64) == (32, 2)            # CPU: num_gpus 0 treated as one worker
    assert get_batch_sizes(256, 8, 2048) == (256, 1)
    assert get_batch_sizes(32, 8, 2048) == (32, 8)
    assert get_batch_sizes(48, 2, 64) == (32, 1)            # falls back to the largest divisor below the target
    assert get_batch_sizes(7, 2, 20) == (5, 2)
    with pytest.raises(ValueError, match="not divisible by the number of GPUs"):
        get_batch_sizes(32, 3, 64)


def test_optimizer_args_and_schedule(tmp_path):
    from myrtle_vision.utils.models import get_optimizer_args
    from myrtle_vision.utils.optim import CosineLRScheduler
    from oracle.optim_oracle import cosine_lr
    cfg = json.load(open(os.path.join(ROOT, "classification", "train_configs", "vit_base.json")))["train_config"]
    a = get_optimizer_args(cfg)
    assert (a.opt, a.sched, a.lr, a.weight_decay, a.warmup_epochs, a.min_lr) == ("adamw", "cosine", 6.25e-5, 0.05, 5, 1e-5)

    class FakeOpt:
        param_groups = [{"lr": a.lr, "initial_lr": a.lr}, {"lr": a.lr, "initial_lr": a.lr}]
    sched = CosineLRScheduler(FakeOpt, t_initial=a.epochs, lr_min=a.min_lr, warmup_t=a.warmup_epochs, warmup_lr_init=a.warmup_lr)
    assert FakeOpt.param_groups[0]["lr"] == a.warmup_lr          # timm sets warmup_lr_init at construction
    for epoch in [0, 1, 4, 5, 6, 100, 299, 300, 400]:
        sched.step(epoch)
        want = cosine_lr(epoch, base_lr=a.lr, t_initial=a.epochs, lr_min=a.min_lr, warmup_t=a.warmup_epochs, warmup_lr_init=a.warmup_lr)
        assert math.isclose(FakeOpt.param_groups[1]["lr"], want, rel_tol=1e-12), epoch


def test_param_arena_groups_and_views():
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.optim import ParamArena
    from oracle.optim_oracle import timm_param_groups
    vit = ViT(decoder="classification", image_size=224, patch_size=16, num_classes=5, dim=64, depth=2, heads=1, mlp_dim=128)
    before = {k: v.clone() for k, v in vit.state_dict().items()}
    arena = ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names())
    ref = timm_param_groups([(n, p) for n, p in vit.named_parameters() if n not in vit.unused_parameter_names()], 0.05)
    n_no_decay = sum(p.numel() for p in ref[0]["params"])
    n_decay = sum(p.numel() for p in ref[1]["params"])
    assert arena.n_decay >= n_decay and arena.total - arena.n_decay >= n_no_decay
    assert "pos_embedding" in arena.names[: len(ref[1]["params"])]         # ViT has no no_weight_decay(): pos/cls ARE decayed
    assert "pos_embedding_det" not in arena.names and "det_tokens" not in arena.names
    for k, v in vit.state_dict().items():
        assert torch.equal(v, before[k])                                   # flattening preserved every value
    from myrtle_vision.hip import ops
    for j, (p, o) in enumerate(zip(arena.params, arena.offsets)):
        assert p.data_ptr() == arena.flat_param.data_ptr() + 4 * o and o % 4 == 0
        # no gradient attached: the backward kernels get the arena slot itself as their output (ops.grad_out)
        assert p.grad is None and arena.slot(j).data_ptr() == arena.flat_grad.data_ptr() + 4 * o
        assert ops.grad_out(p, p.shape, p.device).data_ptr() == arena.slot(j).data_ptr()
    # fallback path: a gradient autograd materialised elsewhere is moved into its slot, a missing one zero-fills it
    arena.flat_grad.fill_(1.0)
    p0 = arena.params[0]
    p0.grad = torch.full_like(p0, 2.0)
    assert ops.grad_out(p0, p0.shape, p0.device).data_ptr() != arena.slot(0).data_ptr()    # accumulation: fresh tensor
    arena.sync_grads()
    assert p0.grad.data_ptr() == arena.slot(0).data_ptr() and float(arena.slot(0).min()) == 2.0
    assert sum(float(arena.slot(j).abs().sum()) for j in range(1, len(arena.params))) == 0.0   # (alignment gaps are in no slot)
    arena.zero_grad()
    assert all(p.grad is None for p in arena.params)


def test_checkpoint_wire_format_roundtrip(tmp_path):
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.models import load_checkpoint, save_checkpoint
    from myrtle_vision.utils.optim import AdamW, CosineLRScheduler, ParamArena
    kw = dict(decoder="classification", image_size=224, patch_size=16, num_classes=5, dim=64, depth=1, heads=1, mlp_dim=64)
    vit = ViT(**kw)
    opt = AdamW(ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names()), lr=1e-3, weight_decay=0.05)
    opt.exp_avg.normal_()
    opt.step_count = 7
    sched = CosineLRScheduler(opt, t_initial=10, lr_min=1e-5, warmup_t=2, warmup_lr_init=1e-6)
    sched.step(3)
    path = str(tmp_path / "vit_000123")
    save_checkpoint(vit, opt, sched, 123, path)
    ckpt = torch.load(path, map_location="cpu", weights_only=False)
    assert set(ckpt) == {"model", "optimizer", "lr_scheduler", "iteration"}          # reference utils/models.py:120-126
    assert list(ckpt["model"].keys()) == list(vit.state_dict().keys())
    vit2 = ViT(**kw)
    opt2 = AdamW(ParamArena(vit2.named_parameters(), skip=vit2.unused_parameter_names()), lr=1e-3, weight_decay=0.05)
    sched2 = CosineLRScheduler(opt2, t_initial=10, lr_min=1e-5, warmup_t=2, warmup_lr_init=1e-6)
    assert load_checkpoint(vit2, opt2, sched2, path) == 123
    for (k, a), (_, b) in zip(vit.state_dict().items(), vit2.state_dict().items()):
        assert torch.equal(a, b), k
    s1, s2 = opt.state_dict()["state"], opt2.state_dict()["state"]
    assert all(torch.equal(s1[k]["exp_avg"], s2[k]["exp_avg"]) for k in s1) and opt2.step_count == 7   # (arena padding is not state)
    assert opt2.param_groups[0]["lr"] == opt.param_groups[0]["lr"]
    # a reference-produced "model" dict (plain tensors under the same keys) loads unchanged
    vit2.load_state_dict({k: v.clone() for k, v in ckpt["model"].items()})


def test_optimizer_state_interchanges_with_torch_adamw():
    """The optimizer part of a checkpoint is torch.optim.AdamW.state_dict() of the optimizer the reference builds through
    timm (no-decay group first, detection parameters included but stateless): both directions round-trip."""
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.optim import AdamW, ParamArena
    from oracle.optim_oracle import reference_adamw
    kw = dict(decoder="classification", image_size=224, patch_size=16, num_classes=5, dim=64, depth=1, heads=1, mlp_dim=64)
    torch.manual_seed(0)
    ref_model = ViT(**kw)
    ref_opt = reference_adamw(list(ref_model.named_parameters()), lr=1e-3, weight_decay=0.05)
    unused = set(ref_model.unused_parameter_names())
    for _ in range(2):
        for n, p in ref_model.named_parameters():
            p.grad = None if n in unused else torch.randn_like(p)
        ref_opt.step()
    sd = ref_opt.state_dict()                                  # what the reference's save_checkpoint stores
    names = [n for g in sd["param_groups"] for n in [None] * len(g["params"])]
    vit = ViT(**kw)
    opt = AdamW(ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names()), lr=1e-3, weight_decay=0.05)
    opt.load_state_dict(sd)
    assert opt.step_count == 2
    index, order = opt._torch_index()
    assert len(order) == len(names) and set(order) == {n for n, _ in ref_model.named_parameters()}
    ref_params = dict(ref_model.named_parameters())
    ref_index = {id(p): i for i, p in enumerate(q for g in ref_opt.param_groups for q in g["params"])}
    for n, p, o in zip(opt.arena.names, opt.arena.params, opt.arena.offsets):
        i = ref_index[id(ref_params[n])]
        assert i == index[n], n                                # same integer key as torch assigns
        assert torch.equal(opt.exp_avg[o:o + p.numel()].view(p.shape), sd["state"][i]["exp_avg"]), n
        assert torch.equal(opt.exp_avg_sq[o:o + p.numel()].view(p.shape), sd["state"][i]["exp_avg_sq"]), n
    # and back: torch loads what we write
    ours = opt.state_dict()
    assert [g["params"] for g in ours["param_groups"]] == [g["params"] for g in sd["param_groups"]]
    assert [g["weight_decay"] for g in ours["param_groups"]] == [0.0, 0.05]
    assert set(ours["state"]) == set(sd["state"])              # the unused detection parameters carry no state
    fresh = reference_adamw(list(ViT(**kw).named_parameters()), lr=1e-3, weight_decay=0.05)
    fresh.load_state_dict({"state": ours["state"], "param_groups": ours["param_groups"]})
    back = fresh.state_dict()["state"]
    for i, ent in sd["state"].items():
        assert torch.equal(back[i]["exp_avg"], ent["exp_avg"]) and float(back[i]["step"]) == float(ent["step"]) == 2.0


def test_rename_timm_state_dict_rules():
    from myrtle_vision.utils.models import apply_rules, rename_timm_state_dict
    D, depth = 32, 2
    timm = {"cls_token": torch.zeros(1, 1, D), "pos_embed": torch.zeros(1, 197, D),
            "patch_embed.proj.weight": torch.arange(D * 3 * 16 * 16, dtype=torch.float32).reshape(D, 3, 16, 16),
            "patch_embed.proj.bias": torch.zeros(D), "norm.weight": torch.ones(D), "norm.bias": torch.zeros(D),
            "head.weight": torch.zeros(10, D), "head.bias": torch.zeros(10)}
    for i in range(depth):
        for k, shape in [("norm1.weight", (D,)), ("norm1.bias", (D,)), ("attn.qkv.weight", (3 * D, D)), ("attn.qkv.bias", (3 * D,)),
                         ("attn.proj.weight", (D, D)), ("attn.proj.bias", (D,)), ("norm2.weight", (D,)), ("norm2.bias", (D,)),
                         ("mlp.fc1.weight", (4 * D, D)), ("mlp.fc1.bias", (4 * D,)), ("mlp.fc2.weight", (D, 4 * D)), ("mlp.fc2.bias", (D,))]:
            timm[f"blocks.{i}.{k}"] = torch.zeros(shape)
    out = rename_timm_state_dict(timm, {"embed_dim": D, "patch_size": 16}, 10)
    assert "head.weight" not in out and "norm.weight" not in out and "decoder.norm.weight" not in out
    assert out["patch_to_embedding.weight"].shape == (D, 768)
    w = timm["patch_embed.proj.weight"]
    assert out["patch_to_embedding.weight"][3, (5 * 16 + 7) * 3 + 2] == w[3, 2, 5, 7]      # (O,I,H,W) -> (O,(H,W,I))
    assert "transformer.layers.1.0.fn.fn.to_out.0.weight" in out and "transformer.layers.0.1.fn.fn.net.3.bias" in out
    assert apply_rules("pos_embed", [(r"pos_embed", r"pos_embedding")]) == "pos_embedding"
    with pytest.raises(FileNotFoundError, match="cannot be downloaded"):
        rename_timm_state_dict("vit_base_patch16_224", {"embed_dim": D, "patch_size": 16}, 10)


def test_rename_timm_matches_reference_fixture(tmp_path):
    """Row f4 pinned: ``tests/golden/timm_rename.*`` is what the REFERENCE's ``rename_timm_state_dict`` (utils/models.py:154-223,
    run by ``gen_golden.py`` on a formula-defined timm-keyed state dict) returned -- key order, which timm key each output
    came from, the dropped classifier head, the conv -> linear permutation of the patch embedding, every value."""
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.models import rename_timm_state_dict
    from oracle.detinit import det_param, summarize, timm_source_shapes
    arrays, meta = load_golden("timm_rename")
    cfg = meta["cfg"]
    src = {k: det_param("timm:" + k, s) for k, s in timm_source_shapes(cfg).items()}
    assert list(src) == meta["timm_keys"]
    path = str(tmp_path / "vit_nano_patch16_224.pth")
    torch.save(src, path)
    for source in (path, src):                                         # a local file (what engine.train_worker passes) or a dict
        out = rename_timm_state_dict(source, {"embed_dim": cfg["embed_dim"], "patch_size": cfg["patch_size"]}, cfg["num_classes"])
        assert list(out) == meta["renamed_keys"]                       # same keys, same ORDER
        assert not set(meta["dropped"]) & set(out)
        assert {k: list(v.shape) for k, v in out.items()} == meta["shapes"]
        assert np.array_equal(out["patch_to_embedding.weight"].numpy(), arrays["patch_to_embedding.weight"])
        for k, v in out.items():
            assert np.array_equal(summarize(v).numpy(), arrays["sum:" + k]), k
            if k != "patch_to_embedding.weight":
                assert torch.equal(v, src[meta["origin"][k]]), k
    vit = ViT(patch_size=cfg["patch_size"], q_format="FP32", decoder="classification", image_size=cfg["image_size"],
              num_classes=cfg["num_classes"], dim=cfg["embed_dim"], depth=cfg["depth"], heads=cfg["heads"], mlp_dim=cfg["mlp_dim"])
    res = vit.load_state_dict(out, strict=False)                       # segmentation/train.py:164-175
    assert res.unexpected_keys == [] and list(res.missing_keys) == meta["missing_keys_after_load"]


@pytest.mark.parametrize("n,world", [(10, 2), (11, 2), (1000, 8), (1003, 8), (7, 4), (3, 8), (1, 2), (64, 1)])
def test_shard_sampler_is_torch_distributed_sampler(n, world):
    """``engine.ShardSampler`` yields exactly ``DistributedSampler(dataset)``'s indices (classification/train.py:116: default
    arguments -> seed 0), for every rank and epoch, including the wrap-around padding and n < world."""
    from torch.utils.data import DistributedSampler
    from myrtle_vision.engine import ShardSampler
    ds = list(range(n))
    for epoch in (0, 1, 5):
        seen = []
        for rank in range(world):
            ref = DistributedSampler(ds, num_replicas=world, rank=rank)
            ours = ShardSampler(n, rank, world)
            ref.set_epoch(epoch)
            ours.set_epoch(epoch)
            assert list(ours) == list(ref) and len(ours) == len(ref)
            seen += list(ours)
        assert set(seen) == set(range(n))


def test_segment_scope_is_per_thread_and_follows_the_precision():
    """Round 4 host logic: the segment count of the split-operand modes (6 = bf16x6 / fp32, 3 = bf16x3, 4 = bf16x3 + half attention)
    is a per-THREAD scope -- forward runs on the caller's thread, backward on autograd's workers, DataLoader threads see the
    default -- that nests and restores, and every autograd function records the value its forward ran under."""
    import threading
    from myrtle_vision.hip import ops
    assert [ops.prec_segments(p) for p in ("fp32", "bf16", "bf16x3", "bf16x3h")] == [6, 6, 3, 4]
    assert all(ops.act_dtype(p) == torch.float32 for p in ("fp32", "bf16x3", "bf16x3h")) and ops.act_dtype("bf16") == torch.bfloat16
    with pytest.raises(ValueError):
        ops.act_dtype("fp16")
    assert ops.current_segments() == 6 and not ops.half_attention()
    seen = {}
    with ops.segments(4):
        assert ops.current_segments() == 3 and ops.half_attention()
        with ops.segments(6):
            assert ops.current_segments() == 6 and not ops.half_attention()
        t = threading.Thread(target=lambda: seen.update(other=(ops.current_segments(), ops.half_attention())))
        t.start(); t.join()
        assert ops.current_segments() == 3 and ops.half_attention()
    assert seen["other"] == (6, False) and ops.current_segments() == 6
    with pytest.raises(ValueError):
        ops.segments(5)
    # the pruning switch: classification only, off by default, env override
    from myrtle_vision.models.vit import ViT
    kw = dict(image_size=224, patch_size=16, num_classes=5, dim=64, depth=1, heads=1, mlp_dim=64)
    assert not ViT(decoder="classification", **kw).transformer.cls_only_tail
    assert ViT(decoder="classification", prune_dead_tokens=True, **kw).transformer.cls_only_tail
    assert not ViT(decoder="segmentation", prune_dead_tokens=True, **kw).transformer.cls_only_tail


def test_get_models_reads_reference_config_schema(tmp_path):
    from myrtle_vision.utils.models import get_models
    cfg = json.load(open(os.path.join(ROOT, "classification", "train_configs", "vit_tiny.json")))
    cfg["data_config_path"] = os.path.join(ROOT, "classification", cfg["data_config_path"])
    vit, distiller = get_models(cfg)
    assert distiller is None and sum(p.numel() for p in vit.parameters()) == 5571501       # SURVEY 8a (a1), Tiny, 45 classes


# ---------------------------------------------------------------- N > 1: gradient exchange over gloo
def _ddp_worker(rank, world, port, tmpdir, exchange=torch.float32):
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from myrtle_vision.utils.ddp import GradAllReducer, broadcast_parameters
    from myrtle_vision.utils.optim import ParamArena
    torch.manual_seed(100 + rank)                                   # deliberately different initial weights per rank
    model = torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.Tanh(), torch.nn.Linear(40, 24), torch.nn.Tanh(), torch.nn.Linear(24, 3))
    unused = torch.nn.Parameter(torch.randn(5))                     # a parameter that never gets a gradient
    named = list(model.named_parameters()) + [("unused", unused)]
    arena = ParamArena(named)
    broadcast_parameters(arena, src=0)                              # DDP's constructor broadcast
    red = GradAllReducer(arena, bucket_bytes=2048, exchange_dtype=exchange)   # several buckets
    assert len(red.ranges) > 2
    assert red.describe()["bytes"] == arena.total * (2 if exchange == torch.bfloat16 else 4)
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)
    xs, ys = X[rank::world], Y[rank::world]                         # DistributedSampler-style shard
    for step in range(3):
        arena.zero_grad()
        loss = ((model(xs) - ys) ** 2).mean()
        loss.backward()
        red.finish()
        if step == 0:
            torch.save({"grad": arena.flat_grad.clone() * red.grad_scale, "param": arena.flat_param.clone()},
                       os.path.join(tmpdir, f"r{rank}.pt"))
        with torch.no_grad():
            arena.flat_param -= 0.1 * red.grad_scale * arena.flat_grad
    torch.save(arena.flat_param.clone(), os.path.join(tmpdir, f"final{rank}.pt"))
    dist.destroy_process_group()


def test_gradient_allreduce_world2_gloo(tmp_path):
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["param"], r1["param"])                     # broadcast made the ranks identical
    assert torch.allclose(r0["grad"], r1["grad"], atol=0, rtol=0)    # all-reduce: same gradient everywhere
    # single-process gradient of the mean loss over the CONCATENATED batch equals the averaged rank gradients
    from myrtle_vision.utils.optim import ParamArena
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.Tanh(), torch.nn.Linear(40, 24), torch.nn.Tanh(), torch.nn.Linear(24, 3))
    unused = torch.nn.Parameter(torch.randn(5))
    arena = ParamArena(list(model.named_parameters()) + [("unused", unused)])
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)
    ((model(X) - Y) ** 2).mean().backward()
    arena.sync_grads()                                               # torch's own gradients -> arena slots; unused -> zeros
    assert torch.allclose(arena.flat_grad, r0["grad"], atol=1e-6, rtol=1e-5)
    f0, f1 = torch.load(tmp_path / "final0.pt"), torch.load(tmp_path / "final1.pt")
    assert torch.equal(f0, f1)                                       # identical parameters on all ranks after K steps


def test_gradient_allreduce_bf16_exchange_world2_gloo(tmp_path):
    """``exchange_dtype=torch.bfloat16`` (half the bytes on the links): every rank still ends with the SAME gradient and the
    same parameters; the gradient equals the fp32 exchange's to bf16 resolution (one rounding per rank + one bf16 sum)."""
    port = 31500 + (os.getpid() % 2000)
    mp.spawn(_ddp_worker, args=(2, port, str(tmp_path), torch.bfloat16), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["grad"], r1["grad"])
    f0, f1 = torch.load(tmp_path / "final0.pt"), torch.load(tmp_path / "final1.pt")
    assert torch.equal(f0, f1)
    from myrtle_vision.utils.optim import ParamArena
    torch.manual_seed(100)
    model = torch.nn.Sequential(torch.nn.Linear(12, 40), torch.nn.Tanh(), torch.nn.Linear(40, 24), torch.nn.Tanh(), torch.nn.Linear(24, 3))
    arena = ParamArena(list(model.named_parameters()) + [("unused", torch.nn.Parameter(torch.randn(5)))])
    g = torch.Generator().manual_seed(7)
    X, Y = torch.randn(8, 12, generator=g), torch.randn(8, 3, generator=g)
    ((model(X) - Y) ** 2).mean().backward()
    arena.sync_grads()
    want, got = arena.flat_grad, r0["grad"]
    assert float((want - got).norm() / want.norm()) < 2.0 ** -7            # measured 2e-3: bf16 is 2^-9 per rounding
    assert float((want - got).abs().max() / want.abs().max()) < 2.0 ** -6
    assert not torch.equal(want, got)                                        # it really went through bf16
    with pytest.raises(ValueError):
        from myrtle_vision.utils.ddp import GradAllReducer
        GradAllReducer(arena, exchange_dtype=torch.float16)


def _synthetic_arena(shapes):
    """[(name, shape)] -> ParamArena; 1-D tensors and '.bias' names land in the no-decay region."""
    from myrtle_vision.utils.optim import ParamArena
    return ParamArena([(n, torch.nn.Parameter(torch.zeros(s))) for n, s in shapes])


@pytest.mark.parametrize("case", ["no_decay_only", "decay_only", "one_param_over_cap", "tail_taper", "mixed_small"])
def test_gradient_buckets_tile_any_arena(case):
    """on ANY arena the bucket ranges are contiguous, cover [0, total) exactly once, every parameter belongs to
    exactly one bucket whose range contains its slot, no bucket straddles the decay / no-decay boundary, and a bucket exceeds
    the cap only when it holds a single parameter that is itself larger."""
    from myrtle_vision.utils.ddp import GradAllReducer
    shapes = {
        "no_decay_only": [(f"l{i}.bias", (100,)) for i in range(7)],                       # zero decay parameters
        "decay_only": [(f"l{i}.weight", (64, 64)) for i in range(9)],                      # zero no-decay parameters
        "one_param_over_cap": [("a.weight", (8, 8)), ("big.weight", (600, 512)), ("b.weight", (16, 16)), ("b.bias", (16,))],
        "tail_taper": [(f"l{i}.weight", (256, 256)) for i in range(12)] + [(f"l{i}.bias", (256,)) for i in range(12)],
        "mixed_small": [("w0.weight", (3, 5)), ("w0.bias", (3,)), ("w1.weight", (7, 3)), ("w1.bias", (7,)), ("scale", (1,))],
    }[case]
    arena = _synthetic_arena(shapes)
    cap, tail = 256 << 10, 64 << 10
    red = GradAllReducer(arena, bucket_bytes=cap, tail_bytes=tail)
    ranges = sorted((lo, hi, j0, j1) for lo, hi, j0, j1 in red.ranges)
    assert ranges[0][0] == 0 and ranges[-1][1] == arena.total
    assert all(a[1] == b[0] and a[3] == b[2] for a, b in zip(ranges, ranges[1:]))       # contiguous in elements AND in parameters
    assert ranges[0][2] == 0 and ranges[-1][3] == len(arena.params)
    assert sorted(red.bucket_of) == list(range(len(arena.params)))                      # every parameter exactly once
    for j, b in red.bucket_of.items():
        lo, hi, j0, j1 = red.ranges[b]
        assert j0 <= j < j1 and lo <= arena.offsets[j] and arena.offsets[j] + arena.params[j].numel() <= hi
    for lo, hi, j0, j1 in ranges:
        assert hi <= arena.n_decay or lo >= arena.n_decay or arena.n_decay in (0, arena.total)
        assert (hi - lo) * 4 <= cap or j1 - j0 == 1
    if case == "one_param_over_cap":
        assert any((hi - lo) * 4 > cap and j1 - j0 == 1 for lo, hi, j0, j1 in ranges)
    if case == "tail_taper":                                                            # the arena's first buckets = the last to complete
        assert (ranges[0][1] - ranges[0][0]) * 4 <= tail + 256 * 256 * 4
        assert max((hi - lo) * 4 for lo, hi, _, _ in ranges) > tail
    assert sum(red.sizes) == len(arena.params) and red.pending == red.sizes
    red.remove()


def test_gradient_bucket_layout_vit_b():
    """GradAllReducer's buckets at ViT-B size: they tile the arena exactly, none straddles the decay / no-decay boundary (the
    no-decay region completes only at the very end of backward), full-size buckets stay under the cap, and the buckets that
    complete LAST in backward (the first layers' weights = the start of the arena) taper to the tail size."""
    from myrtle_vision.utils.ddp import GradAllReducer
    from myrtle_vision.utils.optim import ParamArena
    nn = torch.nn
    blocks = [nn.ModuleDict(dict(n1=nn.LayerNorm(768), qkv=nn.Linear(768, 2304), proj=nn.Linear(768, 768), n2=nn.LayerNorm(768),
                                 fc1=nn.Linear(768, 3072), fc2=nn.Linear(3072, 768))) for _ in range(12)]
    model = nn.ModuleDict(dict(pe=nn.Linear(768, 768), blocks=nn.ModuleList(blocks), head=nn.Linear(768, 1000)))
    arena = ParamArena(model.named_parameters())
    red = GradAllReducer(arena, bucket_bytes=48 << 20, tail_bytes=12 << 20)
    ranges = sorted((lo, hi) for lo, hi, _, _ in red.ranges)
    assert ranges[0][0] == 0 and ranges[-1][1] == arena.total
    assert all(a[1] == b[0] for a, b in zip(ranges, ranges[1:]))                       # exact tiling
    assert all(hi <= arena.n_decay or lo >= arena.n_decay for lo, hi in ranges)        # no straddling
    sizes = [(hi - lo) * 4 for lo, hi in ranges]
    assert max(sizes) <= 48 << 20
    assert sizes[0] <= 12 << 20 and sizes[1] <= 12 << 20                               # the arena's first (= last finished) buckets
    assert sum(1 for s_ in sizes if s_ > 40 << 20) >= 5                                # the bulk still travels in large buckets
    assert set(red.bucket_of) == set(range(len(arena.params)))
    red.remove()


@pytest.mark.parametrize("q_format", ["FP16_32", "TF32", "FP16_16", "PyTorchINT8"])
def test_quantised_formats_force_fp32_on_every_leaf(q_format):
    """Fake-quantised values are fp32 by definition (reference utils/quantize.py:84): a model built with the default
    bf16 precision must switch EVERY leaf, including the Linear/LayerNorm modules ModelQuantizer re-classes."""
    from myrtle_vision.models.vit import ViT
    v = ViT(precision="bf16", q_format=q_format, decoder="classification", image_size=80, patch_size=16, num_classes=10,
            dim=128, depth=1, heads=2, mlp_dim=128, dropout=0.0, emb_dropout=0.0)
    precs = {n: m.precision for n, m in v.named_modules() if hasattr(m, "precision")}
    assert len(precs) > 8 and set(precs.values()) == {"fp32"}, precs
    # (convert() quantises the weights on the GPU kernels -- for every format: its precision plumbing is covered by the GPU
    # test tests/test_vit_parity.py::test_fake_quant_convert_matches_reference)


def test_bench_self_launches_two_ranks_and_prints_one_json_line():
    """``python bench.py --gpus 2`` started plainly spawns its own ranks (the reference scripts mp.spawn themselves,
    classification/train.py:349-356) and rank 0 prints ONE JSON line.  --dry-run: plumbing only, gloo, no GPU needed."""
    import json
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["MV_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                        "--dry-run"], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["value"] is None and "dry-run" in out["data"]
    # a failing rank fails the command: an unknown workload is rejected by every child
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "nope", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0


def test_bench_eight_ranks_dry_run_and_a_dying_rank_fails_the_command():
    """The shape of the driver's first real 8-GPU run, rehearsed on the CPU: ``bench.py --gpus 8`` -> eight fresh children
    (never a re-exec of a process that touched a GPU), one rendezvous on 127.0.0.1, ONE JSON line from rank 0 with n_gpus 8 and
    each rank's host-thread share; and when ONE rank exits non-zero before the rendezvous the command fails promptly instead
    of hanging the other seven in their first barrier."""
    import json
    import subprocess
    import sys
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR",
                                                             "LOCAL_WORLD_SIZE", "OMP_NUM_THREADS")}
    env["MV_DIST_BACKEND"] = "gloo"
    bench = os.path.join(ROOT, "bench.py")
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--steps", "2", "--warmup", "1", "--dry-run"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 8 and out["config"]["parallelism"] == "dp8" and out["config"]["global_batch"] == 8 * 256
    ncores = len(os.sched_getaffinity(0))
    assert out["config"]["host_threads_per_rank"] == max(1, min(16, ncores // 8))
    t0 = time.time()
    r = subprocess.run([sys.executable, bench, "--gpus", "8", "--steps", "2", "--dry-run", "--dry-run-fail-rank", "5"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]          # no throughput line from a broken job
    assert time.time() - t0 < 300


@pytest.mark.parametrize("fmt,outputs", [("FP16_32", False), ("TF32", False), ("FP16_16", True)])
def test_prepared_modules_actually_hold_their_quantisers(fmt, outputs):
    """Regression (round 2): quantisers assigned to a class-swapped module live in nn.Module._modules, which a class-level
    default of the same name shadows -- the weight quantiser and the FP16_16 output quantisers were silently skipped."""
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.quantize import QATLinear, Quantizer, _QATLayerNorm
    vit = ViT(decoder="classification", image_size=224, patch_size=16, num_classes=5, dim=64, depth=1, heads=1, mlp_dim=64,
              q_format=fmt)
    lin = [m for m in vit.modules() if isinstance(m, QATLinear)]
    lns = [m for m in vit.modules() if isinstance(m, _QATLayerNorm)]
    assert len(lin) == 6 and len(lns) == 3
    calls = []
    for m in lin:
        wq = m._sub("weight_fake_quant")
        assert isinstance(wq, Quantizer) and not wq.is_identity
        assert isinstance(m._sub("activation_post_process"), Quantizer) == outputs
        wq.register_forward_pre_hook(lambda mod, a: calls.append("w"))
    for m in lns:
        assert isinstance(m._modules.get("weight_quantizer"), Quantizer)
        assert isinstance(m._modules.get("activation_post_process"), Quantizer) == outputs
    # the forward really routes the weight through it (CPU tensors stop at the first HIP call, after the stub + weight quantiser)
    w = lin[0].weight
    try:
        lin[0](torch.zeros(2, w.shape[1]))
    except RuntimeError:
        pass
    assert calls == ["w"]
