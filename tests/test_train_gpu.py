"""GPU: the reference's CPU-runnable plumbing case (BASELINE configs[0]) on the HIP path -- ViT-Tiny, synthetic
RESISC-45 layout, batch 8, one process; plus the segmentation loop and the quantised evaluation script path."""
import copy
import json
import os

import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import ROOT  # noqa: E402


def _config(tmp_path, task, size="tiny"):
    from myrtle_vision.datasets.synthetic import make_dlrsd, make_resisc45
    name = {"classification": f"vit_{size}.json", "segmentation": f"seg_{size}.json"}[task]
    cfg = json.load(open(os.path.join(ROOT, task, "train_configs", name)))
    data = json.load(open(os.path.join(ROOT, task, "data_configs", "data_config.json")))
    if task == "classification":
        data["dataset_path"] = make_resisc45(str(tmp_path / "NWPU-RESISC45"), classes=45, per_class=2)
    else:
        data["dataset_path"] = make_dlrsd(str(tmp_path / "DLRSD_dataset"), count=48)
    dpath = str(tmp_path / "data_config.json")
    json.dump(data, open(dpath, "w"))
    cfg["data_config_path"] = dpath
    t = cfg["train_config"]
    t.update(output_directory=str(tmp_path / "ckpt"), epochs=1, local_batch_size=8, global_batch_size=8, iters_per_checkpoint=2,
             iters_per_val=2, distributed=False, pretrained_backbone=None, tensorboard_dir=str(tmp_path / "runs"))
    cfg["vit_config"]["depth"] = 2                                          # keep the test short
    return cfg


@pytest.mark.parametrize("task", ["classification", "segmentation"])
def test_train_loop_writes_reloadable_checkpoints(tmp_path, task, capsys):
    from myrtle_vision.engine import evaluate, train_worker
    cfg = _config(tmp_path, task)
    iters = train_worker(0, 1, copy.deepcopy(cfg), task)
    out = capsys.readouterr().out
    assert iters >= 2 and "Iteration 1:" in out and "nan" not in out.lower()
    ckpts = sorted(os.listdir(cfg["train_config"]["output_directory"]))
    assert "vit_000000" in ckpts and "vit_000002" in ckpts                  # rank-0 checkpoints, reference naming
    ck = torch.load(os.path.join(cfg["train_config"]["output_directory"], "vit_000002"), map_location="cpu", weights_only=False)
    assert set(ck) == {"model", "optimizer", "lr_scheduler", "iteration"} and ck["iteration"] == 2
    if task == "segmentation":                                              # segmentation/train.py:69-71: accuracy, loss, miou
        logged = os.listdir(cfg["train_config"]["tensorboard_dir"])
        assert logged, "validation scalars were not written"
        if "scalars.jsonl" in logged:
            rows = [json.loads(l) for l in open(os.path.join(cfg["train_config"]["tensorboard_dir"], "scalars.jsonl"))]
            assert {r["tag"] for r in rows} == {"accuracy", "loss", "miou"} and {r["step"] for r in rows} >= {0, 2}
    # resume + evaluate from the checkpoint
    cfg2 = copy.deepcopy(cfg)
    cfg2["train_config"]["checkpoint_path"] = os.path.join(cfg["train_config"]["output_directory"], "vit_000002")
    res = evaluate(cfg2, task)
    assert 0.0 <= res["accuracy"] <= 1.0


def test_train_loop_clips_after_every_micro_batch(tmp_path, capsys):
    """clip_grad with gradient accumulation through the real loop (classification/train.py:265-270: clip_grad_norm_ after EVERY
    backward): local 4 / global 8 = two micro-batches per iteration, a threshold that engages.  The loop must run without the old
    "clips once per step" warning and take finite steps; the arithmetic of the sequence (backward, clip, backward, clip, step) is
    held to torch's in test_clip_grad_after_every_micro_batch_matches_the_reference_loop."""
    from myrtle_vision.engine import train_worker
    cfg = _config(tmp_path, "classification")
    cfg["train_config"].update(local_batch_size=4, global_batch_size=8, iters_per_checkpoint=1, iters_per_val=1000)
    cfg["train_config"]["clip_grad"] = 0.05
    iters = train_worker(0, 1, copy.deepcopy(cfg), "classification")
    out = capsys.readouterr().out
    assert iters >= 2 and "WARNING: clip_grad" not in out and "nan" not in out.lower()
    d = cfg["train_config"]["output_directory"]
    ck0 = torch.load(os.path.join(d, "vit_000000"), map_location="cpu", weights_only=False)
    ck1 = torch.load(os.path.join(d, "vit_000001"), map_location="cpu", weights_only=False)
    moved = max(float((ck1["model"][k].float() - ck0["model"][k].float()).abs().max()) for k in ck0["model"])
    assert 0 < moved < 1e-2                      # one AdamW step at the configured learning rate, finite


def test_train_starts_from_local_timm_backbone(tmp_path, capsys):
    """Row f4 end to end (segmentation/train.py:98,163-175; classification configs carry the same key): ``pretrained_backbone``
    = a LOCAL timm-format state dict.  The loop must start from exactly the renamed weights -- the iteration-0 checkpoint it
    writes before the first optimizer step holds them bit for bit, the decoder head (which the reference drops) stays at its
    random init, and the first forward equals a model that received the renamed dict through ``load_state_dict`` directly."""
    from myrtle_vision.engine import train_worker
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.models import rename_timm_state_dict
    from oracle.detinit import det_images, det_param, timm_source_shapes
    cfg = _config(tmp_path, "classification")
    v = cfg["vit_config"]
    shape_cfg = dict(embed_dim=v["embed_dim"], mlp_dim=v["mlp_dim"], patch_size=v["patch_size"], image_size=v["image_size"],
                     depth=v["depth"], num_classes=1000)                   # a timm checkpoint carries its own 1000-class head
    src = {k: det_param("timm:" + k, s) for k, s in timm_source_shapes(shape_cfg).items()}
    path = str(tmp_path / "vit_tiny_patch16_224.pth")
    torch.save(src, path)
    cfg["train_config"]["pretrained_backbone"] = path
    train_worker(0, 1, copy.deepcopy(cfg), "classification")
    assert "WARNING: pretrained_backbone" not in capsys.readouterr().out
    ck = torch.load(os.path.join(cfg["train_config"]["output_directory"], "vit_000000"), map_location="cpu", weights_only=False)
    renamed = rename_timm_state_dict(path, v, 45)
    assert len(renamed) == 4 + 12 * v["depth"]
    for k, w in renamed.items():
        assert torch.equal(ck["model"][k], w), k
    assert not torch.equal(ck["model"]["decoder.linear.weight"][:, :8], src["head.weight"][:45, :8])
    # the first forward: checkpointed start state == renamed dict loaded directly (+ the loop's random decoder)
    kw = dict(patch_size=v["patch_size"], q_format="FP32", decoder="classification", image_size=v["image_size"], num_classes=45,
              dim=v["embed_dim"], depth=v["depth"], heads=v["heads"], mlp_dim=v["mlp_dim"])
    a, b = ViT(**kw), ViT(**kw)
    a.load_state_dict(ck["model"])
    res = b.load_state_dict(renamed, strict=False)
    assert res.unexpected_keys == []
    b.load_state_dict({k: ck["model"][k] for k in res.missing_keys}, strict=False)
    a, b = a.cuda().eval(), b.cuda().eval()
    img = det_images("timm_backbone", 4, v["image_size"]).cuda()
    with torch.no_grad():
        assert torch.equal(a(img), b(img))


@pytest.mark.parametrize("precision", ["bf16x3", "bf16x3h"])
def test_train_loop_in_the_tolerance_meeting_precisions(tmp_path, capsys, precision):
    """The reference's classification loop (classification/train.py:226-303) with ``vit_config["precision"]`` = bf16x3 / bf16x3h (the
    extension key ``get_models`` reads): iterations run, the loss is finite and falls on the 90-image synthetic set, checkpoints
    reload, evaluation works -- the modes are wired through the engine, not only through the model class."""
    from myrtle_vision.engine import evaluate, train_worker
    cfg = _config(tmp_path, "classification")
    cfg["vit_config"]["precision"] = precision
    cfg["train_config"]["epochs"] = 2
    iters = train_worker(0, 1, copy.deepcopy(cfg), "classification")
    out = capsys.readouterr().out
    assert iters >= 4 and "nan" not in out.lower()
    losses = [float(l.split("loss=")[1].split()[0]) for l in out.splitlines() if l.startswith("Iteration")]
    assert len(losses) == iters and all(l == l and l < 20 for l in losses)
    cfg2 = copy.deepcopy(cfg)
    cfg2["train_config"]["checkpoint_path"] = os.path.join(cfg["train_config"]["output_directory"], "vit_000002")
    res = evaluate(cfg2, "classification")
    assert 0.0 <= res["accuracy"] <= 1.0


def test_quantized_evaluation_path(tmp_path):
    from myrtle_vision.engine import evaluate, train_worker
    cfg = _config(tmp_path, "classification")
    train_worker(0, 1, copy.deepcopy(cfg), "classification")
    cfg["train_config"]["checkpoint_path"] = os.path.join(cfg["train_config"]["output_directory"], "vit_000002")
    fp32 = evaluate(copy.deepcopy(cfg), "classification")
    cfg["vit_config"]["q_format"] = "FP16_32"
    q = evaluate(copy.deepcopy(cfg), "classification", quantize=True, calib_steps=1)
    assert abs(q["accuracy"] - fp32["accuracy"]) <= 0.2                     # fp16 weights barely move predictions


@pytest.mark.parametrize("decoder,precision,size", [("classification", "bf16", 224), ("classification", "fp32", 224),
                                                    ("segmentation", "bf16", 224), ("classification", "bf16", 80)])
def test_gradients_land_in_their_arena_slots(decoder, precision, size):
    """Zero-copy gradients: after backward EVERY arena parameter's .grad IS its slot of the flat buffer (the kernels wrote
    there; autograd adopted the view -- no add, no copy), values equal the plain path without an arena, and a second
    micro-batch accumulates (slot += new) exactly like torch's ``grad += new``."""
    from myrtle_vision.hip.functional import cross_entropy
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.optim import ParamArena
    from myrtle_vision.utils.utils import seed_everything
    kw = dict(decoder=decoder, image_size=size, patch_size=16, num_classes=7, dim=128, depth=2, heads=2, mlp_dim=256,
              dropout=0.0, emb_dropout=0.0)
    g = torch.Generator().manual_seed(3)
    img = torch.randn(2, 3, size, size, generator=g).cuda()
    labels = (torch.randint(0, 7, (2,), generator=g) if decoder == "classification"
              else torch.randint(0, 7, (2, size, size), generator=g)).cuda()

    def loss_of(v, x, y):
        return v.segmentation_loss(x, y)[0] if decoder == "segmentation" else cross_entropy(v(x), y)

    seed_everything(11)
    ref = ViT(precision=precision, q_format="FP32", **kw).cuda()
    loss_of(ref, img, labels).backward()
    want = {n: p.grad.clone() for n, p in ref.named_parameters() if p.grad is not None}

    seed_everything(11)
    vit = ViT(precision=precision, q_format="FP32", **kw).cuda()
    arena = ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names())
    arena.flat_grad.fill_(float("nan"))                      # nothing may rely on a zero-filled buffer
    arena.zero_grad()
    loss_of(vit, img, labels).backward()
    names = dict(zip(arena.names, range(len(arena.names))))
    assert set(names) == set(want)
    for n, j in names.items():
        p = arena.params[j]
        assert p.grad is not None, n
        # at a non-native grid the positional embedding goes through the bicubic-resize glue on torch: its gradient is
        # materialised by autograd and moved into the slot by sync_grads() (the fallback path)
        if not (n == "pos_embedding" and size != 224):
            assert p.grad.data_ptr() == arena.slot(j).data_ptr(), n                         # written in place
        assert torch.equal(p.grad, want[n]), n                                              # same kernels, same bits
    # second micro-batch: gradients accumulate into the slots
    loss_of(vit, img, labels).backward()
    arena.sync_grads()
    for n, j in names.items():
        p = arena.params[j]
        assert p.grad.data_ptr() == arena.slot(j).data_ptr(), n
        assert torch.allclose(p.grad, 2 * want[n], rtol=1e-6, atol=1e-30), n


_DDP_KW = {"classification": dict(decoder="classification", num_classes=10), "segmentation": dict(decoder="segmentation", num_classes=17)}


def _ddp_model(task):
    from myrtle_vision.models.vit import ViT
    return ViT(precision="bf16", q_format="FP32", image_size=224, patch_size=16, dim=128, depth=2, heads=2, mlp_dim=256,
               dropout=0.0, emb_dropout=0.0, **_DDP_KW[task]).cuda()


def _ddp_data(task):
    g = torch.Generator().manual_seed(5)
    X = torch.randn(8, 3, 224, 224, generator=g)
    Y = torch.randint(0, 10, (8,), generator=g) if task == "classification" else torch.randint(0, 17, (8, 224, 224), generator=g)
    return X, Y


def _ddp_loss(task, vit, x, y):
    from myrtle_vision.hip.functional import cross_entropy
    if task == "segmentation":
        return vit.segmentation_loss(x, y)[0]                               # what the segmentation loop runs (engine.py)
    return cross_entropy(vit(x), y)


def _ddp_rank(rank, world, port, tmpdir, task="classification", exchange="fp32"):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)          # both ranks share GPU 0; the exchange is the point
    torch.cuda.set_device(0)
    from myrtle_vision.utils.ddp import GradAllReducer, broadcast_parameters
    from myrtle_vision.utils.optim import AdamW, ParamArena
    from myrtle_vision.utils.utils import seed_everything
    seed_everything(100 + rank)                                             # different initial weights: the broadcast must fix it
    vit = _ddp_model(task)
    arena = ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names())
    opt = AdamW(arena, lr=1e-3, weight_decay=0.05)
    red = GradAllReducer(arena, bucket_bytes=256 << 10,                     # several buckets
                         exchange_dtype=torch.bfloat16 if exchange == "bf16" else torch.float32, measure=True)
    assert len(red.ranges) > 2
    broadcast_parameters(arena)
    opt.grad_scale = red.grad_scale
    X, Y = _ddp_data(task)
    x, y = X[rank::world].cuda(), Y[rank::world].cuda()                     # DistributedSampler-style shard
    for step in range(3):
        opt.zero_grad()
        _ddp_loss(task, vit, x, y).backward()
        red.finish()
        if step == 0:
            torch.save({"grad": (arena.flat_grad * red.grad_scale).cpu(), "param": arena.flat_param.cpu()},
                       os.path.join(tmpdir, f"r{rank}.pt"))
        opt.step()
    exposed = red.exposed_ms()
    assert exposed is not None and exposed >= 0.0                           # the span events were recorded and are readable
    torch.save(arena.flat_param.cpu(), os.path.join(tmpdir, f"final{rank}.pt"))
    dist.destroy_process_group()


@pytest.mark.parametrize("task,exchange", [("classification", "fp32"), ("segmentation", "fp32"), ("classification", "bf16")])
def test_ddp_two_ranks_match_single_process(tmp_path, task, exchange):
    """SURVEY 8e parity check on the HIP path, for BOTH training loops (BASELINE configs 3 and 4): all-reduced gradients of a
    rank-sharded batch == the single-process gradient of the concatenated batch (mean loss), and the ranks hold identical
    parameters after K optimizer steps.  ``bf16``: the opt-in half-width exchange (utils/ddp.py), same statements at bf16
    resolution."""
    import torch.multiprocessing as mp
    from myrtle_vision.utils.optim import ParamArena
    from myrtle_vision.utils.utils import seed_everything
    port = 29700 + (os.getpid() % 200) + {"classification": 0, "segmentation": 211}[task] + (427 if exchange == "bf16" else 0)
    mp.spawn(_ddp_rank, args=(2, port, str(tmp_path), task, exchange), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["param"], r1["param"]) and torch.equal(r0["grad"], r1["grad"])
    seed_everything(100)                                                    # rank 0's initial weights
    vit = _ddp_model(task)
    arena = ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names())
    assert torch.equal(arena.flat_param.cpu(), r0["param"])
    X, Y = _ddp_data(task)
    _ddp_loss(task, vit, X.cuda(), Y.cuda()).backward()
    arena.sync_grads()
    want, got = arena.flat_grad.cpu(), r0["grad"]
    # bf16 activations: the two shards round differently from the concatenated batch; compare at bf16 resolution
    e = float((want - got).norm() / want.norm())
    assert e < 2e-2, e
    f0, f1 = torch.load(tmp_path / "final0.pt"), torch.load(tmp_path / "final1.pt")
    assert torch.equal(f0, f1)


@pytest.mark.parametrize("task", ["classification", "segmentation"])
def test_graphed_step_equals_eager_step(task):
    """utils/graph.py: the training step captured in ONE HIP graph and replayed gives, bit for bit, the parameters of the
    eager step (same kernels, same order, same addresses of the scalars' VALUES): three replays on three different batches
    with a learning-rate change in between (the schedule reaches the captured AdamW launch through device memory), then an
    eager evaluation forward that must see the LAST parameters (the weight caches' epoch moved)."""
    from myrtle_vision.hip.functional import cross_entropy
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.graph import GraphedTrainStep
    from myrtle_vision.utils.optim import AdamW, ParamArena
    from myrtle_vision.utils.utils import seed_everything
    kw = dict(image_size=224, patch_size=16, dim=128, depth=2, heads=2, mlp_dim=256, dropout=0.0, emb_dropout=0.0, **_DDP_KW[task])
    loss_fn = (lambda m, x, y: m.segmentation_loss(x, y)[0]) if task == "segmentation" else (lambda m, x, y: cross_entropy(m(x), y))
    g = torch.Generator().manual_seed(9)
    nc = kw["num_classes"]
    batches = [(torch.randn(4, 3, 224, 224, generator=g).cuda(),
                (torch.randint(0, nc, (4,), generator=g) if task == "classification"
                 else torch.randint(0, nc, (4, 224, 224), generator=g)).cuda()) for _ in range(4)]
    lrs = [1e-3, 1e-3, 4e-4, 7e-4]

    def build():
        seed_everything(21)
        vit = ViT(precision="bf16", q_format="FP32", **kw).cuda().train()
        opt = AdamW(ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names()), lr=1e-3, weight_decay=0.05)
        opt.max_grad_norm = 1.0                                              # the clip coefficient is a device scalar as well
        return vit, opt

    def set_lr(opt, lr):
        for grp in opt.param_groups:
            grp["lr"] = lr

    # eager reference: 3 warm-up steps on batch 0 (what GraphedTrainStep's constructor performs), then batches 1..3
    vit_e, opt_e = build()
    losses_e = []
    for i in [0, 0, 0, 1, 2, 3]:
        set_lr(opt_e, lrs[i])
        opt_e.zero_grad()
        loss = loss_fn(vit_e, *batches[i])
        loss.backward()
        opt_e.step()
        losses_e.append(float(loss))
    vit_g, opt_g = build()
    graphed = GraphedTrainStep(vit_g, opt_g, loss_fn, *batches[0], warmup=3)
    assert opt_g.step_count == 3
    losses_g = []
    for i in (1, 2, 3):
        set_lr(opt_g, lrs[i])
        losses_g.append(float(graphed(*batches[i])))
    torch.cuda.synchronize()
    assert opt_g.step_count == opt_e.step_count == 6
    assert losses_g == losses_e[3:]
    assert torch.equal(opt_g.arena.flat_param, opt_e.arena.flat_param)
    assert torch.equal(opt_g.exp_avg, opt_e.exp_avg) and torch.equal(opt_g.exp_avg_sq, opt_e.exp_avg_sq)
    # an eager forward AFTER the replays uses the updated parameters (bf16 weight copies re-prepared)
    vit_g.eval(), vit_e.eval()
    with torch.no_grad():
        assert torch.equal(vit_g(batches[0][0]), vit_e(batches[0][0]))


def test_graph_owns_its_weight_tables_while_other_models_come_and_go():
    """the captured step contains a batched weight-preparation launch that reads a device TABLE of
    pointers.  The graph must own that table and prepare ITS arena's weights only: other models are created, used (which
    rebuilds the preparation tables of THEIR owners) and freed between replays, the allocator is pushed to recycle their
    memory, and the replays must still equal the eager steps bit for bit.  Also: one model's optimizer step does not make
    another model's prepared copies stale, and epochs never repeat after an arena dies."""
    import gc
    from myrtle_vision.hip import ops
    from myrtle_vision.hip.functional import cross_entropy
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.graph import GraphedTrainStep
    from myrtle_vision.utils.optim import AdamW, ParamArena
    from myrtle_vision.utils.utils import seed_everything
    kw = dict(decoder="classification", num_classes=7, image_size=224, patch_size=16, dim=128, depth=2, heads=2, mlp_dim=256)
    loss_fn = lambda m, x, y: cross_entropy(m(x), y)
    g = torch.Generator().manual_seed(3)
    batches = [(torch.randn(4, 3, 224, 224, generator=g).cuda(), torch.randint(0, 7, (4,), generator=g).cuda()) for _ in range(5)]

    def build(seed=21, with_opt=True):
        seed_everything(seed)
        vit = ViT(precision="bf16", q_format="FP32", **kw).cuda().train()
        opt = AdamW(ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names()), lr=1e-3, weight_decay=0.05) if with_opt else None
        return vit, opt

    vit_e, opt_e = build()
    for i in [0, 0, 0, 1, 2, 3, 4]:
        opt_e.zero_grad()
        loss_fn(vit_e, *batches[i]).backward()
        opt_e.step()
    want = opt_e.arena.flat_param.clone()
    del vit_e, opt_e

    vit_g, opt_g = build()
    graphed = GraphedTrainStep(vit_g, opt_g, loss_fn, *batches[0], warmup=3)
    assert graphed._keep, "the capture recorded no preparation launch to own"
    epochs_seen = []
    for i in (1, 2, 3, 4):
        # another model with its own arena: trains a step (its table replaces nothing of the graph's), then dies
        other, opt_o = build(seed=100 + i)
        opt_o.zero_grad()
        loss_fn(other, *batches[0]).backward()
        opt_o.step()
        with torch.no_grad():
            other.eval()(batches[0][0])
        # a free-standing evaluation model (no arena): its prepared copies must survive the graph's replays untouched
        frozen, _ = build(seed=7, with_opt=False)
        frozen.eval()
        with torch.no_grad():
            y0 = frozen(batches[0][0]).clone()
        w = frozen.transformer.layers[0][0].fn.fn.to_qkv.weight
        pw_before = ops.prepared_weight(w)
        stamp = (pw_before.version, pw_before.ptr, pw_before.epoch)
        graphed(*batches[i])
        torch.cuda.synchronize()
        pw_after = ops.prepared_weight(w)
        assert pw_after is pw_before and (pw_after.version, pw_after.ptr, pw_after.epoch) == stamp    # not re-prepared
        with torch.no_grad():
            assert torch.equal(frozen(batches[0][0]), y0)
        epochs_seen.append(ops.cache_epoch(ops._owner_of(vit_g.transformer.layers[0][0].fn.fn.to_qkv.weight)))
        del other, opt_o, frozen, w, pw_before, pw_after
        gc.collect()
        torch.cuda.empty_cache()
        junk = [torch.full((1 << 20,), float(i), device="cuda") for _ in range(8)]    # recycle whatever was freed
        del junk
    assert epochs_seen == sorted(set(epochs_seen)), epochs_seen                      # strictly increasing: never repeats
    assert torch.equal(opt_g.arena.flat_param, want)


def test_direct_arena_write_refreshes_weight_copies():
    """a torch-side in-place write to ``arena.flat_param`` (an EMA, loading into the arena) is seen by the bf16
    weight caches without any bump_versions() call -- the arena buffer's own version counter is part of the cache epoch."""
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.optim import ParamArena
    from myrtle_vision.utils.utils import seed_everything
    seed_everything(4)
    kw = dict(decoder="classification", image_size=224, patch_size=16, num_classes=7, dim=128, depth=1, heads=2, mlp_dim=256)
    vit = ViT(precision="bf16", q_format="FP32", **kw).cuda().eval()
    arena = ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names())
    x = torch.randn(2, 3, 224, 224, generator=torch.Generator().manual_seed(1)).cuda()
    with torch.no_grad():
        a = vit(x).clone()
        arena.flat_param.mul_(0.5)                                           # no bump_versions()
        b = vit(x).clone()
        seed_everything(4)
        ref = ViT(precision="bf16", q_format="FP32", **kw).cuda().eval()
        for p in ref.parameters():
            p.mul_(0.5)
        want = ref(x)
    assert not torch.equal(a, b)
    assert torch.equal(b, want)                                              # same halved weights, same kernels: same bits


def test_int8_quantized_evaluation_path(tmp_path):
    """classification/test_quantize.py with q_format PyTorchINT8 (BASELINE config 5's entry point): the model is moved to
    the GPU FIRST and prepared afterwards, so the observers must be created on the device (the min/max kernel updates
    their state with device atomics)."""
    from myrtle_vision.engine import evaluate, train_worker
    cfg = _config(tmp_path, "classification")
    train_worker(0, 1, copy.deepcopy(cfg), "classification")
    cfg["train_config"]["checkpoint_path"] = os.path.join(cfg["train_config"]["output_directory"], "vit_000002")
    fp32 = evaluate(copy.deepcopy(cfg), "classification")
    cfg["vit_config"]["q_format"] = "PyTorchINT8"
    q = evaluate(copy.deepcopy(cfg), "classification", quantize=True, calib_steps=1)
    assert abs(q["accuracy"] - fp32["accuracy"]) <= 0.25


def test_minmax_observer_rejects_host_state():
    from myrtle_vision.hip import ops
    with pytest.raises(RuntimeError):
        ops.minmax_update(torch.randn(100).cuda(), torch.tensor([float("inf"), float("-inf"), 0.0, 0.0]))


def test_clip_grad_matches_torch_clip_grad_norm(tmp_path):
    """train_config.clip_grad (classification/train.py:265-270): arena norm + coefficient folded into the AdamW kernel ==
    torch.nn.utils.clip_grad_norm_ followed by torch.optim.AdamW.step, for a clipping and a non-clipping threshold."""
    from myrtle_vision.hip.functional import cross_entropy
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.optim import AdamW, ParamArena
    from myrtle_vision.utils.utils import seed_everything
    from oracle.optim_oracle import reference_adamw
    kw = dict(decoder="classification", image_size=224, patch_size=16, num_classes=7, dim=128, depth=2, heads=2, mlp_dim=256)
    g = torch.Generator().manual_seed(3)
    img, labels = torch.randn(4, 3, 224, 224, generator=g).cuda(), torch.randint(0, 7, (4,), generator=g).cuda()
    for max_norm in (0.05, 1e6):
        seed_everything(5)
        vit = ViT(precision="fp32", q_format="FP32", **kw).cuda()
        opt = AdamW(ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names()), lr=1e-3, weight_decay=0.05)
        opt.max_grad_norm = max_norm
        opt.zero_grad()
        cross_entropy(vit(img), labels).backward()
        used = [(n, p) for n, p in vit.named_parameters() if p.grad is not None]
        # torch on copies of the same parameters and gradients
        ref_params = [(n, torch.nn.Parameter(p.detach().clone())) for n, p in used]
        for (_, rp), (_, p) in zip(ref_params, used):
            rp.grad = p.grad.detach().clone()
        ref_opt = reference_adamw(ref_params, lr=1e-3, weight_decay=0.05)
        total = torch.nn.utils.clip_grad_norm_([rp for _, rp in ref_params], max_norm)
        ref_opt.step()
        opt.step()
        got_norm, coef = (float(v) for v in opt.last_grad_norm.tolist())
        assert abs(got_norm - float(total)) < 1e-5 * float(total)
        assert abs(coef - min(1.0, max_norm / (float(total) + 1e-6))) < 1e-6
        assert (coef < 1.0) == (max_norm < float(total))
        for (n, rp), (_, p) in zip(ref_params, used):
            assert torch.allclose(p.detach(), rp.detach(), rtol=2e-6, atol=2e-7), n


@pytest.mark.parametrize("grad_scale", [1.0, 0.5])
def test_clip_grad_after_every_micro_batch_matches_the_reference_loop(grad_scale):
    """clip_grad with n_batch_accum = 2 (classification/train.py:257-275): backward, clip_grad_norm_ on the RUNNING accumulated
    gradient, backward again (accumulating onto the clipped gradient), clip again, step.  Here: AdamW.clip_accumulated between the
    micro-batches (arena rescaled in place) and the last clip inside the AdamW kernel, against torch doing exactly the reference's
    sequence on copies.  grad_scale = 0.5 plays a two-rank SUM all-reduce whose 1/world is still pending (both "ranks" hold the
    same data, so the exchange doubles the arena): the first clip must apply the 1/world, and so must the step."""
    from myrtle_vision.hip.functional import cross_entropy
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.optim import AdamW, ParamArena
    from myrtle_vision.utils.utils import seed_everything
    from oracle.optim_oracle import reference_adamw
    kw = dict(decoder="classification", image_size=224, patch_size=16, num_classes=7, dim=128, depth=2, heads=2, mlp_dim=256)
    g = torch.Generator().manual_seed(3)
    imgs = [torch.randn(4, 3, 224, 224, generator=g).cuda() for _ in range(2)]
    labs = [torch.randint(0, 7, (4,), generator=g).cuda() for _ in range(2)]
    world = round(1.0 / grad_scale)
    for max_norm in (0.05, 1e6):
        seed_everything(5)
        vit = ViT(precision="fp32", q_format="FP32", **kw).cuda()
        opt = AdamW(ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names()), lr=1e-3, weight_decay=0.05)
        opt.max_grad_norm, opt.grad_scale = max_norm, grad_scale
        start = {n: p.detach().clone() for n, p in vit.named_parameters()}
        opt.zero_grad()
        micro = []                                                   # the per-micro-batch gradients, for the torch side
        for k in range(2):
            before = opt.arena.flat_grad.clone()
            cross_entropy(vit(imgs[k]), labs[k]).backward()
            opt.arena.sync_grads()
            micro.append(opt.arena.flat_grad - before)
            opt.arena.flat_grad.mul_(world)                          # the SUM all-reduce over `world` ranks holding the same data
            if k == 0:
                opt.clip_accumulated()
                first = [float(v) for v in opt.last_grad_norm.tolist()]
        used = [(n, p) for n, p in vit.named_parameters() if p.grad is not None]
        ref_params = [(n, torch.nn.Parameter(start[n].clone())) for n, _ in used]
        ref_opt = reference_adamw(ref_params, lr=1e-3, weight_decay=0.05)
        totals = []
        for k in range(2):
            for (n, rp), (_, p) in zip(ref_params, used):
                lo = p.grad.data_ptr() - opt.arena.flat_grad.data_ptr()
                gk = micro[k][lo // 4: lo // 4 + p.numel()].view_as(p)
                rp.grad = gk.clone() if rp.grad is None else rp.grad + gk
            totals.append(float(torch.nn.utils.clip_grad_norm_([rp for _, rp in ref_params], max_norm)))
        ref_opt.step()
        opt.step()
        assert abs(first[0] - totals[0]) < 1e-5 * totals[0]
        got_norm, coef = (float(v) for v in opt.last_grad_norm.tolist())
        assert abs(got_norm - totals[1]) < 2e-5 * totals[1]
        assert (coef < 1.0) == (max_norm < totals[1])
        for (n, rp), (_, p) in zip(ref_params, used):
            assert torch.allclose(p.detach(), rp.detach(), rtol=5e-6, atol=5e-7), n


def test_dropout_statistics_determinism_and_backward():
    """nn.Dropout on the Philox kernel (vit.py:50,52,75,311): keep rate, 1/(1-p) scaling, mask = f(seed, offset) (bit-exact
    against the numpy restatement in oracle/), backward uses the same mask, eval / p = 0 are the identity."""
    import numpy as np
    from myrtle_vision.hip import functional as F
    from myrtle_vision.hip import ops
    from oracle.dropout_oracle import dropout_keep_mask
    x = torch.randn(1001, 333).cuda()
    for p in (0.1, 0.5):
        y = ops.dropout(x, p, seed=1234567, offset=89)
        keep = (y != 0)
        assert abs(float(keep.float().mean()) - (1 - p)) < 4e-3                   # 333 k samples: 4 sigma ~ 3.5e-3 at p = 0.5
        assert torch.allclose(y[keep], x[keep] / (1 - p), rtol=1e-6)
        want = dropout_keep_mask(x.numel(), p, 1234567, 89).reshape(x.shape)
        assert np.array_equal(keep.cpu().numpy(), want)
        assert torch.equal(y, ops.dropout(x, p, seed=1234567, offset=89))          # deterministic
        assert not torch.equal(y, ops.dropout(x, p, seed=1234567, offset=90))
        yb = ops.dropout(x.bfloat16(), p, seed=1234567, offset=89)
        assert torch.equal(yb != 0, keep)
    torch.manual_seed(7)
    xr = torch.randn(64, 128, device="cuda", requires_grad=True)
    y = F.dropout(xr, 0.3, training=True)
    y.backward(torch.ones_like(y))
    assert torch.equal(xr.grad != 0, y != 0) and torch.allclose(xr.grad[xr.grad != 0], torch.tensor(1 / 0.7, device="cuda"))
    assert F.dropout(xr, 0.3, training=False) is xr and F.dropout(xr, 0.0, training=True) is xr
    torch.manual_seed(7)
    assert torch.equal(F.dropout(xr.detach(), 0.3, True), y.detach())              # reproducible under the torch seed


def test_vit_with_dropout_trains_and_eval_is_deterministic():
    from myrtle_vision.hip.functional import cross_entropy
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.utils import seed_everything
    kw = dict(decoder="classification", image_size=224, patch_size=16, num_classes=7, dim=128, depth=2, heads=2, mlp_dim=256)
    g = torch.Generator().manual_seed(3)
    img, labels = torch.randn(4, 3, 224, 224, generator=g).cuda(), torch.randint(0, 7, (4,), generator=g).cuda()
    for precision in ("bf16", "fp32"):
        seed_everything(5)
        vit = ViT(precision=precision, q_format="FP32", dropout=0.1, emb_dropout=0.1, **kw).cuda()
        base = ViT(precision=precision, q_format="FP32", dropout=0.0, emb_dropout=0.0, **kw).cuda()
        base.load_state_dict(vit.state_dict())
        vit.train()
        seed_everything(9)
        l1 = cross_entropy(vit(img), labels)
        l1.backward()
        assert torch.isfinite(l1) and all(torch.isfinite(p.grad).all() for p in vit.parameters() if p.grad is not None)
        g1 = vit.transformer.layers[0][1].fn.fn.net[0].weight.grad.clone()
        seed_everything(9)
        vit.zero_grad()
        l2 = cross_entropy(vit(img), labels)
        l2.backward()
        assert float(l1) == float(l2) and torch.equal(g1, vit.transformer.layers[0][1].fn.fn.net[0].weight.grad)   # same seed, same masks
        seed_everything(10)
        assert float(cross_entropy(vit(img), labels)) != float(l1)                                             # other seed, other masks
        vit.eval()
        base.eval()
        with torch.no_grad():
            assert torch.equal(vit(img), base(img))            # eval: dropout is the identity, the fused path runs


def test_block_chain_is_by_identity_not_address():
    """The forward-order link that routes a LayerNorm backward's column sums to the producing Linear's bias gradient is
    keyed by tensor identity: a freed-and-reallocated buffer at the same address fed to an unrelated block inherits
    nothing."""
    from myrtle_vision.hip import functional as F
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.utils import seed_everything
    seed_everything(1)
    kw = dict(decoder="classification", image_size=224, patch_size=16, num_classes=7, dim=128, depth=2, heads=2, mlp_dim=256)
    vit = ViT(precision="bf16", q_format="FP32", **kw).cuda()
    blk_a, blk_b = vit.transformer.layers[0][1], vit.transformer.layers[1][1]
    x1 = torch.randn(2, 197, 128, device="cuda")
    out = blk_a(x1)                                               # records (out, fc2 bias of block a) as the chain link
    addr = out.data_ptr()
    del out
    x2 = torch.empty(2, 197, 128, device="cuda").normal_()       # very likely the same address (caching allocator)
    x2.requires_grad_(True)
    y = blk_b(x2)
    y.float().sum().backward()
    assert blk_a.fn.fn.net[3].bias.grad is None, f"block a's bias inherited a gradient (x2 at {x2.data_ptr():#x}, old out at {addr:#x})"
    assert blk_b.fn.fn.net[3].bias.grad is not None and x2.grad is not None
    F.chain_reset()


def test_bench_two_ranks_on_one_gpu_gloo():
    """The N > 1 path of bench.py end to end with real compute: ``python bench.py --gpus 2`` self-launches two ranks (both on
    GPU 0, gradients exchanged over gloo -- a rehearsal of the RCCL path's control flow) and rank 0 prints one JSON line with
    the whole-job throughput."""
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env["MV_DIST_BACKEND"] = "gloo"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "16", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 32 and out["value"] > 0 and out["scaling"] == "weak"
