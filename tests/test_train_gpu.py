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
             iters_per_val=2, distributed=False, pretrained_backbone=None)
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
    # resume + evaluate from the checkpoint
    cfg2 = copy.deepcopy(cfg)
    cfg2["train_config"]["checkpoint_path"] = os.path.join(cfg["train_config"]["output_directory"], "vit_000002")
    res = evaluate(cfg2, task)
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


def _ddp_rank(rank, world, port, tmpdir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)          # both ranks share GPU 0; the exchange is the point
    torch.cuda.set_device(0)
    from myrtle_vision.hip.functional import cross_entropy
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.ddp import GradAllReducer, broadcast_parameters
    from myrtle_vision.utils.optim import AdamW, ParamArena
    from myrtle_vision.utils.utils import seed_everything
    seed_everything(100 + rank)                                             # different initial weights: the broadcast must fix it
    vit = ViT(precision="bf16", q_format="FP32", decoder="classification", image_size=224, patch_size=16, num_classes=10,
              dim=128, depth=2, heads=2, mlp_dim=256, dropout=0.0, emb_dropout=0.0).cuda()
    arena = ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names())
    opt = AdamW(arena, lr=1e-3, weight_decay=0.05)
    red = GradAllReducer(arena, bucket_bytes=256 << 10)                     # several buckets
    assert len(red.ranges) > 2
    broadcast_parameters(arena)
    opt.grad_scale = red.grad_scale
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 3, 224, 224, generator=g), torch.randint(0, 10, (8,), generator=g)
    x, y = X[rank::world].cuda(), Y[rank::world].cuda()                     # DistributedSampler-style shard
    for step in range(3):
        opt.zero_grad()
        cross_entropy(vit(x), y).backward()
        red.finish()
        if step == 0:
            torch.save({"grad": (arena.flat_grad * red.grad_scale).cpu(), "param": arena.flat_param.cpu()},
                       os.path.join(tmpdir, f"r{rank}.pt"))
        opt.step()
    torch.save(arena.flat_param.cpu(), os.path.join(tmpdir, f"final{rank}.pt"))
    dist.destroy_process_group()


def test_ddp_two_ranks_match_single_process(tmp_path):
    """SURVEY 8e parity check on the HIP path: all-reduced gradients of a rank-sharded batch == the single-process gradient
    of the concatenated batch (mean loss), and the ranks hold identical parameters after K optimizer steps."""
    import torch.multiprocessing as mp
    from myrtle_vision.hip.functional import cross_entropy
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.optim import ParamArena
    from myrtle_vision.utils.utils import seed_everything
    port = 29700 + (os.getpid() % 200)
    mp.spawn(_ddp_rank, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = torch.load(tmp_path / "r0.pt"), torch.load(tmp_path / "r1.pt")
    assert torch.equal(r0["param"], r1["param"]) and torch.equal(r0["grad"], r1["grad"])
    seed_everything(100)                                                    # rank 0's initial weights
    vit = ViT(precision="bf16", q_format="FP32", decoder="classification", image_size=224, patch_size=16, num_classes=10,
              dim=128, depth=2, heads=2, mlp_dim=256, dropout=0.0, emb_dropout=0.0).cuda()
    arena = ParamArena(vit.named_parameters(), skip=vit.unused_parameter_names())
    assert torch.equal(arena.flat_param.cpu(), r0["param"])
    g = torch.Generator().manual_seed(5)
    X, Y = torch.randn(8, 3, 224, 224, generator=g), torch.randint(0, 10, (8,), generator=g)
    cross_entropy(vit(X.cuda()), Y.cuda()).backward()
    arena.sync_grads()
    want, got = arena.flat_grad.cpu(), r0["grad"]
    # bf16 activations: the two shards round differently from the concatenated batch; compare at bf16 resolution
    assert float((want - got).norm() / want.norm()) < 2e-2
    f0, f1 = torch.load(tmp_path / "final0.pt"), torch.load(tmp_path / "final1.pt")
    assert torch.equal(f0, f1)
