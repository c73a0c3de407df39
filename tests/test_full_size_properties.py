"""GPU: size-independent properties at BASELINE.json's FULL sizes (configs[1]: ViT-B/16 224^2 bf16, batch 256, M = 50 432
token rows), where the oracle would take minutes: random-row samples of the full-size kernels against fp64 math, and the
batch-decomposition properties a ViT without batch statistics must satisfy exactly (up to fp32 summation order):

  * per-sample independence: the logits of image i do not depend on which batch it travels in;
  * gradient additivity: the gradient of the mean loss over 256 images is the mean of the gradients of its two halves --
    the same identity the DDP path relies on (reference: classification/train.py:170-176 + DistributedDataParallel).

Tolerances are written next to each check."""
import pytest
import torch

pytestmark = pytest.mark.gpu

VIT_B = dict(decoder="classification", image_size=224, patch_size=16, num_classes=1000, dim=768, depth=12, heads=12,
             mlp_dim=3072, dropout=0.0, emb_dropout=0.0)
M_FULL = 256 * 197


@pytest.fixture(scope="module")
def ops():
    from myrtle_vision.hip import ops as _ops
    _ops.lib()
    assert torch.cuda.is_available(), "these tests need a GPU"
    return _ops


def g(seed):
    return torch.Generator().manual_seed(seed)


def rel_l2(a, b):
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


# ---------------------------------------------------------------- full-size kernels, sampled rows vs fp64
@pytest.mark.parametrize("N,K", [(2304, 768), (768, 3072), (3072, 768)])
def test_full_size_gemm_nt_sampled_rows(ops, N, K):
    M = M_FULL
    x = (torch.randn(M, K, generator=g(1)) * 0.5).to(torch.bfloat16).cuda()
    w = (torch.randn(N, K, generator=g(2)) * K ** -0.5).cuda()
    b = torch.randn(N, generator=g(3)).cuda()
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    ops.linear_fwd(x, M, K, w, b, out, N)
    rows = torch.cat([torch.arange(0, 300), torch.randint(0, M, (2048,), generator=g(4)), torch.arange(M - 300, M)]).cuda()
    want = x[rows].double() @ w.to(torch.bfloat16).double().t() + b.double()
    err = (out[rows].double() - want).abs().max() / want.abs().max()
    assert float(err) < 6e-3                       # one bf16 output rounding (2^-9 relative) on top of fp32 accumulation
    # every row was written (no tile skipped by the tail split): compare row norms against a cheap full-size proxy
    assert torch.isfinite(out.float()).all() and int((out.float().abs().sum(1) == 0).sum()) == 0


def test_full_size_gemm_tn_matches_fp64_on_sampled_columns(ops):
    M, N, K = M_FULL, 768, 3072                    # dW of fc2: [N, K] = dy[M, N]^T x[M, K]
    dy = (torch.randn(M, N, generator=g(5)) * 0.1).to(torch.bfloat16).cuda()
    x = (torch.randn(M, K, generator=g(6)) * 0.5).to(torch.bfloat16).cuda()
    dw, db = ops.linear_dw(dy, x, M, N, K)
    cols = torch.randint(0, K, (64,), generator=g(7)).cuda()
    want = dy.double().t() @ x[:, cols].double()
    assert rel_l2(dw[:, cols], want) < 2e-5        # fp32 accumulation over 50 432 rows, fixed split order
    assert rel_l2(db, dy.double().sum(0)) < 2e-5


def test_full_size_attention_sampled_heads_and_key_permutation(ops):
    B, S, H, D = 256, 197, 12, 64
    qkv = (torch.randn(B, S, 3 * H * D, generator=g(8)) * 0.7).to(torch.bfloat16).cuda()
    out, lse = ops.attention_fwd(qkv, B, S, H, D ** -0.5)
    q5 = qkv.view(B, S, 3, H, D)
    for b, h in [(0, 0), (17, 5), (255, 11)]:
        q, k, v = (q5[b, :, i, h].double() for i in range(3))
        want = torch.softmax(q @ k.t() * D ** -0.5, -1) @ v
        got = out.view(B, S, H, D)[b, :, h].double()
        assert float((got - want).abs().max() / want.abs().max()) < 2e-2      # P is rounded to bf16 before P.V
    # permuting the keys and values of every sequence the same way leaves softmax(QK^T)V unchanged (up to the order of
    # the fp32 sums over keys and the bf16 rounding of P, which sees different running maxima)
    perm = torch.randperm(S, generator=g(9)).cuda()
    q5p = q5.clone()
    q5p[:, :, 1] = q5[:, perm, 1]
    q5p[:, :, 2] = q5[:, perm, 2]
    out_p, _ = ops.attention_fwd(q5p.view(B, S, 3 * H * D).contiguous(), B, S, H, D ** -0.5)
    assert rel_l2(out_p.float(), out.float()) < 6e-3


def test_full_size_attention_f16_forward_and_backward_sampled_heads(ops):
    """The half-operand attention core of precision "bf16x3h" at the benchmark's size (256 images x 12 heads x 197 tokens): sampled
    (image, head) pairs against fp64 autograd on the same half-rounded q / k / v -- forward 6e-4, each of dq / dk / dv 2e-3 (the
    unit test's bars), with a gradient whose magnitude differs 1e6-fold between images (the per-(image, head) scale)."""
    B, S, H, D = 256, 197, 12, 64
    qkv = (torch.randn(B, S, 3 * H * D, generator=g(20)) * 0.9).cuda()
    mag = torch.logspace(-5, 1, B).view(B, 1, 1)                                     # per-image gradient magnitude
    dout = (torch.randn(B, S, H * D, generator=g(21)) * mag).cuda()
    with ops.segments(4):
        q16 = ops.cast_f16(qkv)
        out, lse = ops.attention_fwd_f16(q16, B, S, H, D ** -0.5)
        dqkv = ops.attention_bwd_f16(q16, out, dout, lse, B, S, H, D ** -0.5)
    assert bool(torch.isfinite(out).all()) and bool(torch.isfinite(dqkv).all())
    q5 = q16.view(B, S, 3, H, D)
    for b, h in [(0, 0), (100, 7), (255, 11)]:
        ref = q5[b, :, :, h].double().cpu().requires_grad_(True)                      # [S, 3, D]
        p = torch.softmax(ref[:, 0] @ ref[:, 1].t() * D ** -0.5, -1)
        want = p @ ref[:, 2]
        want.backward(dout.view(B, S, H, D)[b, :, h].double().cpu())
        got, want = out.view(B, S, H, D)[b, :, h].double().cpu(), want.detach()
        assert float((got - want).abs().max() / want.abs().max()) < 6e-4
        gq = dqkv.view(B, S, 3, H, D)[b, :, :, h].double().cpu()
        for i, name in enumerate("qkv"):
            assert rel_l2(gq[:, i], ref.grad[:, i]) < 2e-3, (b, h, name)


def test_full_size_layernorm_rows_are_normalised(ops):
    M, D = M_FULL, 768
    x = (torch.randn(M, D, generator=g(10)) * 3 + 1.5).cuda()
    gamma, beta = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
    y, mean, rstd = ops.layernorm_fwd(x, D, M, D, gamma, beta, torch.float32)
    assert float(y.mean(1).abs().max()) < 1e-5 and float((y.var(1, unbiased=False) - 1).abs().max()) < 1e-4
    assert rel_l2(mean, x.double().mean(1)) < 1e-6


# ---------------------------------------------------------------- the whole model at batch 256
# (the benchmark arithmetic and the fastest one inside north_star's tolerance, both at the benchmark's batch)
@pytest.fixture(scope="module", params=["bf16", "bf16x3h"])
def vit_b(request):
    from myrtle_vision.models.vit import ViT
    from myrtle_vision.utils.utils import seed_everything
    seed_everything(7)
    vit = ViT(precision=request.param, q_format="FP32", **VIT_B).cuda()
    vit.train()
    return vit


def _grads(vit, img, labels):
    from myrtle_vision.hip.functional import cross_entropy
    for p in vit.parameters():
        p.grad = None
    loss = cross_entropy(vit(img), labels)
    loss.backward()
    skip = set(vit.unused_parameter_names())
    return float(loss.detach()), {n: p.grad.detach().clone() for n, p in vit.named_parameters() if n not in skip and p.grad is not None}


def test_full_batch_logits_are_per_sample_independent(vit_b):
    img = torch.randn(256, 3, 224, 224, generator=g(11)).cuda()
    with torch.no_grad():
        full = vit_b(img).float()
        part = vit_b(img[40:48].contiguous()).float()
    # same per-row arithmetic whatever the batch (the GEMM kernels differ with M -- 8-phase 256^2 tiles vs 128^2 tiles --
    # but every variant accumulates K in the same order): identical up to fp32 noise, and identical class decisions
    # (bf16x3h: the split-operand products pick a K-split by M, a different fp32 summation order: 3e-5 measured)
    assert rel_l2(full[40:48], part) < (1e-5 if vit_b.precision == "bf16" else 2e-4)
    assert torch.equal(full[40:48].argmax(1), part.argmax(1))


def test_full_batch_gradient_is_the_mean_of_its_halves(vit_b):
    img = torch.randn(256, 3, 224, 224, generator=g(12)).cuda()
    labels = torch.randint(0, 1000, (256,), generator=g(13)).cuda()
    loss, full = _grads(vit_b, img, labels)
    l0, h0 = _grads(vit_b, img[:128].contiguous(), labels[:128].contiguous())
    l1, h1 = _grads(vit_b, img[128:].contiguous(), labels[128:].contiguous())
    assert abs(loss - 0.5 * (l0 + l1)) < 1e-5 * abs(loss)
    assert set(full) == set(h0) == set(h1) and len(full) > 140
    worst = max(rel_l2(0.5 * (h0[n] + h1[n]), full[n]) for n in full)
    # activations are per-sample identical; only the fp32 order of the sums over the batch rows differs
    # (bf16x3h: the attention gradient's power-of-two scale is per (image, head), so it is batch-independent too)
    assert worst < (2e-4 if vit_b.precision == "bf16" else 5e-4), worst
