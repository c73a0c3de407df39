"""Model-level parity of the HIP ViT against golden vectors from the reference (tests/golden/*.npz).

precision="fp32": the north-star tolerance, 1e-3 relative to the tensor's max magnitude (measured: ~1e-5), and
bit-exact argmax.  precision="bf16" (the benchmark configuration): bf16 activations/weights on MFMA cannot meet
1e-3 end to end (each bf16 rounding is 2^-9 = 2e-3 relative); the tests pin its measured envelope with little slack:
1.5e-2 of max |logit| (measured <= 8.3e-3) and 2e-2 relative L2 per gradient tensor (measured <= 1.1e-2, ViT-B depth 12
included), plus argmax equality wherever the reference's top-2 margin exceeds that envelope.

MV_TEST_REPORT=<file>: every check appends the value it measured (the numbers quoted above come from that file).
"""
import os
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from conftest import load_golden  # noqa: E402
from oracle.detinit import det_images, det_labels, det_param, summarize  # noqa: E402

CASES = ["micro_cls", "micro_cls_256", "micro_seg", "micro_seg_256", "tiny_cls", "base_cls", "base_seg", "base_seg_256"]
BF16_LOGITS, BF16_GRAD = 1.5e-2, 2e-2


EXACT = ("fp32", "bf16x3", "bf16x3h")      # the modes held to north_star's 1e-3 / bit-exact arg-max against the reference


def report(tag, value):
    path = os.environ.get("MV_TEST_REPORT")
    if path:
        with open(path, "a") as f:
            f.write(f"{tag} {value:.3e}\n")


def build(name, precision, q_format=None, **extra):
    from myrtle_vision.models.vit import ViT
    arrays, meta = load_golden(name)
    kw = dict(meta["kwargs"])
    vit = ViT(patch_size=16, q_format="FP32", precision=precision, **kw, **extra)
    sd = vit.state_dict()
    assert {k: list(v.shape) for k, v in sd.items()} == meta["param_shapes"]
    assert list(sd.keys()) == meta["state_keys"]            # the reference's state-dict ORDER (checkpoint wire format)
    vit.load_state_dict({k: det_param(k, v.shape) for k, v in sd.items()})
    vit = vit.cuda()
    if q_format is not None:
        vit.quantizer.prepare_qat(q_format)
    b = meta["batch"]
    img = det_images(name, b, kw["image_size"]).cuda()
    if kw["decoder"] == "classification":
        labels = det_labels(name, (b,), kw["num_classes"]).cuda()
    else:
        labels = det_labels(name, (b, kw["image_size"], kw["image_size"]), kw["num_classes"]).cuda()
    return vit, img, labels, arrays, meta


def rel(got, want):
    want = np.asarray(want, dtype=np.float64)
    return float(np.abs(np.asarray(got, dtype=np.float64) - want).max() / max(np.abs(want).max(), 1e-30))


def canonical(name):
    parts = name.split(".")
    if len(parts) >= 3 and parts[-2] == "1" and parts[-1] in ("weight", "bias"):
        parts = parts[:-2] + parts[-1:]
    return ".".join(parts)


def check_case(name, precision, tol_logits, tol_grad, q_format=None, **extra):
    from myrtle_vision.hip.functional import cross_entropy
    vit, img, labels, arrays, meta = build(name, precision, q_format, **extra)
    vit.train()
    logits = vit(img)
    loss = cross_entropy(logits, labels)
    loss.backward()
    torch.cuda.synchronize()
    lg = logits.detach().float().cpu()
    tag = f"{name}/{precision}" + (f"/{q_format}" if q_format else "")
    if "logits" in arrays:
        want = arrays["logits"]
        report(f"{tag} logits", rel(lg.numpy(), want))
        assert rel(lg.numpy(), want) < tol_logits
        top2 = np.sort(want, axis=1)[:, -2:]
        margin_ok = (top2[:, 1] - top2[:, 0]) > 2 * tol_logits * np.abs(want).max()
        assert (lg.argmax(1).numpy() == want.argmax(1))[margin_ok].all()
        if precision in EXACT:
            assert (lg.argmax(1).numpy() == want.argmax(1)).all()          # bit-exact class indices
    else:
        want = arrays["logits_sub"]
        report(f"{tag} logits", rel(lg[:, :, ::7, ::7].numpy(), want))
        assert rel(lg[:, :, ::7, ::7].numpy(), want) < tol_logits
        am = lg.argmax(1)[:, ::7, ::7].numpy()
        if precision in EXACT and q_format is None:
            assert (am == arrays["argmax_sub"]).all()                          # bit-exact class indices
        else:
            # a fake-quantiser is discontinuous (fp32-ulp differences flip ~1e-3 of its roundings): class indices are
            # compared wherever the reference's own top-2 margin exceeds the logit tolerance
            top2 = np.sort(want, axis=1)[:, -2:]
            margin_ok = (top2[:, 1] - top2[:, 0]) > 2 * tol_logits * np.abs(want).max()
            assert margin_ok.mean() > 0.5 and (am == arrays["argmax_sub"])[margin_ok].all()
        s_got, s_want = summarize(lg).numpy(), arrays["logits_summary"]
        # bf16 and fake-quantised paths: norms only (their rounding errors are correlated, and a plain sum over 1.7 M logits
        # compared against the l2 scale amplifies them 1000-fold: measured 3.7e-4 of the sum itself for FP16_32)
        # (bf16x3: its 2^-16 product errors are correlated across the 256 pixels one source logit is upsampled to, so the plain sum
        # over 1.1 M logits is held to 1e-3 of ITSELF (measured 6e-6) and the norms to 1e-3 of the l2 scale)
        idx = slice(0, 4) if precision == "fp32" and q_format is None else slice(1, 3)
        assert np.abs(s_got[idx] - s_want[idx]).max() / s_want[1] < tol_logits
        if precision in ("bf16x3", "bf16x3h"):
            assert abs(s_got[0] - s_want[0]) < tol_logits * max(abs(s_want[0]), s_want[1])
            assert abs(s_got[3] - s_want[3]) < tol_logits * max(abs(s_want[3]), s_want[1])
    assert abs(float(loss) - float(arrays["loss"])) < tol_logits * max(1.0, abs(float(arrays["loss"])))
    unused = []
    worst = 0.0
    for pname, p in vit.named_parameters():
        c = canonical(pname)
        if p.grad is None:
            unused.append(c)
            continue
        w = arrays[f"gsum:{c}"]
        got = summarize(p.grad.float().cpu()).numpy()
        if precision in EXACT and precision != "bf16x3h":
            # [sum, l2, absmax, weighted sum] against the l2 scale; first 16 values against absmax
            e = max(np.abs(got[:4] - w[:4]).max() / max(w[1], 1e-30), np.abs(got[4:] - w[4:]).max() / max(w[2], 1e-30))
        else:
            # bf16 rounding errors are correlated along rows/columns of a weight gradient, so plain sums over a
            # tensor amplify them; compare norms and sampled values here, full tensors in test_bf16_vs_fp32_*
            e = max(abs(got[1] - w[1]) / max(w[1], 1e-30), abs(got[2] - w[2]) / max(w[2], 1e-30),
                    np.abs(got[4:] - w[4:]).max() / max(w[2], 1e-30))
        worst = max(worst, e)
        assert e < tol_grad, (c, e)
        if f"grad:{c}" in arrays:
            assert rel(p.grad.float().cpu().numpy(), arrays[f"grad:{c}"]) < tol_grad, c
    assert sorted(unused) == sorted(meta["unused_params"])
    report(f"{tag} grad-summaries", worst)
    return worst


@pytest.mark.parametrize("precision,tol,tol_grad", [("fp32", 1e-3, 1e-3), ("bf16", BF16_LOGITS, 3e-2)])
def test_fused_segmentation_tail_matches_reference(precision, tol, tol_grad):
    """``vit.segmentation_loss`` (decoder + bilinear upsample + CE + argmax without the [B,C,H,W] logits) against the
    reference's loss / arg-max / gradients for micro_seg, and against ``cross_entropy(vit(img))`` on the same model."""
    from myrtle_vision.hip.functional import cross_entropy
    vit, img, labels, arrays, meta = build("micro_seg", precision)
    vit.train()
    loss, acc, pred = vit.segmentation_loss(img, labels)
    assert pred.dtype == torch.uint8 and pred.shape == labels.shape and not pred.requires_grad
    loss.backward()
    torch.cuda.synchronize()
    assert abs(float(loss) - float(arrays["loss"])) < tol * max(1.0, abs(float(arrays["loss"])))
    if precision == "fp32":
        assert (pred[:, ::7, ::7].cpu().numpy() == arrays["argmax_sub"]).all()
    fused = {}
    for pname, p in vit.named_parameters():
        if p.grad is None:
            continue
        c = canonical(pname)
        w = arrays[f"gsum:{c}"]
        got = summarize(p.grad.float().cpu()).numpy()
        if precision == "fp32":
            e = max(np.abs(got[:4] - w[:4]).max() / max(w[1], 1e-30), np.abs(got[4:] - w[4:]).max() / max(w[2], 1e-30))
        else:
            e = max(abs(got[1] - w[1]) / max(w[1], 1e-30), abs(got[2] - w[2]) / max(w[2], 1e-30),
                    np.abs(got[4:] - w[4:]).max() / max(w[2], 1e-30))
        assert e < tol_grad, (c, e)
        fused[pname] = p.grad.clone()
        p.grad = None
    # the drop-in path on the same weights: same loss, same accuracy, same gradients
    logits = vit(img)
    loss_u = cross_entropy(logits, labels)
    loss_u.backward()
    same = 1e-5 if precision == "fp32" else 2e-2
    assert abs(float(loss_u) - float(loss)) < same * max(1.0, abs(float(loss_u)))
    assert abs(float(acc) - float((logits.argmax(1) == labels).float().mean())) < 1e-3
    for pname, p in vit.named_parameters():
        if p.grad is None:
            continue
        d = (p.grad - fused[pname]).float().norm() / p.grad.float().norm().clamp_min(1e-30)
        assert float(d) < same, (pname, float(d))
    # evaluation: no graph, no backward kernel
    with torch.no_grad():
        l2, a2, p2 = vit.segmentation_loss(img, labels)
    assert float(l2) == float(loss) and torch.equal(p2, pred)


@pytest.mark.parametrize("name", CASES)
def test_fp32_matches_reference(name):
    check_case(name, "fp32", 1e-3, 1e-3)


@pytest.mark.parametrize("name", CASES)
def test_bf16x3_matches_reference(name):
    """``precision="bf16x3"`` (two bf16 pieces per Linear operand, three pairings: 2^-16 relative per product; attention,
    LayerNorm, GELU, residual stream as in fp32 mode) at the SAME bar as fp32 mode: logits, loss and gradients to 1e-3 of the
    reference, bit-exact class indices -- the fast arithmetic inside north_star's tolerance."""
    check_case(name, "bf16x3", 1e-3, 1e-3)


@pytest.mark.parametrize("name", CASES)
def test_bf16x3h_matches_reference(name):
    """``precision="bf16x3h"``: bf16x3 with the attention core on IEEE-half operands (fp32 sums, softmax and outputs; up to
    288 tokens: the 197- and the 257-token fixtures alike).  Against the REFERENCE: logits and loss at north_star's 1e-3
    (measured <= 1.6e-4), bit-exact class indices, gradient norms and sampled values at 1e-3; every gradient element is held to
    1e-3 (relative L2 per tensor) through the fp32 mode in test_bf16x3h_vs_fp32_full_tensors."""
    check_case(name, "bf16x3h", 1e-3, 1e-3)


@pytest.mark.parametrize("name", CASES)
def test_bf16x3h_vs_fp32_full_tensors(name):
    """Full-tensor chain for the gradients of ``bf16x3h``: reference -(summaries to 1e-3, measured 4e-5)-> fp32 HIP -(here: EVERY
    element of EVERY gradient tensor)-> bf16x3h HIP.  Bar 1e-3 relative L2 per tensor (measured <= 4.5e-4 on ViT-B and ViT-Tiny at
    depth 12; bf16 mode: 1.0e-2) and 1e-3 on the logits.  (The fixtures' summaries hold plain and position-weighted SUMS over up
    to 2.4 M elements against the l2 scale, which multiplies a correlated error of 1e-6 per element a thousand-fold: fit for the
    fp32-accurate modes, not a measure of a 2^-12 arithmetic -- test_bf16x3h_matches_reference compares norms and sampled values.)"""
    from myrtle_vision.hip.functional import cross_entropy
    res = {}
    for prec in ("fp32", "bf16x3h"):
        vit, img, labels, arrays, meta = build(name, prec)
        logits = vit(img)
        cross_entropy(logits, labels).backward()
        res[prec] = (logits.detach().float(), {k: p.grad.float() for k, p in vit.named_parameters() if p.grad is not None})
    l32, lh = res["fp32"][0], res["bf16x3h"][0]
    e = float((l32 - lh).abs().max() / l32.abs().max())
    report(f"{name}/bf16x3h-vs-fp32 logits", e)
    assert e < 1e-3
    worst = 0.0
    for k, g32 in res["fp32"][1].items():
        e = float((g32 - res["bf16x3h"][1][k]).norm() / g32.norm().clamp_min(1e-30))
        worst = max(worst, e)
        assert e < 1e-3, (k, e)
    report(f"{name}/bf16x3h-vs-fp32 grad-rel-l2", worst)


@pytest.mark.parametrize("precision", ["fp32", "bf16x3h", "bf16"])
def test_config4_at_its_own_shape_batch_64(precision):
    """Config 4 (ViT-B/16 segmentation on 256^2 inputs, 257 tokens) at a production batch, against the REFERENCE through the batch-1
    fixture ``base_seg_256`` and two size-independent properties:
      (A) a batch of 64 copies of the fixture's image and mask: every copy's logits, the loss and every gradient summary are the
          fixture's (the mean over identical items), at the precision's own bar -- the whole training step at batch 64;
      (B) the fixture's image at positions 0, 17 and 63 of a batch of 64 otherwise different images: its logits are still the
          fixture's, and every OTHER image's logits equal what that image gives in a batch of 4 (no leakage between images in the
          257-token attention, the position-embedding resize, the decoder or the upsampling at this batch)."""
    from myrtle_vision.hip.functional import cross_entropy
    exact = precision in EXACT
    tol = 1e-3 if exact else BF16_LOGITS
    tol_g = 1e-3 if exact else 3e-2
    vit, img1, lab1, arrays, meta = build("base_seg_256", precision)
    vit.train()
    want = arrays["logits_sub"]
    scale = float(np.abs(want).max())
    # (A)
    img, lab = img1.expand(64, -1, -1, -1).contiguous(), lab1.expand(64, -1, -1).contiguous()
    logits = vit(img)
    loss = cross_entropy(logits, lab)
    loss.backward()
    sub = logits.detach().float()[:, :, ::7, ::7].cpu().numpy()
    e = float(np.abs(sub - want).max() / scale)
    report(f"config4-b64/{precision} logits", e)
    assert e < tol
    if exact:
        assert (sub.argmax(1) == arrays["argmax_sub"]).all()
    assert abs(float(loss) - float(arrays["loss"])) < tol * max(1.0, abs(float(arrays["loss"])))
    worst = 0.0
    for pname, p in vit.named_parameters():
        if p.grad is None:
            continue
        w = arrays[f"gsum:{canonical(pname)}"]
        got = summarize(p.grad.float().cpu()).numpy()
        if precision == "fp32":
            e = max(np.abs(got[:4] - w[:4]).max() / max(w[1], 1e-30), np.abs(got[4:] - w[4:]).max() / max(w[2], 1e-30))
        else:                       # norms and sampled values (see check_case)
            e = max(abs(got[1] - w[1]) / max(w[1], 1e-30), abs(got[2] - w[2]) / max(w[2], 1e-30),
                    np.abs(got[4:] - w[4:]).max() / max(w[2], 1e-30))
        worst = max(worst, e)
        assert e < tol_g, (pname, e)
    report(f"config4-b64/{precision} grad-summaries", worst)
    # (B)
    vit.zero_grad(set_to_none=True)
    others = det_images("config4_batch", 64, 256).cuda()
    mixed = others.clone()
    for pos in (0, 17, 63):
        mixed[pos] = img1[0]
    with torch.no_grad():
        lm = vit(mixed).float()
        for pos in (0, 17, 63):
            got = lm[pos:pos + 1, :, ::7, ::7].cpu().numpy()
            assert float(np.abs(got - want).max() / scale) < tol
        # fp32-accumulating paths are batch-independent to rounding; the bound is the precision's own rounding unit
        small = {"fp32": 2e-5, "bf16x3h": 2e-4, "bf16": 2e-5}[precision]
        for lo in (4, 28, 56):
            alone = vit(mixed[lo:lo + 4].contiguous()).float()
            e = float((alone - lm[lo:lo + 4]).abs().max() / lm.abs().max())
            report(f"config4-b64/{precision} batch-independence", e)
            assert e < small, (lo, e)


@pytest.mark.parametrize("name", ["micro_cls", "micro_seg", "tiny_cls"])
@pytest.mark.parametrize("precision", ["bf16x3", "bf16x3h"])
def test_split_operand_gradients_vs_oracle_full_tensors(name, precision):
    """EVERY element of EVERY gradient tensor of the two tolerance-meeting modes against the pinned CPU oracle
    (``oracle.vit_oracle.loss_and_grads``: fp32 autograd of the restated reference, itself held to the reference's fixtures in
    tests/test_oracle_golden.py) -- not through our own fp32 mode: relative L2 per tensor < 1e-3 (measured: bf16x3 <= 2e-5,
    bf16x3h <= 5.2e-4, twelve layers deep for tiny_cls), logits < 1e-3, same class indices."""
    from myrtle_vision.hip.functional import cross_entropy
    from oracle.vit_oracle import ViTConfig, loss_and_grads
    torch.set_num_threads(8)
    vit, img, labels, arrays, meta = build(name, precision)
    cfg = ViTConfig(patch_size=16, **meta["kwargs"])
    params = {k: det_param(k, s) for k, s in cfg.param_shapes().items()}
    ref_logits, ref_loss, ref_grads = loss_and_grads(params, img.cpu(), labels.cpu(), cfg)
    vit.train()
    logits = vit(img)
    cross_entropy(logits, labels).backward()
    lg = logits.detach().float().cpu()
    err = float((lg - ref_logits).abs().max() / ref_logits.abs().max())
    assert err < 1e-3
    if lg.dim() == 2:
        assert torch.equal(lg.argmax(1), ref_logits.argmax(1))
    else:
        # 100 k upsampled pixels: bilinear interpolation puts many of them on a class boundary (top-2 margin ~ 0); every pixel whose
        # reference margin exceeds twice the measured logit error must agree, and they are nearly all of them
        top2 = ref_logits.topk(2, dim=1).values
        safe = (top2[:, 0] - top2[:, 1]) > 2 * err * ref_logits.abs().max()
        assert float(safe.float().mean()) > 0.99 and bool((lg.argmax(1) == ref_logits.argmax(1))[safe].all())
    worst = 0.0
    for k, p in vit.named_parameters():
        if ref_grads[k] is None:
            assert p.grad is None
            continue
        e = float((p.grad.float().cpu() - ref_grads[k]).norm() / ref_grads[k].norm().clamp_min(1e-30))
        worst = max(worst, e)
        assert e < 1e-3, (k, e)
    report(f"{name}/{precision}-vs-oracle grad-rel-l2", worst)


@pytest.mark.parametrize("name,precision", [("micro_cls", "fp32"), ("micro_cls_256", "fp32"), ("tiny_cls", "fp32"), ("base_cls", "fp32"),
                                            ("base_cls", "bf16x3"), ("base_cls", "bf16x3h"), ("tiny_cls", "bf16"), ("base_cls", "bf16")])
def test_prune_dead_tokens_changes_nothing(name, precision):
    """``ViT(prune_dead_tokens=True)`` (extension, off by default): the last block's FeedForward runs on the cls rows only -- the
    classification decoder reads nothing else (reference vit.py:335-342) and the other rows' gradient is exactly zero.  Same bars
    against the REFERENCE's logits, loss and every parameter gradient as the unpruned model, and against the unpruned model
    itself: fp32 modes to fp32 rounding (the two run the last MLP's products over different row counts, i.e. kernels)."""
    from myrtle_vision.hip.functional import cross_entropy
    exact = precision in EXACT
    check_case(name, precision, 1e-3 if exact else BF16_LOGITS, 1e-3 if exact else 3e-2, prune_dead_tokens=True)
    outs = []
    for prune in (False, True):
        vit, img, labels, _, _ = build(name, precision, prune_dead_tokens=prune)
        assert vit.transformer.cls_only_tail == prune
        vit.train()
        logits = vit(img)
        cross_entropy(logits, labels).backward()
        outs.append((logits.detach().float(), {k: p.grad.float().clone() for k, p in vit.named_parameters() if p.grad is not None}))
    (l0, g0), (l1, g1) = outs
    assert set(g0) == set(g1)
    tol = 2e-3 if precision == "bf16x3h" else 2e-5 if exact else 2e-2     # (x3h: the last MLP's different row count changes what
    #                                                                       the half roundings upstream see through dX)
    assert float((l0 - l1).abs().max() / l0.abs().max()) < tol
    for k in g0:
        assert float((g0[k] - g1[k]).norm() / g0[k].norm().clamp_min(1e-30)) < tol, k
    # a hook on the last block switches the pruning off for that call (something observes the full output)
    vit, img, _, _, _ = build(name, precision, prune_dead_tokens=True)
    seen = []
    h = vit.transformer.layers[-1][1].register_forward_hook(lambda m, a, o: seen.append(tuple(o.shape)))
    with torch.no_grad():
        vit(img)
    h.remove()
    assert seen and seen[0][1] == img.shape[-1] // 16 * (img.shape[-2] // 16) + 1


@pytest.mark.parametrize("name", CASES)
def test_bf16_matches_reference_within_bf16_envelope(name):
    # gradient SUMMARIES (norms + 16 sampled values against the tensor's abs-max) are a noisier statistic than the
    # per-tensor relative L2 pinned at 2e-2 in test_bf16_vs_fp32_full_tensors: one sampled value sets them
    check_case(name, "bf16", BF16_LOGITS, 3e-2)


@pytest.mark.parametrize("name", CASES)
def test_bf16_vs_fp32_full_tensors(name):
    """Full-tensor chain: reference -(1e-3, test above)-> fp32 HIP -(here)-> bf16 HIP.  Measured on MI355X:
    logits rel-max <= 8.3e-3, every gradient tensor rel-L2 <= 1.1e-2 (ViT-B depth 12 included); bounds 1.5e-2 / 2e-2."""
    from myrtle_vision.hip.functional import cross_entropy
    res = {}
    for prec in ("fp32", "bf16"):
        vit, img, labels, arrays, meta = build(name, prec)
        logits = vit(img)
        cross_entropy(logits, labels).backward()
        res[prec] = (logits.detach().float(), {k: p.grad.float() for k, p in vit.named_parameters() if p.grad is not None})
    l32, l16 = res["fp32"][0], res["bf16"][0]
    e = float((l32 - l16).abs().max() / l32.abs().max())
    report(f"{name}/bf16-vs-fp32 logits", e)
    assert e < BF16_LOGITS
    worst = 0.0
    for k, g32 in res["fp32"][1].items():
        g16 = res["bf16"][1][k]
        e = float((g32 - g16).norm() / g32.norm().clamp_min(1e-30))
        worst = max(worst, e)
        assert e < BF16_GRAD, (k, e)
    report(f"{name}/bf16-vs-fp32 grad-rel-l2", worst)


@pytest.mark.parametrize("name", ["base_cls_b32", "tiny_cls_b64"])
def test_top1_agreement_rate(name):
    """Top-1 agreement with the REFERENCE over every image of the round-3 logits-only fixtures (ViT-B x 32 images,
    ViT-Tiny x 64).  fp32 mode: 100 %, no margin filter.  bf16 (the benchmarked arithmetic): 100 % of the images whose
    reference top-2 margin exceeds twice the measured logit error of THIS run (not the 1.5e-2 envelope), and every flip
    among the remaining images is reported and bounded: a flipped image's reference margin must be below that error."""
    arrays, meta = load_golden(name)
    want = arrays["logits"]
    scale = np.abs(want).max()
    top2 = np.sort(want, axis=1)[:, -2:]
    margin = (top2[:, 1] - top2[:, 0]) / scale
    for precision in ("fp32", "bf16x3", "bf16x3h", "bf16"):
        vit, img, labels, _, _ = build(name, precision)
        vit.eval()
        with torch.no_grad():
            lg = vit(img).float().cpu().numpy()
        err = rel(lg, want)
        agree = lg.argmax(1) == want.argmax(1)
        report(f"{name}/{precision} top1-images", float(len(agree)))
        report(f"{name}/{precision} top1-agreement", float(agree.mean()))
        report(f"{name}/{precision} logits", err)
        if precision in EXACT:
            assert err < 1e-3 and agree.all()
            continue
        assert err < BF16_LOGITS
        safe = margin > 2 * err
        report(f"{name}/bf16 images-outside-margin", float(safe.sum()))
        report(f"{name}/bf16 flips-inside-margin", float((~agree).sum()))
        assert agree[safe].all()
        assert (margin[~agree] <= 2 * err).all()
        assert (~agree).sum() <= len(agree) // 8
        # the value the reference ranks first is, in our logits, within the error of our own maximum
        ours_at_ref = lg[np.arange(len(lg)), want.argmax(1)]
        assert ((lg.max(1) - ours_at_ref) / scale <= 2 * err).all()


def test_block_taps_and_attention_hook_fp32():
    """Per-block activations and the attn_output hook point (reference vit.py:80-82,94)."""
    vit, img, labels, arrays, meta = build("micro_cls", "fp32")
    taps = {}
    hooks = [blk[1].register_forward_hook(lambda m, a, o, i=i: taps.__setitem__(f"block{i}", o.detach()))
             for i, blk in enumerate(vit.transformer.layers)]
    attn0 = vit.transformer.layers[0][0].fn.fn
    hooks.append(attn0.attn_output.register_forward_hook(lambda m, a, o: taps.__setitem__("attn0", o.detach())))
    with torch.no_grad():
        logits = vit(img)
    for h in hooks:
        h.remove()
    assert rel(logits.cpu().numpy(), arrays["logits"]) < 1e-3
    for i in range(len(vit.transformer.layers)):
        assert rel(taps[f"block{i}"][:, :3].cpu().numpy(), arrays[f"block{i}_head"]) < 1e-3
    assert taps["attn0"].shape[-1] == 197
    assert float(np.abs(taps["attn0"][:, :, :4, :].cpu().numpy() - arrays["attn0_head"]).max()) < 1e-5
    # with the hook removed the fused path gives the same answer
    with torch.no_grad():
        logits2 = vit(img)
    assert rel(logits2.cpu().numpy(), arrays["logits"]) < 1e-3


@pytest.mark.parametrize("name,fmt", [("micro_cls_fp16_32", "FP16_32"), ("micro_cls_tf32", "TF32"),
                                      ("micro_cls_fp16_16", "FP16_16"), ("micro_seg_fp16_32", "FP16_32")])
def test_fake_quant_paths_match_reference_plumbing(name, fmt):
    # quantised paths: rounding flips of the discontinuous quantiser (see tests/test_oracle_golden.py) put the floor near
    # 1e-3; measured after round 2's fix of the skipped weight quantisers: logits <= 8.1e-4, gradient summaries <= 2.1e-3
    # (they were 1.1e-2 while the weights went unquantised: the old 2e-2 bound hid that)
    check_case(name, "fp32", 2e-3, 6e-3, q_format=fmt)


def test_fake_quant_convert_matches_reference():
    vit, img, labels, arrays, meta = build("micro_cls_fp16_32_conv", "fp32", "FP16_32")
    vit.train()
    with torch.no_grad():
        vit(img)
    vit.convert()
    vit.eval()
    with torch.no_grad():
        logits = vit(img)
    assert rel(logits.cpu().numpy(), arrays["logits"]) < 3e-3
    assert list(vit.state_dict().keys()) == meta["state_keys_after_convert"]


def test_cpu_forward_fails_loudly():
    from myrtle_vision.models.vit import ViT
    vit = ViT(decoder="classification", image_size=224, patch_size=16, num_classes=5, dim=64, depth=1, heads=1, mlp_dim=64)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        vit(torch.randn(1, 3, 224, 224))
