"""CPU fp32 restatement of the reference ViT hot path (TEST INFRASTRUCTURE).

A *functional* re-derivation of ``/root/reference/src/myrtle_vision/models/vit.py``
in plain torch fp32 ops on a ``{state-dict name: tensor}`` parameter dict; the
backward comes from torch autograd of this forward.  Pinned against golden
vectors produced by the reference's own ``ViT`` (``tests/golden/gen_golden.py``),
see ``tests/test_oracle_golden.py``.  Never imported by the product package.

Line citations are into the reference file above unless stated.
"""
import math
from typing import Callable, Dict, Optional

import torch
import torch.nn.functional as F


class ViTConfig:
    """Constructor arguments of the reference ``ViT`` (``vit.py:165-184``)."""

    def __init__(self, *, decoder="classification", image_size=224, patch_size=16,
                 num_classes=45, dim=192, depth=12, heads=3, mlp_dim=768,
                 channels=3, dim_head=64, num_det_tokens=100):
        assert image_size % patch_size == 0                       # :186-188
        assert (image_size // patch_size) ** 2 > 16               # :191-195 (MIN_NUM_PATCHES)
        assert decoder in ("classification", "segmentation")      # :197-201 (detection: out of scope)
        self.decoder, self.image_size, self.patch_size = decoder, image_size, patch_size
        self.num_classes, self.dim, self.depth, self.heads = num_classes, dim, depth, heads
        self.mlp_dim, self.channels, self.dim_head = mlp_dim, channels, dim_head
        self.num_det_tokens = num_det_tokens

    def param_shapes(self) -> Dict[str, tuple]:
        """State-dict names and shapes, in the reference's registration order
        (``vit.py:218-222``, Transformer ``:127-153``, decoders ``:332-333,353-354``)."""
        d, m = self.dim, self.mlp_dim
        pd = self.channels * self.patch_size ** 2
        inner = self.dim_head * self.heads
        s = {
            "pos_embedding": (1, 14 * 14 + 1, d),
            "pos_embedding_det": (1, self.num_det_tokens, d),
            "cls_token": (1, 1, d),
            "det_tokens": (1, self.num_det_tokens, d),
            "patch_to_embedding.weight": (d, pd),
            "patch_to_embedding.bias": (d,),
        }
        for i in range(self.depth):
            p = f"transformer.layers.{i}"
            s[f"{p}.0.fn.norm.weight"] = (d,)
            s[f"{p}.0.fn.norm.bias"] = (d,)
            s[f"{p}.0.fn.fn.to_qkv.weight"] = (3 * inner, d)
            s[f"{p}.0.fn.fn.to_qkv.bias"] = (3 * inner,)
            s[f"{p}.0.fn.fn.to_out.0.weight"] = (d, inner)
            s[f"{p}.0.fn.fn.to_out.0.bias"] = (d,)
            s[f"{p}.1.fn.norm.weight"] = (d,)
            s[f"{p}.1.fn.norm.bias"] = (d,)
            s[f"{p}.1.fn.fn.net.0.weight"] = (m, d)
            s[f"{p}.1.fn.fn.net.0.bias"] = (m,)
            s[f"{p}.1.fn.fn.net.3.weight"] = (d, m)
            s[f"{p}.1.fn.fn.net.3.bias"] = (d,)
        s["decoder.norm.weight"] = (d,)
        s["decoder.norm.bias"] = (d,)
        s["decoder.linear.weight"] = (self.num_classes, d)
        s["decoder.linear.bias"] = (self.num_classes,)
        return s


def patchify(img: torch.Tensor, p: int) -> torch.Tensor:
    """``vit.py:271-275``: (B,C,H,W) -> (B, H/p*W/p, p*p*C), k = (py*p+px)*C + c."""
    b, c, h, w = img.shape
    return (img.reshape(b, c, h // p, p, w // p, p)
            .permute(0, 2, 4, 3, 5, 1)
            .reshape(b, (h // p) * (w // p), p * p * c))


def resized_pos_embedding(pos: torch.Tensor, gh: int, gw: int) -> torch.Tensor:
    """``vit.py:292-302``: cls slot kept, 14x14 grid bicubically resized to (gh, gw)."""
    cls_pos, grid = pos[:, 0:1, :], pos[:, 1:, :]
    grid = grid.transpose(1, 2).reshape(1, -1, 14, 14)
    grid = F.interpolate(grid, size=(gh, gw), mode="bicubic", align_corners=False)
    grid = grid.reshape(1, -1, gh * gw).transpose(1, 2)
    return torch.cat((cls_pos, grid), dim=1)


def layer_norm(x, w, b, eps=1e-5):
    """``nn.LayerNorm(dim)`` (``vit.py:37,332,353``): biased variance, eps 1e-5."""
    mu = x.mean(dim=-1, keepdim=True)
    var = ((x - mu) ** 2).mean(dim=-1, keepdim=True)
    return (x - mu) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    """``nn.GELU()`` default = exact erf form (``vit.py:49``)."""
    return 0.5 * x * (1.0 + torch.erf(x * (1.0 / math.sqrt(2.0))))


Quant = Optional[Callable[[str, torch.Tensor], torch.Tensor]]


def vit_forward(params: Dict[str, torch.Tensor], img: torch.Tensor, cfg: ViTConfig,
                quant: Quant = None, taps: Optional[dict] = None, quant_outputs: bool = False,
                probs_site: bool = False, linear_fn=None) -> torch.Tensor:
    """``ViT.forward`` (``vit.py:267-320``) for the classification and segmentation
    decoders.  ``quant(site, tensor)`` is the fake-quant hook (identity when None);
    sites follow ``ModelQuantizer._prepare_qat_fp16_32/_tf32``
    (``utils/quantize.py:289-327``): the input of every Linear / LayerNorm
    ("act:<module>") and every Linear weight ("w:<module>").  ``quant_outputs``
    adds the sites of ``_prepare_qat_fp16_16`` (``utils/quantize.py:253-287``): the
    output of every Linear / LayerNorm ("out:<module>"), the input of every GELU
    ("act:gelu": nn.GELU gets a QuantStub in front but, not being in torch's
    observed-module list, no output observer) and every FloatFunctional result
    ("ff:<name>": cls_token_cat, pos_embedding_cat, pos_embedding_add, res_add).
    ``taps`` (optional dict) receives intermediate activations.  ``probs_site`` adds one site the reference does not
    have, "attn:probs" on the softmax output: the rounding-error budget of the bf16 kernels (tests/test_error_budget.py)
    needs it, no fixture uses it.  ``linear_fn(name, x, weight, bias)`` (optional) replaces the arithmetic of every nn.Linear
    product -- emulation studies of GEMM number formats (tests/test_error_budget.py); no fixture uses it.
    """
    q = quant if quant is not None else (lambda site, t: t)
    qo = q if quant_outputs else (lambda site, t: t)
    P = params
    p = cfg.patch_size
    b, _, h, w = img.shape

    def linear(name, x):
        if linear_fn is not None:
            return linear_fn(name, x, P[f"{name}.weight"], P[f"{name}.bias"])
        y = F.linear(q(f"act:{name}", x), q(f"w:{name}", P[f"{name}.weight"]), P[f"{name}.bias"])
        return qo(f"out:{name}", y)

    def norm(name, x):
        return qo(f"out:{name}", layer_norm(q(f"act:{name}", x), P[f"{name}.weight"], P[f"{name}.bias"]))

    x = linear("patch_to_embedding", patchify(img, p))              # :271-278
    x = qo("ff:cls_token_cat", torch.cat((P["cls_token"].expand(b, -1, -1), x), dim=1))  # :283-290 (det branch dead)
    pos = qo("ff:pos_embedding_cat", resized_pos_embedding(P["pos_embedding"], h // p, w // p))  # :292-302
    x = qo("ff:pos_embedding_add", x + pos)                         # :305-310; dropout p=0 :311
    if taps is not None:
        taps["embed"] = x

    H = cfg.heads
    scale = cfg.dim_head ** -0.5                                    # :70
    for i in range(cfg.depth):                                      # Transformer.forward :155-161
        pre = f"transformer.layers.{i}"
        # Residual(PreNorm(Attention)) :131-141, Attention.forward :84-99
        y = norm(f"{pre}.0.fn.norm", x)
        n, c = y.shape[1], y.shape[2]
        qkv = linear(f"{pre}.0.fn.fn.to_qkv", y)
        qkv = qkv.reshape(b, n, 3, H, c // H).permute(2, 0, 3, 1, 4)  # :87-89
        qh, kh, vh = qkv[0], qkv[1], qkv[2]
        attn = (qh @ kh.transpose(-2, -1)) * scale                   # :92
        attn = attn.softmax(dim=-1)                                  # :93
        if probs_site:
            attn = q("attn:probs", attn)
        if taps is not None:
            taps[f"attn{i}"] = attn
        o = (attn @ vh).transpose(1, 2).reshape(b, n, c)             # :96
        x = qo("ff:res_add", linear(f"{pre}.0.fn.fn.to_out.0", o) + x)   # :98, Residual :27
        # Residual(PreNorm(FeedForward)) :142-151, FeedForward :47-53
        y = norm(f"{pre}.1.fn.norm", x)
        hdn = gelu_erf(qo("act:gelu", linear(f"{pre}.1.fn.fn.net.0", y)))
        x = qo("ff:res_add", linear(f"{pre}.1.fn.fn.net.3", hdn) + x)
        if taps is not None:
            taps[f"block{i}"] = x

    if cfg.decoder == "classification":                             # :335-342
        out = linear("decoder.linear", norm("decoder.norm", x[:, 0]))
    else:                                                           # :359-374
        y = linear("decoder.linear", norm("decoder.norm", x[:, 1:]))
        g = cfg.image_size // p
        y = y.transpose(1, 2).reshape(b, cfg.num_classes, g, g)
        out = F.interpolate(y, size=(cfg.image_size, cfg.image_size), mode="bilinear",
                            align_corners=False)                    # nn.Upsample(size, 'bilinear') :355
    return out


def loss_and_grads(params, img, labels, cfg, quant: Quant = None, quant_outputs: bool = False):
    """One training micro-step as the reference's loop does it
    (``classification/train.py:245-259``, ``segmentation/train.py:260-270``):
    forward, mean cross-entropy, backward.  Returns (logits, loss, grads)."""
    leaves = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
    logits = vit_forward(leaves, img, cfg, quant, quant_outputs=quant_outputs)
    loss = F.cross_entropy(logits, labels)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else None) for k, v in leaves.items()}
    return logits.detach(), loss.detach(), grads
