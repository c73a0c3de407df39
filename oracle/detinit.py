"""Deterministic, formula-defined parameters and inputs (test infrastructure).

Golden fixtures store only *outputs*; parameters and inputs are regenerated
from (name, shape) on both sides (reference import in ``gen_golden.py``, oracle
and HIP path in ``tests/``).  CPU ``torch.Generator`` streams are stable for a
given torch build, and the fixtures record the torch version they came from.
"""
import zlib

import torch


def _gen(tag: str) -> torch.Generator:
    g = torch.Generator(device="cpu")
    g.manual_seed(zlib.crc32(tag.encode("utf-8")) & 0x7FFFFFFF)
    return g


def det_tensor(tag: str, shape, scale: float = 1.0, shift: float = 0.0):
    """fp32 tensor ~ N(shift, scale^2) that depends only on (tag, shape)."""
    t = torch.randn(tuple(shape), generator=_gen(tag), dtype=torch.float32)
    return t * scale + shift


def det_param(name: str, shape):
    """Parameter value as a function of its state-dict name and shape.

    Linear weights ~ N(0, 1/fan_in), biases ~ N(0, 0.1^2), LayerNorm weight
    ~ 1 + 0.1 N, embeddings ~ 0.5 N: "lively" enough that every term of the
    forward/backward contributes at O(1).
    """
    shape = tuple(shape)
    if name.endswith("norm.weight"):
        return det_tensor(name, shape, 0.1, 1.0)
    if name.endswith(".bias"):
        return det_tensor(name, shape, 0.1)
    if name.endswith(".weight") and len(shape) == 2:
        return det_tensor(name, shape, shape[1] ** -0.5)
    return det_tensor(name, shape, 0.5)


def det_state_dict(shapes: dict):
    """{name: shape} -> {name: tensor}."""
    return {k: det_param(k, s) for k, s in shapes.items()}


def det_images(tag: str, batch: int, size: int, channels: int = 3):
    """Images ~ N(0,1): the range Normalize(0.5, 0.5) produces."""
    return det_tensor(f"img:{tag}", (batch, channels, size, size))


def det_labels(tag: str, shape, num_classes: int):
    return torch.randint(0, num_classes, tuple(shape), generator=_gen(f"lab:{tag}"))


def summarize(t: torch.Tensor, head: int = 16):
    """Small, order-sensitive summary of a tensor for fixtures: [sum, l2, abs-max,
    position-weighted sum] + first ``head`` values."""
    f = t.detach().double().flatten()
    w = torch.linspace(-1.0, 1.0, f.numel(), dtype=torch.float64)
    stats = torch.stack([f.sum(), f.norm(), f.abs().max(), (f * w).sum()])
    return torch.cat([stats, f[:head]]).float()


def timm_source_shapes(cfg: dict):
    """Key ORDER and shapes of a timm 0.5.4 ``VisionTransformer.state_dict()`` (cls_token, pos_embed, conv patch
    embedding, blocks, final norm, head) at the width ``cfg`` names -- the INPUT of the reference's
    ``rename_timm_state_dict`` (utils/models.py:154-223); values are ``det_param("timm:" + key, shape)``."""
    D, M, P = cfg["embed_dim"], cfg["mlp_dim"], cfg["patch_size"]
    n = (cfg["image_size"] // P) ** 2 + 1
    shapes = {"cls_token": (1, 1, D), "pos_embed": (1, n, D),
              "patch_embed.proj.weight": (D, 3, P, P), "patch_embed.proj.bias": (D,)}
    for i in range(cfg["depth"]):
        b = f"blocks.{i}."
        shapes.update({b + "norm1.weight": (D,), b + "norm1.bias": (D,),
                       b + "attn.qkv.weight": (3 * D, D), b + "attn.qkv.bias": (3 * D,),
                       b + "attn.proj.weight": (D, D), b + "attn.proj.bias": (D,),
                       b + "norm2.weight": (D,), b + "norm2.bias": (D,),
                       b + "mlp.fc1.weight": (M, D), b + "mlp.fc1.bias": (M,),
                       b + "mlp.fc2.weight": (D, M), b + "mlp.fc2.bias": (D,)})
    shapes.update({"norm.weight": (D,), "norm.bias": (D,),
                   "head.weight": (cfg["num_classes"], D), "head.bias": (cfg["num_classes"],)})
    return shapes
