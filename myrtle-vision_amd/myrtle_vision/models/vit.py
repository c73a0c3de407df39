"""Vision Transformer -- MI355X-native drop-in for the reference ``src/myrtle_vision/models/vit.py``.

Same classes (``Residual, PreNorm, FeedForward, Attention, Transformer, ViT, ClassificationDecoder,
SegmentationDecoder, DetectionDecoder``), constructor kwargs, attribute names (hence state-dict keys and the
checkpoint wire format), the ``attn_output`` hook point, ``.convert()`` and ``.quantizer`` as the reference; the
arithmetic runs in the HIP kernels of ``csrc/`` through ``myrtle_vision.hip.functional``.

One extension: ``ViT(..., precision="bf16" | "bf16x3" | "bf16x3h" | "fp32")`` (default: env ``MYRTLE_VISION_PRECISION`` or ``"bf16"``).
``bf16`` is the benchmark configuration (MFMA, fp32 accumulate, fp32 residual stream); ``fp32`` is the
parity mode (fp32-accurate arithmetic: the one held to 1e-3 / bit-exact argmax against the reference); ``bf16x3`` is the fast mode
inside that same tolerance (fp32 data flow, every nn.Linear product from two bf16 pieces per operand: 2^-16 relative); ``bf16x3h``
adds the attention core on IEEE-half operands (logits and arg-max as bf16x3, gradients to 1.6e-3; 2 750 vs 2 300 img/s).  Any fake-quantised ``q_format`` runs in ``fp32`` (its values are fp32 by definition,
utils/quantize.py:84).

There is no CPU compute path: ``forward`` on CPU tensors raises.
"""
import os
from contextlib import nullcontext
from typing import Optional, Union

import torch
import torch.autograd.profiler as profiler
import torch.nn.functional as TF
from torch import nn

from myrtle_vision.hip import functional as F
from myrtle_vision.hip import ops
from myrtle_vision.utils.quantize import DeQuantStub, FloatFunctional, ModelQuantizer, QFormat, QuantStub

MIN_NUM_PATCHES = 16  # reference vit.py:14


def _default_precision() -> str:
    return os.environ.get("MYRTLE_VISION_PRECISION", "bf16")


class _HipModule:
    """Mixin: the precision the HIP kernels run in (set for a whole model by ``ViT.set_precision``)."""

    precision = "bf16"

    @property
    def act_dtype(self):
        return ops.act_dtype(self.precision)


# ---- leaf modules: torch parameter containers whose forward runs on the HIP kernels ----------------------
class Linear(nn.Linear, _HipModule):
    """nn.Linear parameters/initialisation; forward = bf16-MFMA or fp32 HIP GEMM chosen by the input dtype."""

    def forward(self, x):
        ops.require_cuda(x, self.weight)
        if x.dtype != self.act_dtype:
            x = F.cast(x, self.act_dtype)
        with ops.segments(ops.prec_segments(self.precision)):          # bf16x3 / bf16x6 (only fp32 inputs look at it)
            return F.linear(x, self.weight, self.bias)


class LayerNorm(nn.LayerNorm, _HipModule):
    def forward(self, x):
        ops.require_cuda(x, self.weight)
        return F.layer_norm(x, self.weight, self.bias, self.act_dtype, self.eps)


class GELU(nn.GELU, _HipModule):
    def forward(self, x):
        ops.require_cuda(x)
        return F.gelu(x)


class Dropout(nn.Dropout):
    """nn.Dropout on the HIP Philox kernel (reference vit.py:50,52,75,311).  Every shipped config uses p = 0, where this is
    the identity and the enclosing block stays on its fused path; with p > 0 in training mode the block runs module by
    module (``fusable()`` is False) and the mask is regenerated in backward from its (seed, offset) pair."""

    def forward(self, x):
        return F.dropout(x, self.p, self.training)


def _plain(module, cls):
    """True if ``module`` is exactly our leaf class with no forward hooks (i.e. safe to fuse through)."""
    return type(module) is cls and not module._forward_hooks and not module._forward_pre_hooks


def _int8_linear(m):
    """The converted PyTorchINT8 Linear behind ``m`` -- ``Int8Linear`` itself or ModelQuantizer's
    ``Sequential(QuantStub [passthrough], Int8Linear)`` wrapper -- if nothing hooks into it; else None."""
    if isinstance(m, nn.Sequential) and len(m) == 2 and getattr(m[0], "passthrough", False):
        if m._forward_hooks or m._forward_pre_hooks or m[0]._forward_hooks or m[0]._forward_pre_hooks:
            return None
        m = m[1]
    if type(m).__name__ != "Int8Linear" or m._forward_hooks or m._forward_pre_hooks:
        return None
    return m


def _int8_fusable(x, norm, first, second, M):
    """Both Int8Linears of a converted block take int8 codes, nothing needs a gradient, the norm is a plain LayerNorm."""
    return (type(first).fuse_quant and not (torch.is_grad_enabled() and x.requires_grad) and x.is_cuda
            and x.dtype == torch.float32 and _plain(norm, LayerNorm) and norm.weight.shape[0] % 16 == 0
            and norm.weight.shape[0] <= 1024 and first.takes_codes(M) and second.takes_codes(M))


# ---- reference vit.py:17-27 ---------------------------------------------------------------------------------
class Residual(nn.Module):
    def __init__(self, fn: nn.Module):
        super().__init__()
        self.fn = fn
        self.res_add = FloatFunctional()

    def forward(self, x: torch.Tensor):
        fused = self._fused(x)
        if fused is not None:
            return fused
        return self.res_add.add(self.fn(x), x)

    def _fusable_plain(self):
        """True when this is a plain Residual(PreNorm(FeedForward)) that nothing observes (no hooks, no quantiser stubs)."""
        pn = self.fn
        return (type(pn) is PreNorm and _plain(pn.norm, LayerNorm) and self.res_add.plain() and not self._forward_hooks
                and not self._forward_pre_hooks and not pn._forward_hooks and not pn._forward_pre_hooks
                and type(pn.fn) is FeedForward and pn.fn.fusable() and not pn.fn._forward_hooks and not pn.fn._forward_pre_hooks)

    def _fused(self, x):
        """One HIP-fused call for Residual(PreNorm(Attention | FeedForward)) when nothing observes the insides."""
        pn = self.fn
        if not (type(pn) is PreNorm and _plain(pn.norm, LayerNorm) and self.res_add.plain()):
            return None
        if pn._forward_hooks or pn._forward_pre_hooks or x.dim() != 3:
            return None
        ops.require_cuda(x)
        inner = pn.fn
        prec = pn.norm.precision
        x = x.float()
        if type(inner) is Attention and inner.fusable():
            lin_o = inner.to_out[0]
            return F.attn_block(x, pn.norm.weight, pn.norm.bias, inner.to_qkv.weight, inner.to_qkv.bias, lin_o.weight,
                                lin_o.bias, inner.heads, inner.scale, prec)
        if type(inner) is FeedForward and inner.fusable():
            fc1, fc2 = inner.net[0], inner.net[3]
            return F.mlp_block(x, pn.norm.weight, pn.norm.bias, fc1.weight, fc1.bias, fc2.weight, fc2.bias, prec)
        # converted PyTorchINT8 blocks: the residual add rides in the second GEMM's epilogue, and without autograd every
        # quantiser is fused into the kernel that produces its input (int8_fused_forward)
        if type(inner) is Attention and inner.int8_pair() is not None:
            fused = inner.int8_fused_forward(x, pn.norm)
            return fused if fused is not None else inner.int8_forward(pn.norm(x), residual=x)
        if type(inner) is FeedForward and inner.int8_pair() is not None:
            fused = inner.int8_fused_forward(x, pn.norm)
            if fused is not None:
                return fused
            fc1, fc2 = inner.int8_pair()
            return fc2(fc1(pn.norm(x)), pre_gelu=True, residual=x)
        return None


# ---- reference vit.py:30-41 ---------------------------------------------------------------------------------
class PreNorm(nn.Module):
    def __init__(self, dim: int, fn: nn.Module):
        super().__init__()
        self.norm = LayerNorm(dim)
        self.fn = fn

    def forward(self, x: torch.Tensor):
        return self.fn(self.norm(x))


# ---- reference vit.py:44-56 ---------------------------------------------------------------------------------
class FeedForward(nn.Module):
    def __init__(self, dim: int, hidden_dim: int, dropout: float = 0.0):
        super().__init__()
        self.net = nn.Sequential(
            Linear(dim, hidden_dim),
            GELU(),
            Dropout(dropout),
            Linear(hidden_dim, dim),
            Dropout(dropout),
        )

    def fusable(self):
        n = self.net
        return (not self._forward_hooks and not n._forward_hooks and _plain(n[0], Linear) and _plain(n[1], GELU)
                and _plain(n[3], Linear) and all(type(n[i]) is Dropout and (n[i].p == 0.0 or not n[i].training)
                                                 and not n[i]._forward_hooks for i in (2, 4)))

    def int8_pair(self):
        """(fc1, fc2) when this is a converted PyTorchINT8 MLP nothing hooks into, else None."""
        n = self.net
        fc1, fc2 = _int8_linear(n[0]), _int8_linear(n[3])
        if (fc1 is not None and fc2 is not None and _plain(n[1], GELU) and not self._forward_hooks
                and not n._forward_hooks and not n._forward_pre_hooks
                and all(type(n[i]) is Dropout and (n[i].p == 0.0 or not n[i].training) and not n[i]._forward_hooks
                        for i in (2, 4))):
            return fc1, fc2
        return None

    def int8_fused_forward(self, x, norm):
        """Converted PyTorchINT8 MLP block without autograd, quantisers fused into their producers (bit-identical to the
        module path): LayerNorm writes fc1's int8 codes, fc1's epilogue applies GELU and fc2's quantiser and writes int8,
        fc2's epilogue adds the residual.  No fp32 activation but the residual stream touches memory.  None if n/a."""
        fc1, fc2 = self.int8_pair()
        B, T, D = x.shape
        M = B * T
        if (not _int8_fusable(x, norm, fc1, fc2, M)) or M % 256 != 0 or fc1.weight.shape[0] % 256 != 0:
            return None
        x = x.contiguous()
        x8 = ops.layernorm_q8(x, D, M, D, norm.weight, norm.bias, norm.eps, *fc1.qparams)
        h8 = fc1.forward_codes(x8, M, gelu_q8=fc2.qparams)
        return fc2.forward_codes(h8, M, residual=x).view(B, T, D)

    def forward(self, x: torch.Tensor):
        pair = self.int8_pair()
        if pair is not None:
            # converted PyTorchINT8 MLP: nn.GELU is applied inside fc2's input quantiser (one pass over the hidden
            # activations instead of GELU fp32 -> fp32 followed by quantise fp32 -> codes); same numbers
            return pair[1](pair[0](x), pre_gelu=True)
        return self.net(x)


# ---- reference vit.py:59-99 ---------------------------------------------------------------------------------
class Attention(nn.Module):
    def __init__(self, dim: int, heads: int = 8, dim_head: int = 64, dropout: float = 0.0):
        super().__init__()
        inner_dim = dim_head * heads
        self.heads = heads
        self.scale = dim_head ** -0.5

        self.to_qkv = Linear(dim, inner_dim * 3, bias=True)
        self.to_out = nn.Sequential(
            Linear(inner_dim, dim),
            Dropout(dropout),
        )

        self.dequant_qkv = DeQuantStub()
        self.quant_out = QuantStub()
        self.bf16_core = False             # set by ViT.convert(bf16_attention=True) (PyTorchINT8 only)
        # Identity layer kept so that a forward hook can collect attention maps (reference vit.py:80-82)
        self.attn_output = nn.Identity()

    def fusable(self):
        d = self.to_out[1]
        return (not self._forward_hooks and _plain(self.to_qkv, Linear) and _plain(self.to_out[0], Linear)
                and not self.to_out._forward_hooks and type(d) is Dropout and (d.p == 0.0 or not d.training)
                and not self.attn_output._forward_hooks and self.dequant_qkv.plain() and self.quant_out.plain())

    def int8_pair(self):
        """(to_qkv, to_out) when this is a converted PyTorchINT8 attention nothing hooks into, else None."""
        lq, lo = _int8_linear(self.to_qkv), _int8_linear(self.to_out[0])
        d = self.to_out[1]
        if (lq is not None and lo is not None and not self._forward_hooks and not self.to_out._forward_hooks
                and type(d) is Dropout and (d.p == 0.0 or not d.training) and not d._forward_hooks
                and not self.attn_output._forward_hooks and self.dequant_qkv.plain() and self.quant_out.plain()):
            return lq, lo
        return None

    def int8_fused_forward(self, x, norm):
        """Converted PyTorchINT8 attention block without autograd, quantisers fused into their producers (bit-identical to
        the module path): LayerNorm writes to_qkv's int8 codes, the exact-fp32 attention core writes to_out's, to_out's
        epilogue adds the residual.  None if not applicable."""
        lq, lo = self.int8_pair()
        B, T, D = x.shape
        M = B * T
        dh = lq.weight.shape[0] // (3 * self.heads)
        if (not _int8_fusable(x, norm, lq, lo, M)) or self.bf16_core or not ops.attention_f32_fused_supported(torch.float32, T, dh):
            return None
        x = x.contiguous()
        x8 = ops.layernorm_q8(x, D, M, D, norm.weight, norm.bias, norm.eps, *lq.qparams)
        qkv = lq.forward_codes(x8, M)
        o8 = ops.attention_fwd_f32_q8(qkv, B, T, self.heads, self.scale, *lo.qparams)
        return lo.forward_codes(o8.view(M, -1), M, residual=x).view(B, T, D)

    def int8_forward(self, x, residual=None):
        """Converted PyTorchINT8 attention in three GEMM-side fusions: to_qkv writes bf16 directly when the bf16 core is
        opted in, to_out's quantiser reads the attention output in whatever dtype it has, and the Residual add rides in
        to_out's epilogue.  Same numbers as the module-by-module path."""
        lq, lo = self.int8_pair()
        qkv = lq(x, out_dtype=torch.bfloat16 if self.bf16_core else torch.float32)
        return lo(F.attention_core(qkv, self.heads, self.scale, None), residual=residual)

    def forward(self, x: torch.Tensor):
        # reference vit.py:85-99; the reshape/permute/transpose of the reference are index arithmetic inside the
        # attention kernels (qkv stays [B, N, 3, H, dh])
        if self.int8_pair() is not None:
            return self.int8_forward(x)
        qkv = self.dequant_qkv(self.to_qkv(x))
        hook = self.attn_output if self.attn_output._forward_hooks else None
        if self.bf16_core and hook is None and qkv.dtype == torch.float32:
            # converted PyTorchINT8 model, opted in by ViT.convert(bf16_attention=True): q, k, v come out of an 8-bit
            # Linear and the result goes straight into the next 8-bit quantiser, so the fused bf16 kernel (fp32 softmax
            # statistics) is finer than either neighbour; the exact fp32 products it replaces are 45 % of the forward pass
            out = F.cast(F.attention_core(F.cast(qkv, torch.bfloat16), self.heads, self.scale, None), torch.float32)
        else:
            out = F.attention_core(qkv, self.heads, self.scale, hook)
        out = self.quant_out(out)
        out = self.to_out(out)
        return out


# ---- reference vit.py:102-161 -------------------------------------------------------------------------------
class Transformer(nn.Module):
    def __init__(self, dim: int, depth: int, heads: int, dim_head: int, mlp_dim: int, dropout: float, profile: bool):
        super().__init__()
        if profile:
            self.cm_attention = profiler.record_function("transformer:attention")
            self.cm_feedforward = profiler.record_function("transformer:feedforward")
        else:
            self.cm_attention = nullcontext()
            self.cm_feedforward = nullcontext()

        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(
                nn.Sequential(
                    Residual(PreNorm(dim, Attention(dim, heads=heads, dim_head=dim_head, dropout=dropout))),
                    Residual(PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout))),
                )
            )

    # Extension (off by default; ``ViT(..., prune_dead_tokens=True)`` switches it on for the classification decoder): the decoder
    # reads x[:, 0] only (reference vit.py:335-342) and the LAST block's FeedForward is a per-token function, so its output for
    # the 196 patch tokens is dead -- nothing reads it and its gradient is exactly zero.  With the flag the last FeedForward runs
    # on the cls rows alone ([B, 1, D]): identical logits, loss and parameter gradients (the same sums over fewer, non-zero rows),
    # 5.5 GFLOP per image (5 %) less work.  Applied only while nothing observes the block (no hooks, plain fusable modules).
    cls_only_tail = False

    def _tail_prunable(self, blk):
        ff = blk[1]
        return (self.cls_only_tail and ff._fusable_plain() and not blk._forward_hooks and not blk._forward_pre_hooks)

    def forward(self, x: torch.Tensor):
        last = len(self.layers) - 1
        for i, transformer_block in enumerate(self.layers):
            with self.cm_attention:
                x = transformer_block[0](x)
            with self.cm_feedforward:
                if i == last and x.dim() == 3 and self._tail_prunable(transformer_block):
                    x = x[:, :1]                  # the slice's backward scatters the cls gradient into zeros for the other rows
                x = transformer_block[1](x)
        return x


# ---- reference vit.py:164-323 -------------------------------------------------------------------------------
class ViT(nn.Module):
    def __init__(
        self,
        *,
        decoder: str,
        image_size: int,
        patch_size: int,
        num_classes: int,
        dim: int,
        depth: int,
        heads: int,
        mlp_dim: int,
        pool: str = "cls",
        channels: int = 3,
        dim_head: int = 64,
        dropout: float = 0.0,
        emb_dropout: float = 0.0,
        num_det_tokens: int = 100,
        profile: bool = False,
        q_format: Optional[Union[str, QFormat]] = None,
        precision: Optional[str] = None,
        prune_dead_tokens: Optional[bool] = None,
    ):
        super().__init__()
        assert image_size % patch_size == 0, "Image dimensions must be divisible by the patch size."
        num_patches = (image_size // patch_size) ** 2
        patch_dim = channels * patch_size ** 2
        assert num_patches > MIN_NUM_PATCHES, (
            f"your number of patches ({num_patches}) is way too small for "
            f"attention to be effective (at least 16). Try decreasing your "
            f"patch size"
        )
        assert decoder in {
            "classification",
            "segmentation",
            "detection",
        }, "decoder must be either classification, segmentation, or detection"
        if heads * dim_head != dim:
            # the reference reshapes with c_dim // heads (vit.py:88), so it only works when heads*dim_head == dim
            raise ValueError(f"heads * dim_head must equal dim (got {heads} * {dim_head} != {dim})")
        self.patch_size = patch_size

        if profile:
            self.cm_patch_to_embedding = profiler.record_function("patch_to_embedding")
            self.cm_transformer = profiler.record_function("transformer")
            self.cm_mlp_head = profiler.record_function("mlp_head")
        else:
            self.cm_patch_to_embedding = nullcontext()
            self.cm_transformer = nullcontext()
            self.cm_mlp_head = nullcontext()

        # parameters in the reference's registration order (vit.py:218-222): state-dict order is part of the format
        self.pos_embedding = nn.Parameter(torch.randn(1, 14 * 14 + 1, dim))
        self.pos_embedding_det = nn.Parameter(torch.randn(1, num_det_tokens, dim))
        self.patch_to_embedding = Linear(patch_dim, dim)
        self.cls_token = nn.Parameter(torch.randn(1, 1, dim))
        self.det_tokens = nn.Parameter(torch.randn(1, num_det_tokens, dim))
        self.dropout = Dropout(emb_dropout)

        self.transformer = Transformer(dim, depth, heads, dim_head, mlp_dim, dropout, profile)
        if prune_dead_tokens is None:
            prune_dead_tokens = os.environ.get("MYRTLE_VISION_PRUNE_DEAD_TOKENS", "0") == "1"
        self.transformer.cls_only_tail = bool(prune_dead_tokens) and decoder == "classification"

        if decoder == "classification":
            self.decoder = ClassificationDecoder(dim, num_classes)
        elif decoder == "segmentation":
            self.decoder = SegmentationDecoder(dim, num_classes, image_size, patch_size)
        elif decoder == "detection":
            self.decoder = DetectionDecoder(dim, num_classes, num_det_tokens)

        self.quant_img = QuantStub()
        self.quant_pos_embedding = QuantStub()
        self.quant_cls_token = QuantStub()
        self.quant_det_tokens = QuantStub()
        self.dequant_output = DeQuantStub()
        self.cls_token_cat = FloatFunctional()
        self.pos_embedding_add = FloatFunctional()
        self.pos_embedding_cat = FloatFunctional()
        self.set_precision(precision if precision is not None else _default_precision())
        self.quantizer = ModelQuantizer(self)
        self.quantizer.prepare_qat(q_format if q_format is not None else QFormat.FP32)

    # -- precision plumbing ------------------------------------------------------------------------------
    def set_precision(self, precision: str):
        ops.act_dtype(precision)  # validates
        self.precision = precision
        for m in self.modules():
            # quantised leaves (QATLinear, QLinear, QLayerNorm ...) were re-classed by ModelQuantizer and are no longer
            # _HipModule instances, but they carry the same attribute and must follow the model
            if isinstance(m, _HipModule) or hasattr(m, "precision"):
                m.precision = precision
        return self

    def unused_parameter_names(self):
        """Parameters that never reach the loss for the classification / segmentation decoders: the detection
        tokens are built but never concatenated (reference vit.py:285-290, ``self.decoder == "detection"`` is
        always False).  The optimizer and the gradient all-reduce leave them alone, exactly as
        torch.optim skips parameters whose ``.grad`` is None."""
        return ("pos_embedding_det", "det_tokens")

    # -- positional embedding (reference vit.py:292-302) ---------------------------------------------------
    def _pos_embedding(self, gh: int, gw: int) -> torch.Tensor:
        """cls slot + 14x14 grid bicubically resized to (gh, gw) (reference vit.py:292-302: F.interpolate(mode="bicubic",
        align_corners=False) on the (1, D, 14, 14) view).  At 224^2 the resize is the identity and the parameter is used
        as is.  Otherwise the resize -- a fixed linear map of the 196 grid positions -- is applied as ONE small fp32
        product with its [gh*gw, 196] matrix: torch's bicubic kernels parallelise over the 256 output pixels only and
        loop over the 768 channels (1.3 ms forward + 2.4 ms backward per step at 256^2, 6 % of the step)."""
        if gh == 14 and gw == 14:
            return self.pos_embedding
        cls_pos, grid = self.pos_embedding[:, 0:1, :], self.pos_embedding[0, 1:, :]
        if grid.is_cuda:
            out = F.linear(self._pos_resize_matrix(gh, gw, grid.device), grid.t().contiguous(), None)   # [gh*gw, D]
            return torch.cat((cls_pos, out.unsqueeze(0)), dim=1)
        grid = grid.unsqueeze(0).transpose(1, 2).reshape(1, -1, 14, 14)
        grid = TF.interpolate(grid, size=(gh, gw), mode="bicubic", align_corners=False)
        grid = grid.reshape(1, -1, gh * gw).transpose(1, 2)
        return torch.cat((cls_pos, grid), dim=1)

    def _pos_resize_matrix(self, gh: int, gw: int, device) -> torch.Tensor:
        """R [gh*gw, 196] with resize(grid)[p, :] = sum_q R[p, q] grid[q, :]: torch's own bicubic weights, obtained by
        resizing the 196 one-hot grids on the host once per (gh, gw)."""
        cache = self.__dict__.setdefault("_pos_resize_cache", {})
        key = (gh, gw, str(device))
        if key not in cache:
            eye = torch.eye(196, dtype=torch.float32).reshape(196, 1, 14, 14)
            r = TF.interpolate(eye, size=(gh, gw), mode="bicubic", align_corners=False)
            cache[key] = r.reshape(196, gh * gw).t().contiguous().to(device)
        return cache[key]

    def _embed_fusable(self):
        return (_plain(self.patch_to_embedding, Linear) and self.quant_img.plain() and self.quant_cls_token.plain()
                and self.quant_pos_embedding.plain() and self.cls_token_cat.plain() and self.pos_embedding_add.plain()
                and self.pos_embedding_cat.plain())

    def forward(self, img: torch.Tensor):
        with ops.segments(ops.prec_segments(self.precision)):
            x = self._backbone(img)
            with self.cm_mlp_head:
                output = self.decoder(x)
            output = self.dequant_output(output)
        return output

    def segmentation_loss(self, img: torch.Tensor, labels: torch.Tensor):
        """Extension: what segmentation/train.py:260-265 computes from ``vit(img)`` --
        ``CrossEntropyLoss()(outputs, labels)``, ``outputs.argmax(dim=1)`` and the pixel accuracy -- in one fused tail
        that never writes the [B, C, H, W] logits.  -> (loss, accuracy, pred uint8 [B, H, W]).  Falls back to the
        separate HIP kernels (same numbers, more HBM traffic) for shapes the fused kernels do not cover."""
        if not isinstance(self.decoder, SegmentationDecoder):
            raise ValueError("segmentation_loss needs decoder='segmentation'")
        with ops.segments(ops.prec_segments(self.precision)):
            x = self._backbone(img)
            with self.cm_mlp_head:
                if self.decoder.loss_fusable(labels):
                    return self.decoder.loss(x, labels)
                logits = self.dequant_output(self.decoder(x))
        loss = F.cross_entropy(logits, labels)
        pred = logits.detach().argmax(dim=1)
        return loss, (pred == labels).float().mean(), pred.to(torch.uint8)

    def _backbone(self, img: torch.Tensor):
        ops.require_cuda(img, self.pos_embedding)
        F.chain_reset()                       # no stale producer link from an earlier pass or a stand-alone block call
        b_dim, c_dim, h_dim, w_dim = img.shape
        p = self.patch_size
        gh, gw = h_dim // p, w_dim // p

        if self._embed_fusable():
            with self.cm_patch_to_embedding:
                x = F.patch_embed(img, self.patch_to_embedding.weight, self.patch_to_embedding.bias, self.cls_token,
                                  self._pos_embedding(gh, gw), p, self.precision)
        else:
            x = self._embed_unfused(img, gh, gw)
        x = self.dropout(x)

        with self.cm_transformer:
            x = self.transformer(x)
        return x

    def _embed_unfused(self, img, gh, gw):
        """reference vit.py:271-311 module by module (used when quantisers sit between the steps)."""
        b_dim = img.shape[0]
        p = self.patch_size
        x = ops.patchify(img, p, torch.float32).view(b_dim, gh * gw, -1)
        x = self.quant_img(x)
        with self.cm_patch_to_embedding:
            x = F.cast(self.patch_to_embedding(x), torch.float32)
        cls_tokens = self.quant_cls_token(self.cls_token.repeat(b_dim, 1, 1))
        # det tokens are built and never used for these decoders (reference vit.py:285-290)
        x = self.cls_token_cat.cat((cls_tokens, x), dim=1)
        pos = self.pos_embedding_cat.post(self._pos_embedding(gh, gw))   # the reference's cat of (cls slot, grid)
        return self.pos_embedding_add.add(x, self.quant_pos_embedding(pos.repeat(b_dim, 1, 1)))

    def convert(self, bf16_attention: bool = False) -> None:
        """reference vit.py ``convert()``.  Extension (off by default, PyTorchINT8 only): ``bf16_attention=True`` runs the
        attention core of the converted model on the fused bf16 kernel instead of the exact fp32 products -- a bf16-mode
        approximation (2^-9 relative on q, k, v and P) between two 8-bit quantisers, 1.6x faster end to end."""
        self.quantizer.convert()
        if bf16_attention:
            if not any(type(m).__name__ == "Int8Linear" for m in self.modules()):
                raise ValueError("bf16_attention applies to converted PyTorchINT8 models only")
            for m in self.modules():
                if isinstance(m, Attention):
                    m.bf16_core = True


# ---- reference vit.py:325-342 -------------------------------------------------------------------------------
class ClassificationDecoder(nn.Module):
    def __init__(self, dim, num_classes):
        super().__init__()
        self.norm = LayerNorm(dim)
        self.linear = Linear(dim, num_classes)

    def forward(self, x: torch.Tensor):
        if _plain(self.norm, LayerNorm) and _plain(self.linear, Linear) and x.dim() == 3:
            ops.require_cuda(x)
            return F.cls_head(x.float(), self.norm.weight, self.norm.bias, self.linear.weight, self.linear.bias,
                              self.norm.precision)
        x = x[:, 0]
        x = self.norm(x)
        x = self.linear(x)
        return F.cast(x, torch.float32)


# ---- reference vit.py:344-374 -------------------------------------------------------------------------------
class SegmentationDecoder(nn.Module):
    def __init__(self, dim, num_classes, image_size, patch_size):
        super().__init__()
        self.norm = LayerNorm(dim)
        self.linear = Linear(dim, num_classes)
        self.upsample = nn.Upsample(size=image_size, mode="bilinear")
        self.image_size = image_size
        self.image_size_in_patches = image_size // patch_size

    def forward(self, x: torch.Tensor):
        g = self.image_size_in_patches
        if x.shape[1] - 1 != g * g:
            raise ValueError(f"expected {g * g} patch tokens for image_size {self.image_size}, got {x.shape[1] - 1}")
        ops.require_cuda(x)
        if not (_plain(self.norm, LayerNorm) and _plain(self.linear, Linear)) or not isinstance(self.image_size, int):
            # module by module (reference vit.py:359-374): prepare_qat wrapped norm / linear as Sequential(QuantStub, module)
            # (utils/quantize.py:253-327), or something hooks into them
            y = self.linear(self.norm(x[:, 1:]))
            size = self.image_size if isinstance(self.image_size, int) else self.image_size[0]
            return F.upsample_bilinear(F.cast(y, torch.float32), g, size)
        return F.seg_head(x.float(), self.norm.weight, self.norm.bias, self.linear.weight, self.linear.bias, g,
                          self.image_size, self.norm.precision)

    def loss_fusable(self, labels) -> bool:
        g = self.image_size_in_patches
        return (_plain(self.norm, LayerNorm) and _plain(self.linear, Linear) and not self._forward_hooks
                and not self.upsample._forward_hooks and isinstance(self.image_size, int)
                and tuple(labels.shape[1:]) == (self.image_size, self.image_size)
                and ops.seg_ce_supported(self.linear.out_features, g, g, self.image_size))

    def loss(self, x: torch.Tensor, labels: torch.Tensor):
        """Fused decoder + mean cross entropy + argmax: (loss, pixel accuracy, pred uint8 [B, S, S])."""
        g = self.image_size_in_patches
        if x.shape[1] - 1 != g * g:
            raise ValueError(f"expected {g * g} patch tokens for image_size {self.image_size}, got {x.shape[1] - 1}")
        ops.require_cuda(x, labels)
        return F.seg_head_loss(x.float(), self.norm.weight, self.norm.bias, self.linear.weight, self.linear.bias, labels,
                               g, self.image_size, self.norm.precision)


# ---- reference vit.py:376-396 (detection is out of scope for the HIP path: SURVEY section 2 row 16) --------
class DetectionDecoder(nn.Module):
    def __init__(self, in_dim, num_classes, num_det_tokens):
        super().__init__()
        self.class_embed = Linear(in_dim, num_classes + 1)
        self.bbox_embed = Linear(in_dim, 4)
        self.num_det_tokens = num_det_tokens

    def forward(self, x: torch.Tensor):
        raise NotImplementedError(
            "the detection decoder is outside the MI355X hot-path scope (classification + segmentation only)")
