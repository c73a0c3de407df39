"""Autograd functions of the ViT hot path on the HIP kernels.

Granular functions (``layer_norm``, ``linear``, ``gelu``, ``attention_core``, ``patch_embed`` ...) mirror one
reference module each and compose freely (used when fake-quant stubs sit between modules, or when a submodule
is called on its own).  The fused functions ``attn_block`` / ``mlp_block`` implement one whole
``Residual(PreNorm(...))`` of the reference (vit.py:131-151) with everything fusable fused into GEMM epilogues:

    attn_block:  LN -> QKV GEMM(+bias) -> fused attention -> proj GEMM(+bias +residual)
    mlp_block :  LN -> fc1 GEMM(+bias, GELU, keeps pre-activation) -> fc2 GEMM(+bias +residual)
    backward  :  dX GEMMs (fc2's with the GELU' epilogue), dW GEMMs (transposed LDS reads, split-K),
                 attention backward, LN backward with the residual gradient folded in.

The residual stream and all gradients w.r.t. parameters are fp32; activations are bf16 (``prec='bf16'``) or fp32
(``prec='fp32'``: parity mode, fp32-accurate arithmetic).  Backward runs on autograd's worker thread: everything here is stateless
apart from caches keyed by tensor identity.
"""
import functools
import weakref

import torch
from torch.autograd import Function

from . import ops
from .ops import (EPI_DGELU, EPI_EMBED, EPI_GELU, EPI_GELU_GRAD, EPI_GELU_GRAD8, EPI_MUL, EPI_MUL8, EPI_NONE,
                  EPI_RESIDUAL)


GELU_GRAD_BITS = 8          # 8 | 16: width of the gelu'(h) the bf16 MLP block keeps for its backward pass


def _c(t):
    return t if t.is_contiguous() else t.contiguous()


def _scoped(backward):
    """Run one backward function inside ``ops.split_scope()``: its dX and dW products share the bf16x6 split of dY (fp32
    mode; a no-op for the bf16 path) -- and with the segment count its forward ran under (``_fwd``: 6 = bf16x6, 3 = bf16x3;
    backward runs on autograd's worker threads, which do not see the caller's thread-local scope)."""
    @functools.wraps(backward)
    def run(ctx, *args, **kwargs):
        with ops.split_scope(), ops.segments(getattr(ctx, "mv_segments", 6)):
            return backward(ctx, *args, **kwargs)
    return run


def _fwd(forward):
    """Forward of an autograd function that may hold fp32 Linear products: remember the segment count of the calling scope."""
    @functools.wraps(forward)
    def run(ctx, *args, **kwargs):
        # fused functions receive the model's precision as their last argument and follow it; granular ones (one reference
        # module each, no precision argument) follow the scope their caller opened (ViT.forward)
        prec = args[-1] if args and isinstance(args[-1], str) else None
        ctx.mv_segments = ops.prec_segments(prec) if prec is not None else getattr(ops._seg_tls, "n", 6)
        with ops.segments(ctx.mv_segments):
            return forward(ctx, *args, **kwargs)
    return run


# ------------------------------------------------------------------------------------------------------------
# granular functions
# ------------------------------------------------------------------------------------------------------------
class _LayerNorm(Function):
    """nn.LayerNorm over the last dim (vit.py:37).  Input fp32 (any leading shape), output ``out_dtype``."""

    @staticmethod
    @_fwd
    def forward(ctx, x, gamma, beta, out_dtype, eps):
        ops.require_cuda(x, gamma, beta)
        x = _c(x.float())
        dim = x.shape[-1]
        rows = x.numel() // dim
        y, mean, rstd = ops.layernorm_fwd(x, dim, rows, dim, gamma, beta, out_dtype, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.beta = beta
        return y.view(x.shape)

    @staticmethod
    @_scoped
    def backward(ctx, dy):
        x, gamma, mean, rstd = ctx.saved_tensors
        dim = x.shape[-1]
        rows = x.numel() // dim
        dy = _c(dy)
        dx = torch.empty_like(x)
        dg, db = ops.layernorm_bwd(dy, x, dim, gamma, mean, rstd, None, dx, dim, rows, dim, beta=ctx.beta)
        return dx, dg, db, None, None


def layer_norm(x, gamma, beta, out_dtype=torch.float32, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, out_dtype, eps)


class _Linear(Function):
    """nn.Linear: y = x W^T + b (vit.py:72,74,48,51,220,333).  x dtype selects the kernel family."""

    @staticmethod
    @_fwd
    def forward(ctx, x, weight, bias, out_dtype):
        ops.require_cuda(x, weight, bias)
        K = x.shape[-1]
        N = weight.shape[0]
        x2 = _c(x).view(-1, K)
        M = x2.shape[0]
        if x2.dtype == torch.bfloat16 and K % 8 != 0:
            raise RuntimeError(f"bf16 Linear needs in_features % 8 == 0 (got {K}); use precision='fp32'")
        out = torch.empty(M, N, dtype=out_dtype, device=x.device)
        ops.linear_fwd(x2, M, K, weight, bias, out, N)
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        ctx.bias = bias
        ctx.in_shape = x.shape
        return out.view(*x.shape[:-1], N)

    @staticmethod
    @_scoped
    def backward(ctx, dy):
        x2, weight = ctx.saved_tensors
        M, K = x2.shape
        N = weight.shape[0]
        dy2 = _c(dy).view(M, N)
        if dy2.dtype != x2.dtype:
            dy2 = ops.cast(dy2, x2.dtype)
        ld_dy = N
        if dy2.dtype == torch.bfloat16 and N % 8 != 0:      # pad the contraction dim of dX / rows of dW^T with zeros
            ld_dy = ops.pad8(N)
            padded = torch.zeros(M, ld_dy, dtype=dy2.dtype, device=dy2.device)
            padded[:, :N] = dy2
            dy2 = padded
        # dW first: in fp32 mode its split of dY also leaves the bias gradient, and the dX product below reuses that split
        dw, db = ops.linear_dw(dy2, x2, M, N, K, ld_dy=ld_dy, want_bias=ctx.has_bias, weight=weight, bias=ctx.bias)
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=x2.dtype, device=x2.device)
            ops.linear_dx(dy2, M, N, weight, dx, K, ld_dy=ld_dy)
            dx = dx.view(ctx.in_shape)
        return dx, dw, db, None


def linear(x, weight, bias, out_dtype=None):
    return _Linear.apply(x, weight, bias, x.dtype if out_dtype is None else out_dtype)


class _LinearQatF16(Function):
    """nn.qat.Linear of the FP16 formats (utils/quantize.py:253-327) when its input already holds float_quantize(5, 10)
    values: y = x W_q^T + b with W_q = weight_fake_quant(W).  Both operands are exactly representable in IEEE half, so the
    forward product runs on the f16 matrix cores with fp32 accumulation -- the same numbers as the reference's fp32 GEMM of
    the same values up to summation order.  Backward is the straight-through estimator's: dX = dY W_q, dW = dY^T x in fp32
    (dY is not quantised, so those products stay on the f32 MFMA)."""

    @staticmethod
    @_fwd
    def forward(ctx, x, weight, bias):
        ops.require_cuda(x, weight, bias)
        K, N = x.shape[-1], weight.shape[0]
        x2 = _c(x.float()).view(-1, K)
        M = x2.shape[0]
        x16 = ops.quant_float_f16(x2)                          # idempotent on already-quantised values: an exact conversion
        w16 = ops.quant_float_f16(weight)                      # weight_fake_quant and the half conversion in one pass
        out = torch.empty(M, N, dtype=torch.float32, device=x.device)
        ops.linear_f16(x16, w16, M, N, K, bias, out)
        ctx.save_for_backward(x2, ops.quant_float(weight, 5, 10))
        ctx.params = (weight, bias)
        ctx.in_shape = x.shape
        return out.view(*x.shape[:-1], N)

    @staticmethod
    @_scoped
    def backward(ctx, dy):
        x2, wq = ctx.saved_tensors
        weight, bias = ctx.params
        M, K = x2.shape
        N = wq.shape[0]
        dy2 = _c(dy.float()).view(M, N)
        dw, db = ops.linear_dw(dy2, x2, M, N, K, want_bias=bias is not None, weight=weight, bias=bias)   # dW first: see _Linear
        dx = None
        if ctx.needs_input_grad[0]:
            dx = torch.empty(M, K, dtype=torch.float32, device=dy.device)
            ops.linear_dx(dy2, M, N, wq, dx, K)
            dx = dx.view(ctx.in_shape)
        return dx, dw, db


def linear_qat_f16(x, weight, bias):
    return _LinearQatF16.apply(x, weight, bias)


class _Gelu(Function):
    @staticmethod
    @_fwd
    def forward(ctx, x):
        ops.require_cuda(x)
        x = _c(x)
        ctx.save_for_backward(x)
        return ops.gelu_fwd(x)

    @staticmethod
    @_scoped
    def backward(ctx, dy):
        (x,) = ctx.saved_tensors
        dy = _c(dy)
        if dy.dtype != x.dtype:
            dy = ops.cast(dy, x.dtype)
        return ops.gelu_bwd(x, dy)


def gelu(x):
    return _Gelu.apply(x)


class _Cast(Function):
    @staticmethod
    @_fwd
    def forward(ctx, x, dtype):
        ctx.src_dtype = x.dtype
        return ops.cast(_c(x), dtype)

    @staticmethod
    @_scoped
    def backward(ctx, dy):
        return ops.cast(_c(dy), ctx.src_dtype), None


def cast(x, dtype):
    return x if x.dtype == dtype else _Cast.apply(x, dtype)


class _Add(Function):
    """FloatFunctional.add of the residual (vit.py:27) in the unfused path."""

    @staticmethod
    @_fwd
    def forward(ctx, a, b):
        ops.require_cuda(a, b)
        return ops.add_f32(_c(a.float()), _c(b.float()))

    @staticmethod
    @_scoped
    def backward(ctx, d):
        return d, d


def add(a, b):
    return _Add.apply(a, b)


class _Dropout(Function):
    """nn.Dropout(p) in training mode (vit.py:50,52,75,311).  The mask is a function of (seed, offset) drawn from torch's
    CPU generator in forward -- reproducible under ``seed_everything`` -- and regenerated in backward, never stored."""

    @staticmethod
    @_fwd
    def forward(ctx, x, p):
        ops.require_cuda(x)
        seed, offset = (int(v) for v in torch.randint(0, 2 ** 62, (2,), dtype=torch.int64).tolist())
        ctx.rng = (float(p), seed, offset)
        return ops.dropout(_c(x), p, seed, offset)

    @staticmethod
    @_scoped
    def backward(ctx, dy):
        p, seed, offset = ctx.rng
        return ops.dropout(_c(dy), p, seed, offset), None


def dropout(x, p, training=True):
    if p == 0.0 or not training:
        return x
    if not 0.0 <= p < 1.0:
        raise ValueError(f"dropout probability has to be in [0, 1), got {p}")
    if x.dtype not in (torch.float32, torch.bfloat16):
        x = x.float()
    return _Dropout.apply(x, p)


class _AttentionFused(Function):
    """softmax(q k^T * scale) v on the fused MFMA kernel.  qkv bf16 [B, N, 3*H*64] -> [B, N, H*64]."""

    @staticmethod
    @_fwd
    def forward(ctx, qkv, heads, scale):
        B, N, three_d = qkv.shape
        qkv = _c(qkv)
        out, lse = ops.attention_fwd(qkv, B, N, heads, scale)
        ctx.save_for_backward(qkv, out, lse)
        ctx.heads, ctx.scale = heads, scale
        return out

    @staticmethod
    @_scoped
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        B, N, _ = qkv.shape
        dout = _c(dout)
        if dout.dtype != torch.bfloat16:
            dout = ops.cast(dout, torch.bfloat16)
        return ops.attention_bwd(qkv, out, dout, lse, B, N, ctx.heads, ctx.scale), None, None


class _AttentionProbs(Function):
    """Materialised fp32 path, part 1: probs = softmax(q k^T * scale)  [B, H, N, N] (vit.py:92-93)."""

    @staticmethod
    @_fwd
    def forward(ctx, qkv, heads, scale):
        B, N, three_d = qkv.shape
        dh = three_d // (3 * heads)
        qkv = _c(qkv)
        probs = ops.attention_probs_fp32(qkv, B, N, heads, dh, scale)
        ctx.save_for_backward(qkv, probs)
        ctx.heads, ctx.scale, ctx.dh = heads, scale, dh
        return probs

    @staticmethod
    @_scoped
    def backward(ctx, dprobs):
        qkv, probs = ctx.saved_tensors
        B, N, _ = qkv.shape
        H, dh, D = ctx.heads, ctx.dh, ctx.heads * ctx.dh
        L = ops.lib()
        dS = torch.empty_like(probs)
        dprobs = _c(dprobs)
        ops.check(L.mv_softmax_bwd(probs.data_ptr(), dprobs.data_ptr(), dS.data_ptr(), B * H * N, N, ctx.scale, ops._s()),
                  "softmax_bwd")
        dqkv = torch.zeros_like(qkv)
        flat, dflat = qkv.view(-1), dqkv.view(-1)
        bq, hq, bp, hp = N * 3 * D, dh, H * N * N, N * N
        ops.check(L.mv_gemm_f32(dS.data_ptr(), N, 1, bp, hp, flat[D:].data_ptr(), 3 * D, 1, bq, hq, dqkv.data_ptr(), 3 * D, 1,
                                bq, hq, N, dh, N, B, H, 1.0, 0, None, EPI_NONE, None, 0, 0, None, 0, ops._s()), "gemm_f32(dQ)")
        ops.check(L.mv_gemm_f32(dS.data_ptr(), 1, N, bp, hp, qkv.data_ptr(), 3 * D, 1, bq, hq, dflat[D:].data_ptr(), 3 * D, 1,
                                bq, hq, N, dh, N, B, H, 1.0, 0, None, EPI_NONE, None, 0, 0, None, 0, ops._s()), "gemm_f32(dK)")
        return dqkv, None, None


class _AttentionPV(Function):
    """Materialised fp32 path, part 2: out = (probs @ v).transpose(1,2).reshape(B,N,C) (vit.py:96)."""

    @staticmethod
    @_fwd
    def forward(ctx, probs, qkv, heads):
        B, N, three_d = qkv.shape
        dh = three_d // (3 * heads)
        probs, qkv = _c(probs), _c(qkv)
        ctx.save_for_backward(probs, qkv)
        ctx.heads, ctx.dh = heads, dh
        return ops.attention_pv_fp32(probs, qkv, B, N, heads, dh)

    @staticmethod
    @_scoped
    def backward(ctx, dout):
        probs, qkv = ctx.saved_tensors
        B, N, _ = qkv.shape
        H, dh, D = ctx.heads, ctx.dh, ctx.heads * ctx.dh
        L = ops.lib()
        dout = _c(dout.float())
        flat = qkv.view(-1)
        bq, hq, bp, hp, bo, ho = N * 3 * D, dh, H * N * N, N * N, N * D, dh
        dP = torch.empty_like(probs)
        ops.check(L.mv_gemm_f32(dout.data_ptr(), D, 1, bo, ho, flat[2 * D:].data_ptr(), 1, 3 * D, bq, hq, dP.data_ptr(), N, 1,
                                bp, hp, N, N, dh, B, H, 1.0, 0, None, EPI_NONE, None, 0, 0, None, 0, ops._s()), "gemm_f32(dP)")
        dqkv = torch.zeros_like(qkv)
        ops.check(L.mv_gemm_f32(probs.data_ptr(), 1, N, bp, hp, dout.data_ptr(), D, 1, bo, ho,
                                dqkv.view(-1)[2 * D:].data_ptr(), 3 * D, 1, bq, hq, N, dh, N, B, H, 1.0, 0, None, EPI_NONE,
                                None, 0, 0, None, 0, ops._s()), "gemm_f32(dV)")
        return dP, dqkv, None


class _AttentionFusedF32(Function):
    """The fp32 attention core with gradients, fused both ways (no [B, H, N, N] tensor): saves qkv, out and the
    log-sum-exp; the backward kernel recomputes the probabilities."""

    @staticmethod
    def forward(ctx, qkv, heads, scale):
        B, N, _ = qkv.shape
        qkv = _c(qkv)
        out, lse = ops.attention_fwd_f32_lse(qkv, B, N, heads, scale)
        ctx.save_for_backward(qkv, out, lse)
        ctx.cfg = (heads, scale)
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, out, lse = ctx.saved_tensors
        heads, scale = ctx.cfg
        B, N, _ = qkv.shape
        return ops.attention_bwd_f32_fused(qkv, out, _c(dout.float()), lse, B, N, heads, scale), None, None


def attention_core(qkv, heads, scale, probs_hook=None):
    """Attention.forward lines vit.py:87-96 on a to_qkv output [B, N, 3*heads*dh].

    ``probs_hook`` (callable or None) is the reference's ``attn_output`` Identity (vit.py:80-82,94): when it has
    forward hooks the probabilities are materialised (fp32) and passed through it; otherwise bf16 inputs take the
    fused kernel."""
    B, N, three_d = qkv.shape
    dh = three_d // (3 * heads)
    if probs_hook is None and ops.attention_fused_supported(qkv.dtype, N, dh):
        return _AttentionFused.apply(qkv, heads, scale)
    if (probs_hook is None and ops.attention_f32_fused_supported(qkv.dtype, N, dh)
            and not (torch.is_grad_enabled() and qkv.requires_grad)):
        # no gradient wanted (converted int8 model, fp32 evaluation): exact fp32 arithmetic without the probabilities
        return ops.attention_fwd_f32(_c(qkv), B, N, heads, scale)
    if probs_hook is None and ops.attention_f32_fused_supported(qkv.dtype, N, dh):
        return _AttentionFusedF32.apply(qkv, heads, scale)          # fp32 with gradients: fused forward + backward
    src_dtype = qkv.dtype
    q32 = cast(qkv, torch.float32)
    probs = _AttentionProbs.apply(q32, heads, scale)
    if probs_hook is not None:
        probs = probs_hook(probs)
    out = _AttentionPV.apply(probs, q32, heads)
    return cast(out, src_dtype)


class _PatchEmbed(Function):
    """patchify + patch_to_embedding + cls token + positional embedding (vit.py:271-311) -> fp32 [B, T, D]."""

    @staticmethod
    @_fwd
    def forward(ctx, img, weight, bias, cls_token, pos, patch, prec):
        ops.require_cuda(img, weight, bias, cls_token, pos)
        adt = ops.act_dtype(prec)
        B, C, H, W = img.shape
        npatch = (H // patch) * (W // patch)
        T, D, pd = npatch + 1, weight.shape[0], weight.shape[1]
        if adt == torch.bfloat16 and pd % 8 != 0:
            raise RuntimeError(f"bf16 patch embedding needs patch_dim % 8 == 0 (got {pd}); use precision='fp32'")
        patches = ops.patchify(img, patch, adt)
        x = torch.empty(B, T, D, dtype=torch.float32, device=img.device)
        pos2 = _c(pos.detach().float()).view(T, D)
        ops.linear_fwd(patches, B * npatch, pd, weight, bias, x, D, epi=EPI_EMBED, aux=pos2, ld_aux=D, aux_i=npatch)
        ops.embed_cls(_c(cls_token.detach()).view(D), pos2, x, B, T, D)
        ctx.save_for_backward(patches)
        ctx.dims = (B, T, D, pd, npatch)
        ctx.pos_shape, ctx.cls_shape = pos.shape, cls_token.shape
        ctx.params = (weight, bias, cls_token, pos if pos.shape[-2:] == (T, D) and pos.is_leaf else None)
        chain_reset()
        return x

    @staticmethod
    @_scoped
    def backward(ctx, dx):
        (patches,) = ctx.saved_tensors
        B, T, D, pd, npatch = ctx.dims
        dx = _c(dx.float())
        weight, bias, cls_token, pos = ctx.params
        dpos, dcls, dy = ops.embed_bwd_gather(dx, B, T, D, patches.dtype, pos=pos, cls_token=cls_token)   # one pass over dx
        dw, db = ops.linear_dw(dy, patches, B * npatch, D, pd, weight=weight, bias=bias)
        return None, dw, db, dcls.view(ctx.cls_shape), dpos.view(ctx.pos_shape), None, None


def patch_embed(img, weight, bias, cls_token, pos, patch, prec):
    if img.requires_grad:
        raise RuntimeError("gradient w.r.t. the input image is not implemented (the reference never needs it)")
    return _PatchEmbed.apply(img, weight, bias, cls_token, pos, patch, prec)


# ------------------------------------------------------------------------------------------------------------
# fused transformer blocks
# ------------------------------------------------------------------------------------------------------------
# One-slot side channel along the backward chain: the LayerNorm-backward kernel that produces a block's input
# gradient dx can also emit (a) dx in bf16 and (b) its column sums -- exactly what the NEXT block function to run
# (the one whose output gradient is this dx) needs as MFMA operand and as output-projection bias gradient.  Keeping
# a strong reference to dx pins its address, so a pointer match identifies the tensor unambiguously.
_side = {}


def _publish_side(dx, dx16, colsum):
    _side.clear()
    _side[dx.data_ptr()] = (dx, dx16, colsum)


def _take_side(dout, rows, dim):
    ent = _side.pop(dout.data_ptr(), None)
    _side.clear()
    if ent is None or ent[0].shape != dout.shape or ent[1].shape[0] != rows or ent[1].shape[1] not in (dim, ops.current_segments() * dim):
        return None, None
    return ent[1], ent[2]


def _ln_bwd_with_side(dy, x, D, g, b, mean, rstd, dout, M, adt, up_bias=None):
    """LayerNorm backward + residual gradient; in bf16 mode also publishes the bf16 copy and column sums of dx.
    ``up_bias``: the bias of the Linear that produced x (recorded in forward, ``_chain``): the column sums ARE its
    gradient, so they are written straight to its destination."""
    dx = torch.empty_like(x)
    if adt == torch.bfloat16:
        dx16 = torch.empty(M, D, dtype=torch.bfloat16, device=x.device)
        cs = ops.grad_out(up_bias, (D,), x.device)
        dg, db = ops.layernorm_bwd(dy, x, D, g, mean, rstd, dout, dx, D, M, D, dx16=dx16, dx_colsum=cs, beta=b)
        _publish_side(dx, dx16, cs)
    elif D <= 1024 and ops.x6_block_ok(M, D):
        # split-operand modes: dx also leaves as the pieces the consumer's dW / dX products read, with its column sums (that Linear's
        # bias gradient): the consumer (the block before this one, _take_side) then needs no split pass over dx
        dx6 = ops._split_buffer(M, D, x.device)
        cs = ops.grad_out(up_bias, (D,), x.device)
        dg, db = ops.layernorm_bwd(dy, x, D, g, mean, rstd, dout, dx, D, M, D, dx_split=dx6, dx_colsum=cs, beta=b)
        _publish_side(dx, dx6, cs)
    else:
        dg, db = ops.layernorm_bwd(dy, x, D, g, mean, rstd, dout, dx, D, M, D, beta=b)
    return dx, dg, db


# Forward-order link between consecutive block functions: (the tensor OBJECT a block returned, held weakly, and the bias of
# the Linear that produced it).  The next block reads it to learn whose bias gradient the column sums of its dx are.
# Identity, not address: a freed-and-reallocated buffer at the same address can never inherit the link.
_chain = [None]


def _chain_set(out, bias):
    _chain[0] = (weakref.ref(out), bias) if bias is not None else None


def _chain_take(x):
    c = _chain[0]
    return c[1] if c is not None and c[0]() is x else None


def chain_reset():
    """Forget the link (start of a forward pass, or a block called on its own)."""
    _chain[0] = None



class _AttnBlock(Function):
    """x + to_out(attention(to_qkv(LN(x))))  ==  Residual(PreNorm(dim, Attention)) (vit.py:131-141, 84-99)."""

    @staticmethod
    @_fwd
    def forward(ctx, x, g, b, wqkv, bqkv, wo, bo, heads, scale, prec):
        adt = ops.act_dtype(prec)
        B, T, D = x.shape
        M = B * T
        x = _c(x)
        ctx.up_bias = _chain_take(x)
        inner3 = wqkv.shape[0]
        inner = inner3 // 3
        ctx.x6 = adt == torch.float32 and ops.x6_block_ok(M, D, inner3, inner)
        if not (ctx.x6 and D <= 1024):
            y, mean, rstd = ops.layernorm_fwd(x, D, M, D, g, b, adt)
        if ctx.x6:
            # fp32 mode, every Linear product on the bf16x6 path: the block keeps the SPLITS of LN(x) and of the attention
            # output (what the forward and the dW products both read), not the tensors themselves
            dh = inner // heads
            # LayerNorm writes the pieces itself (no fp32 y, no split pass)
            y6, mean, rstd = ops.layernorm_fwd_split(x, D, M, D, g, b) if D <= 1024 else (ops.split_ex(y, M, D), mean, rstd)
            ctx.f16 = ops.attention_f16_supported(adt, T, dh)
            # precision "bf16x3h": the fused attention kernels on half operands (2^-12 per rounding, fp32 sums, softmax and outputs);
            # q / k / v leave the to_qkv product as half (one rounding of the fp32 accumulator + bias, no fp32 tensor, no cast pass)
            qkv = torch.empty(B, T, inner3, dtype=torch.float16 if ctx.f16 else adt, device=x.device)
            ops.nt_x6(y6, wqkv, "fwd", M, qkv.view(M, inner3), bias=bqkv)
            o = None
            if ctx.f16:
                o, probs = ops.attention_fwd_f16(qkv, B, T, heads, scale)            # "probs" slot: the log-sum-exp [B, H, T]
                ctx.fused32 = True
            elif ops.attention_f32_fused_supported(adt, T, dh):
                if any(ctx.needs_input_grad):
                    o, probs = ops.attention_fwd_f32_lse(qkv, B, T, heads, scale)   # "probs" slot: the log-sum-exp [B, H, T]
                else:
                    o, probs = ops.attention_fwd_f32(qkv, B, T, heads, scale), x.new_empty(0)   # evaluation
                ctx.fused32 = True
            else:
                probs = ops.attention_probs_fp32(qkv, B, T, heads, dh, scale)
                o = ops.attention_pv_fp32(probs, qkv, B, T, heads, dh)
                ctx.fused32 = False
            o6 = ops.split_ex(o.view(M, inner), M, inner)
            out = torch.empty_like(x)
            ops.nt_x6(o6, wo, "fwd", M, out.view(M, D), bias=bo, residual=x.view(M, D))
            # the fused backward needs the attention output itself (delta = rowsum(dO * O)), not only its split
            ctx.save_for_backward(x, g, mean, rstd, y6, qkv, o6, probs, wqkv, wo, *((o,) if ctx.fused32 else ()))
            ctx.cfg = (heads, scale, False)
            ctx.small = (b, bqkv, bo)
            _chain_set(out, bo)
            return out
        qkv = torch.empty(B, T, inner3, dtype=adt, device=x.device)
        ops.linear_fwd(y, M, D, wqkv, bqkv, qkv, inner3)
        dh = inner // heads
        if ops.attention_fused_supported(adt, T, dh):
            o, lse = ops.attention_fwd(qkv, B, T, heads, scale)
            probs = None
        elif ops.attention_f32_fused_supported(adt, T, dh) and not any(ctx.needs_input_grad):
            # fp32 evaluation: nothing will run backward, so the probabilities need not exist
            o, lse, probs = ops.attention_fwd_f32(qkv, B, T, heads, scale), None, None
        elif ops.attention_f32_fused_supported(adt, T, dh):
            # fp32 training: fused forward + backward kernels, the log-sum-exp rides in the probabilities' slot
            o, probs = ops.attention_fwd_f32_lse(qkv, B, T, heads, scale)
            lse, ctx.fused32 = None, True
        else:                                   # materialised fp32 probabilities (shapes the fused kernels lack)
            q32 = ops.cast(qkv, torch.float32)
            probs = ops.attention_probs_fp32(q32, B, T, heads, dh, scale)
            o = ops.cast(ops.attention_pv_fp32(probs, q32, B, T, heads, dh), adt)
            lse = None
        out = torch.empty_like(x)
        ops.linear_fwd(o.view(M, inner), M, inner, wo, bo, out, D, epi=EPI_RESIDUAL, aux=x, ld_aux=D)
        if lse is None and probs is None:        # evaluation-only path above
            probs = x.new_empty(0)
        ctx.save_for_backward(x, g, mean, rstd, y, qkv, o, lse if lse is not None else probs, wqkv, wo)
        ctx.cfg = (heads, scale, lse is not None)
        ctx.small = (b, bqkv, bo)
        _chain_set(out, bo)
        return out

    @staticmethod
    @_scoped
    def backward(ctx, dout):
        x, g, mean, rstd, y, qkv, o, lse_or_probs, wqkv, wo = ctx.saved_tensors[:10]
        heads, scale, fused = ctx.cfg
        B, T, D = x.shape
        M = B * T
        inner3 = wqkv.shape[0]
        inner = inner3 // 3
        if ctx.x6:
            y6, o6, probs = y, o, lse_or_probs
            dout = _c(dout)
            b, bqkv, bo = ctx.small
            d6, dbo = _take_side(dout, M, D)                                  # the producing LayerNorm backward left both
            if d6 is None or d6.dtype != torch.bfloat16 or d6.shape[1] != ops.current_segments() * D:
                dbo = ops.grad_out(bo, (D,), x.device)
                d6 = ops.split_ex(dout.view(M, D), M, D, colsum_out=dbo)      # one pass: the split and the bias gradient
            dwo = ops.tn_x6(d6, o6, M, wo)
            do = torch.empty(B, T, inner, dtype=torch.float32, device=x.device)
            ops.nt_x6(d6, wo, "dx", M, do.view(M, inner))
            dbqkv = ops.grad_out(bqkv, (inner3,), x.device)
            if ctx.f16:
                # the kernel writes the pieces of dqkv and per-image column sums itself: no fp32 dqkv, no split pass
                part = torch.empty(B, inner3, dtype=torch.float32, device=x.device)
                dq6 = ops.attention_bwd_f16(qkv, ctx.saved_tensors[10], do, probs, B, T, heads, scale, split=True, colsum=part)
                ops.colsum(part, B, inner3, inner3, dbqkv)
            else:
                if ctx.fused32:
                    dqkv = ops.attention_bwd_f32_fused(qkv, ctx.saved_tensors[10], do, probs, B, T, heads, scale)
                else:
                    dqkv = ops.attention_bwd_fp32(probs, qkv, do, B, T, heads, inner // heads, scale)
                dq6 = ops.split_ex(dqkv.view(M, inner3), M, inner3, colsum_out=dbqkv)
            dwqkv = ops.tn_x6(dq6, y6, M, wqkv)
            dy = torch.empty(M, D, dtype=torch.float32, device=x.device)
            ops.nt_x6(dq6, wqkv, "dx", M, dy)
            dx, dg, db = _ln_bwd_with_side(dy, x, D, g, b, mean, rstd, dout, M, torch.float32, ctx.up_bias)
            return dx, dg, db, dwqkv, dbqkv, dwo, dbo, None, None, None
        adt = y.dtype
        dout = _c(dout)
        d_act, dbo = _take_side(dout, M, D) if adt == torch.bfloat16 else (None, None)
        if d_act is None:
            d_act = ops.cast(dout, adt).view(M, D)                # dY of the projection, activation dtype
        b, bqkv, bo = ctx.small
        dwo, dbo2 = ops.linear_dw(d_act, o.view(M, inner), M, D, inner, want_bias=dbo is None, weight=wo, bias=bo)
        dbo = dbo if dbo is not None else dbo2
        do = torch.empty(B, T, inner, dtype=adt, device=x.device)
        ops.linear_dx(d_act, M, D, wo, do, inner)
        dbqkv = None
        if fused:
            # the kernel also leaves per-image column sums of dqkv: to_qkv's bias gradient without another pass over dqkv
            part = torch.empty(B, inner3, dtype=torch.float32, device=x.device)
            dqkv = ops.attention_bwd(qkv, o, do, lse_or_probs, B, T, heads, scale, colsum=part)
            dbqkv = ops.colsum(part, B, inner3, inner3, ops.grad_out(bqkv, (inner3,), x.device))
        elif getattr(ctx, "fused32", False):
            dqkv = ops.attention_bwd_f32_fused(qkv, o, do, lse_or_probs, B, T, heads, scale)
        else:
            dqkv = ops.cast(ops.attention_bwd_fp32(lse_or_probs, ops.cast(qkv, torch.float32), ops.cast(do, torch.float32),
                                                   B, T, heads, inner // heads, scale), adt)
        dwqkv, dbq2 = ops.linear_dw(dqkv.view(M, inner3), y, M, inner3, D, want_bias=dbqkv is None, weight=wqkv, bias=bqkv)
        dbqkv = dbqkv if dbqkv is not None else dbq2
        dy = torch.empty(M, D, dtype=adt, device=x.device)
        ops.linear_dx(dqkv.view(M, inner3), M, inner3, wqkv, dy, D)
        dx, dg, db = _ln_bwd_with_side(dy, x, D, g, b, mean, rstd, dout, M, adt, ctx.up_bias)   # + residual gradient
        return dx, dg, db, dwqkv, dbqkv, dwo, dbo, None, None, None


def attn_block(x, g, b, wqkv, bqkv, wo, bo, heads, scale, prec):
    return _AttnBlock.apply(x, g, b, wqkv, bqkv, wo, bo, heads, scale, prec)


class _MlpBlock(Function):
    """x + fc2(gelu(fc1(LN(x))))  ==  Residual(PreNorm(dim, FeedForward)) (vit.py:142-151, 44-56)."""

    @staticmethod
    @_fwd
    def forward(ctx, x, g, b, w1, b1, w2, b2, prec):
        adt = ops.act_dtype(prec)
        B, T, D = x.shape
        M = B * T
        Hd = w1.shape[0]
        x = _c(x)
        ctx.up_bias = _chain_take(x)
        ctx.x6 = adt == torch.float32 and ops.x6_block_ok(M, D, Hd)
        if not (ctx.x6 and D <= 1024):
            y, mean, rstd = ops.layernorm_fwd(x, D, M, D, g, b, adt)
        if ctx.x6:
            # fp32 mode on the bf16x6 path: keeps the splits of LN(x) and of gelu(h) plus the pre-activation h; the fp32
            # activation gelu(h) is never stored (the split kernel applies the GELU on its way); LayerNorm writes its pieces itself
            y6, mean, rstd = ops.layernorm_fwd_split(x, D, M, D, g, b) if D <= 1024 else (ops.split_ex(y, M, D), mean, rstd)
            h = torch.empty(M, Hd, dtype=adt, device=x.device)
            if ops.nt_split_ok(M, Hd, D):
                a6 = ops.nt_x6_gelu_split(y6, w1, M, h, bias=b1)     # one launch: h (fp32) and the pieces of gelu(h)
            else:
                ops.nt_x6(y6, w1, "fwd", M, h, bias=b1)
                a6 = ops.split_ex(h, M, Hd, op=1)
            out = torch.empty_like(x)
            ops.nt_x6(a6, w2, "fwd", M, out.view(M, D), bias=b2, residual=x.view(M, D))
            ctx.save_for_backward(x, g, mean, rstd, y6, h, a6, w1, w2)
            ctx.small = (b, b1, b2)
            _chain_set(out, b2)
            return out
        # bf16: the fc1 epilogue leaves gelu'(pre-activation) for the backward pass (one multiply there instead of another
        # erf evaluation per element) -- as ONE BYTE per element (GELU_GRAD_BITS = 8: gelu' is confined to [-0.13, 1.13], a
        # 0.005-step code whose grid holds 0 and 1 exactly costs 0.24 % of its rms value and takes a quarter of the bytes out of fc1's store-bound epilogue
        # and of fc2-dX's); fp32 exact mode keeps the pre-activation itself
        g8 = adt == torch.bfloat16 and GELU_GRAD_BITS == 8 and Hd % 8 == 0
        h = torch.empty(M, Hd, dtype=torch.uint8 if g8 else adt, device=x.device)
        a = torch.empty(M, Hd, dtype=adt, device=x.device)
        ops.linear_fwd(y, M, D, w1, b1, a, Hd, epi=(EPI_GELU_GRAD8 if g8 else EPI_GELU_GRAD) if adt == torch.bfloat16 else EPI_GELU,
                       out2=h, ld_out2=Hd)
        out = torch.empty_like(x)
        ops.linear_fwd(a, M, Hd, w2, b2, out, D, epi=EPI_RESIDUAL, aux=x, ld_aux=D)
        ctx.save_for_backward(x, g, mean, rstd, y, h, a, w1, w2)
        ctx.small = (b, b1, b2)
        _chain_set(out, b2)
        return out

    @staticmethod
    @_scoped
    def backward(ctx, dout):
        x, g, mean, rstd, y, h, a, w1, w2 = ctx.saved_tensors
        B, T, D = x.shape
        M = B * T
        Hd = w1.shape[0]
        if ctx.x6:
            y6, a6 = y, a
            dout = _c(dout)
            b, b1, b2 = ctx.small
            d6, db2 = _take_side(dout, M, D)
            if d6 is None or d6.dtype != torch.bfloat16 or d6.shape[1] != ops.current_segments() * D:
                db2 = ops.grad_out(b2, (D,), x.device)
                d6 = ops.split_ex(dout.view(M, D), M, D, colsum_out=db2)
            dw2 = ops.tn_x6(d6, a6, M, w2)
            db1 = ops.grad_out(b1, (Hd,), x.device)
            if ops.nt_split_ok(M, Hd, D):
                dh6 = ops.nt_x6_dgelu_split(d6, w2, M, h, db1)   # one launch: the pieces of (dY W2) * gelu'(h) and its column sums
            else:
                dh = torch.empty(M, Hd, dtype=torch.float32, device=x.device)
                ops.nt_x6(d6, w2, "dx", M, dh)
                dh6 = ops.split_ex(dh, M, Hd, op=2, h=h, colsum_out=db1)     # (dY W2) * gelu'(h), its split and its column sums
            dw1 = ops.tn_x6(dh6, y6, M, w1)
            dy = torch.empty(M, D, dtype=torch.float32, device=x.device)
            ops.nt_x6(dh6, w1, "dx", M, dy)
            dx, dg, db = _ln_bwd_with_side(dy, x, D, g, b, mean, rstd, dout, M, torch.float32, ctx.up_bias)
            return dx, dg, db, dw1, db1, dw2, db2, None
        adt = y.dtype
        dout = _c(dout)
        d_act, db2 = _take_side(dout, M, D) if adt == torch.bfloat16 else (None, None)
        if d_act is None:
            d_act = ops.cast(dout, adt).view(M, D)
        b, b1, b2 = ctx.small
        dw2, db2b = ops.linear_dw(d_act, a, M, D, Hd, want_bias=db2 is None, weight=w2, bias=b2)
        db2 = db2 if db2 is not None else db2b
        dh = torch.empty(M, Hd, dtype=adt, device=x.device)
        if adt == torch.bfloat16:
            # (dY W2) * gelu'(h); the epilogue also leaves per-64-row column sums of dh = fc1's bias-gradient partials
            part = torch.empty((M + 63) // 64, Hd, dtype=torch.float32, device=x.device)
            ops.linear_dx(d_act, M, D, w2, dh, Hd, epi=EPI_MUL8 if h.dtype == torch.uint8 else EPI_MUL, aux=h, ld_aux=Hd,
                          colsum_partial=part)                                                 # h = gelu' here
            db1 = ops.colsum(part, part.shape[0], Hd, Hd, ops.grad_out(b1, (Hd,), x.device))
            dw1, _ = ops.linear_dw(dh, y, M, Hd, D, want_bias=False, weight=w1)
        else:
            ops.linear_dx(d_act, M, D, w2, dh, Hd, epi=EPI_DGELU, aux=h, ld_aux=Hd)
            dw1, db1 = ops.linear_dw(dh, y, M, Hd, D, weight=w1, bias=b1)
        dy = torch.empty(M, D, dtype=adt, device=x.device)
        ops.linear_dx(dh, M, Hd, w1, dy, D)
        dx, dg, db = _ln_bwd_with_side(dy, x, D, g, b, mean, rstd, dout, M, adt, ctx.up_bias)
        return dx, dg, db, dw1, db1, dw2, db2, None


def mlp_block(x, g, b, w1, b1, w2, b2, prec):
    return _MlpBlock.apply(x, g, b, w1, b1, w2, b2, prec)


# ------------------------------------------------------------------------------------------------------------
# decoders
# ------------------------------------------------------------------------------------------------------------
class _ClsHead(Function):
    """ClassificationDecoder: linear(norm(x[:, 0])) (vit.py:335-342) -> fp32 logits [B, num_classes]."""

    @staticmethod
    @_fwd
    def forward(ctx, x, g, b, w, bias, prec):
        adt = ops.act_dtype(prec)
        B, T, D = x.shape
        C = w.shape[0]
        x = _c(x)
        y, mean, rstd = ops.layernorm_fwd(x, T * D, B, D, g, b, adt)        # rows = the cls tokens, T*D apart
        logits = torch.empty(B, C, dtype=torch.float32, device=x.device)
        ops.linear_fwd(y, B, D, w, bias, logits, C)
        ctx.save_for_backward(x, g, mean, rstd, y, w)
        ctx.small = (b, bias)
        return logits

    @staticmethod
    @_scoped
    def backward(ctx, dlogits):
        x, g, mean, rstd, y, w = ctx.saved_tensors
        B, T, D = x.shape
        C = w.shape[0]
        adt = y.dtype
        ld = ops.pad8(C) if adt == torch.bfloat16 else C
        dl = torch.zeros(B, ld, dtype=adt, device=x.device)
        dl[:, :C] = dlogits                                                # tiny (B x C) pad+cast: cold glue
        b, bias = ctx.small
        dw, dbias = ops.linear_dw(dl, y, B, C, D, ld_dy=ld, weight=w, bias=bias)
        dy = torch.empty(B, D, dtype=adt, device=x.device)
        ops.linear_dx(dl, B, C, w, dy, D, ld_dy=ld)
        dx = torch.zeros_like(x)
        dg, db = ops.layernorm_bwd(dy, x, T * D, g, mean, rstd, None, dx, T * D, B, D, beta=b)
        return dx, dg, db, dw, dbias, None


def cls_head(x, g, b, w, bias, prec):
    return _ClsHead.apply(x, g, b, w, bias, prec)


def _seg_small_fwd(x, g, b, w, bias, prec):
    """x[:, 1:] -> LayerNorm -> Linear(D, C): the [B*h*w, C] fp32 class map before upsampling (vit.py:359-366)."""
    adt = ops.act_dtype(prec)
    B, T, D = x.shape
    C = w.shape[0]
    npatch = T - 1
    x = _c(x)
    xp = ops.gather_patch_rows(x, B, T, D, torch.float32)                # x[:, 1:] as dense rows
    M = B * npatch
    y, mean, rstd = ops.layernorm_fwd(xp, D, M, D, g, b, adt)
    small = torch.empty(M, C, dtype=torch.float32, device=x.device)      # [B, h*w, C]
    ops.linear_fwd(y, M, D, w, bias, small, C)
    return small, (xp, g, mean, rstd, y, w), (b, bias)


def _seg_small_bwd(saved, small_params, dims, dl, ld):
    """Backward of _seg_small_fwd from d(small) given as [M, ld] in the activation dtype (columns >= C zero)."""
    xp, g, mean, rstd, y, w = saved
    b, bias = small_params
    B, T, D, C = dims
    npatch = T - 1
    M = B * npatch
    adt = y.dtype
    dw, dbias = ops.linear_dw(dl, y, M, C, D, ld_dy=ld, weight=w, bias=bias)
    dy = torch.empty(M, D, dtype=adt, device=dl.device)
    ops.linear_dx(dl, M, C, w, dy, D, ld_dy=ld)
    dx = torch.zeros(B, T, D, dtype=torch.float32, device=dl.device)
    # rows of image b start at dx[b, 1]: the patch rows of one image are contiguous, images are T*D apart
    dxp = torch.empty(M, D, dtype=torch.float32, device=dl.device)
    dg, db = ops.layernorm_bwd(dy, xp, D, g, mean, rstd, None, dxp, D, M, D, beta=b)
    dx[:, 1:, :] = dxp.view(B, npatch, D)
    return dx, dg, db, dw, dbias


class _UpsampleBilinear(Function):
    """Rearrange('b (h w) c -> b c h w') + nn.Upsample(size, 'bilinear') (vit.py:355,367-371) on a decoder output
    [B, h*w, C] consumed in place -> fp32 [B, C, S, S].  Used when the decoder runs module by module (fake-quantised
    LayerNorm / Linear); the fused heads below call the same kernels."""

    @staticmethod
    @_fwd
    def forward(ctx, small, grid, size):
        ops.require_cuda(small)
        B, hw, C = small.shape
        small = _c(small.float())
        ctx.dims = (B, C, grid, size)
        return ops.upsample_bilinear_fwd(small, hw * C, 1, C, B, C, grid, grid, size, size)

    @staticmethod
    @_scoped
    def backward(ctx, dbig):
        B, C, grid, size = ctx.dims
        dsmall = torch.empty(B, grid * grid, C, dtype=torch.float32, device=dbig.device)
        ops.upsample_bilinear_bwd(_c(dbig.float()), dsmall, grid * grid * C, 1, C, B, C, grid, grid, size, size)
        return dsmall, None, None


def upsample_bilinear(small, grid, size):
    return _UpsampleBilinear.apply(small, grid, size)


class _SegHead(Function):
    """SegmentationDecoder: upsample(rearrange(linear(norm(x[:, 1:])))) (vit.py:359-374) -> fp32 [B, C, S, S]."""

    @staticmethod
    @_fwd
    def forward(ctx, x, g, b, w, bias, grid, size, prec):
        B, T, D = x.shape
        C = w.shape[0]
        npatch = T - 1
        small, saved, ctx.small = _seg_small_fwd(x, g, b, w, bias, prec)
        big = ops.upsample_bilinear_fwd(small, npatch * C, 1, C, B, C, grid, grid, size, size)
        ctx.save_for_backward(*saved)
        ctx.dims = (B, T, D, C, grid, size)
        return big

    @staticmethod
    @_scoped
    def backward(ctx, dbig):
        saved = ctx.saved_tensors
        B, T, D, C, grid, size = ctx.dims
        npatch = T - 1
        M = B * npatch
        adt = saved[4].dtype
        dbig = _c(dbig.float())
        dsmall = torch.empty(M, C, dtype=torch.float32, device=dbig.device)
        ops.upsample_bilinear_bwd(dbig, dsmall, npatch * C, 1, C, B, C, grid, grid, size, size)
        ld = ops.pad8(C) if adt == torch.bfloat16 else C
        if adt == torch.bfloat16:
            dl = torch.zeros(M, ld, dtype=adt, device=dbig.device)
            dl[:, :C] = dsmall                                                # (B*196 x 17) pad+cast: cold glue
        else:
            dl = dsmall
        return (*_seg_small_bwd(saved, ctx.small, (B, T, D, C), dl, ld), None, None, None)


class _SegHeadLoss(Function):
    """SegmentationDecoder + CrossEntropyLoss() + argmax in one fused tail (mv_seg_ce_*): the [B, C, S, S] logits are
    never written (vit.py:355-374 + segmentation/train.py:188,261-265).  -> (loss, pixel accuracy, pred uint8 [B,S,S])."""

    @staticmethod
    @_fwd
    def forward(ctx, x, g, b, w, bias, labels, grid, size, prec):
        B, T, D = x.shape
        C = w.shape[0]
        small, saved, ctx.small = _seg_small_fwd(x, g, b, w, bias, prec)
        stats, lse, pred, labels = ops.seg_ce_fwd(small, labels, B, C, grid, grid, size, size)
        adt = ops.act_dtype(prec)
        ctx.needs = any(ctx.needs_input_grad[:5])
        if ctx.needs:
            ld = ops.pad8(C) if adt == torch.bfloat16 else C
            # the gradient of the mean loss, produced now (labels and lse are hot), scaled by the incoming g in backward
            dl = ops.seg_ce_bwd(small, labels, lse, B, C, grid, grid, size, size, grad_dtype=adt, ld=ld, stats=stats)
            ctx.save_for_backward(dl, *saved)
            ctx.dims = (B, T, D, C, ld)
        ctx.mark_non_differentiable(pred)
        return stats[0], stats[1].detach(), pred

    @staticmethod
    @_scoped
    def backward(ctx, gloss, _gacc, _gpred):
        dl, *saved = ctx.saved_tensors
        B, T, D, C, ld = ctx.dims
        dl = dl * gloss.to(dl.dtype)                                          # [B*h*w, ld]: tiny
        return (*_seg_small_bwd(saved, ctx.small, (B, T, D, C), dl, ld), None, None, None, None)


def seg_head_loss(x, g, b, w, bias, labels, grid, size, prec):
    return _SegHeadLoss.apply(x, g, b, w, bias, labels, grid, size, prec)


def seg_head(x, g, b, w, bias, grid, size, prec):
    return _SegHead.apply(x, g, b, w, bias, grid, size, prec)


# ------------------------------------------------------------------------------------------------------------
# loss
# ------------------------------------------------------------------------------------------------------------
class _CrossEntropy(Function):
    """nn.CrossEntropyLoss() (mean) with the gradient produced in the same pass (classification/train.py:170,250)."""

    @staticmethod
    @_fwd
    def forward(ctx, logits, labels):
        ops.require_cuda(logits, labels)
        loss, dl, _ = ops.cross_entropy(logits.float(), labels, want_grad=True)
        ctx.save_for_backward(dl)
        ctx.shape = logits.shape
        return loss.view(())

    @staticmethod
    @_scoped
    def backward(ctx, g):
        (dl,) = ctx.saved_tensors
        return (dl.view(ctx.shape) * g), None


def cross_entropy(logits, labels):
    return _CrossEntropy.apply(logits, labels)


class CrossEntropyLoss(torch.nn.Module):
    """Drop-in for ``torch.nn.CrossEntropyLoss()`` (mean reduction, no weights) on the HIP kernel."""

    def forward(self, logits, labels):
        return cross_entropy(logits, labels)


# ------------------------------------------------------------------------------------------------------------
# fake quantisation with straight-through gradient (utils/quantize.py:77-89)
# ------------------------------------------------------------------------------------------------------------
class _FakeQuant(Function):
    @staticmethod
    @_fwd
    def forward(ctx, x, kind, a, b):
        dtype = x.dtype
        if kind == "float":
            y = ops.quant_float(x, a, b)
        elif kind == "fixed":
            y = ops.quant_fixed(x, a, b)
        else:
            raise ValueError(kind)
        return y.to(dtype).view(x.shape)

    @staticmethod
    @_scoped
    def backward(ctx, g):
        return g, None, None, None


def fake_quant_float(x, exp_bits, man_bits):
    return _FakeQuant.apply(x, "float", exp_bits, man_bits)


def fake_quant_fixed(x, wl, fl):
    return _FakeQuant.apply(x, "fixed", wl, fl)
