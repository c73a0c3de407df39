/* myrtle_vision HIP hot path -- C ABI (libmyrtle_vision_hip.so), gfx950 / MI355X only.
 *
 * The reference (MyrtleSoftware/myrtle-vision) has no FFI: its hot path is stock torch.nn
 * modules dispatching to ATen kernels.  Each entry point below replaces the ATen dispatch of
 * one reference call site (cited per function as <reference file>:<line>, relative to the
 * reference repo root, file src/myrtle_vision/models/vit.py unless another file is named).
 *
 * Conventions (all entry points):
 *   - plain pointers + sizes; no torch types; device pointers unless stated
 *   - return 0 on success, a negative MV_ERR_* otherwise; never throw, never synchronise,
 *     never allocate (callers pass workspaces); work is enqueued on `stream`
 *   - re-entrant, no thread-local state (autograd calls backward from a worker thread); the ONLY process-global
 *     state is the kernel-variant overrides of mv_gemm_force_variant / mv_gemm_f32_force_fma / mv_attention_bwd_force (tuning / test hooks, atomic,
 *     default 0 = automatic): they change which kernel computes a product, never what is computed
 *   - "rows x dim" tensors are row-major; `ld*` are leading dimensions in ELEMENTS
 *   - dtype codes: MV_F32 / MV_BF16
 *   - MFMA entry points (mv_gemm_*_bf16, mv_attention_*) need 16-byte aligned pointers and
 *     leading dimensions that are multiples of 8 elements
 */
#ifndef MYRTLE_VISION_HIP_H
#define MYRTLE_VISION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* mv_stream_t; /* hipStream_t */

enum { MV_F32 = 0, MV_BF16 = 1, MV_I8 = 2 /* mv_gemm_nt_i8 with MV_EPI_GELU_Q8 only */, MV_F16 = 3 /* IEEE half: mv_cast destination, mv_gemm_nt_bf16 output with MV_EPI_NONE */ };

enum {
  MV_OK = 0,
  MV_ERR_SHAPE = -1,       /* dimension constraint violated */
  MV_ERR_ALIGN = -2,       /* pointer / leading-dimension alignment */
  MV_ERR_LAUNCH = -3,      /* hipGetLastError() after launch */
  MV_ERR_UNSUPPORTED = -4, /* dtype / epilogue combination not built */
  MV_ERR_WORKSPACE = -5    /* workspace too small */
};

/* epilogues of the GEMM entry points */
enum {
  MV_EPI_NONE = 0,     /* C = acc (+ bias) */
  MV_EPI_GELU = 1,     /* out2 = acc + bias (pre-activation, optional); C = gelu_erf(acc + bias) */
  MV_EPI_RESIDUAL = 2, /* C = acc + bias + aux              (aux: fp32 [M, ld_aux]) */
  MV_EPI_DGELU = 3,    /* C = acc * gelu_erf'(aux)          (aux: pre-activation, dtype of A); bf16 kernel only: out2 (optional)
                          = fp32 [ceil(M/64), ld_out2] per-64-row column sums of C (bias-gradient partials) */
  MV_EPI_GELU_GRAD = 5, /* bf16 kernel only: C = gelu_erf(acc + bias), out2 (optional) = gelu_erf'(acc + bias): what the
                           backward needs, so that its epilogue is MV_EPI_MUL instead of re-evaluating erf */
  MV_EPI_MUL = 6,      /* bf16 kernel only: C = acc * aux (aux: dtype of A, e.g. the gelu' left by MV_EPI_GELU_GRAD);
                          out2 as MV_EPI_DGELU (per-64-row column sums of C) */
  MV_EPI_GELU_Q8 = 7,  /* mv_gemm_nt_i8 only: C (int8) = quint8 code - 128 of gelu_erf(acc + bias) under the NEXT layer's
                          quantiser (q_scale, q_zero_point): FeedForward's Linear -> GELU -> next Linear's QuantStub
                          (vit.py:48-51) without the fp32 hidden activations ever reaching memory */
  MV_EPI_GELU_GRAD8 = 8, /* as MV_EPI_GELU_GRAD with out2 as ONE BYTE per element (uint8 [M, ld_out2]): gelu' lies in
                            [-0.129, 1.129]; code = round(gelu' * 200) + 26: 0 and 1 are codes 26 and 226 exactly, |error| <= 0.0025 */
  MV_EPI_MUL8 = 9,       /* as MV_EPI_MUL with aux = those codes (uint8 [M, ld_aux]): C = acc * ((code - 26) * 0.005) */
  MV_EPI_SPLIT_DGELU = 10, /* mv_gemm_nt_bf16 with C bf16 [M, aux_i * N] (aux_i = 3 | 6): the bf16 PIECES (mv_split2_bf16 / mv_split3_bf16,
                              role 0, segments N apart) of acc * gelu_erf'(aux), aux fp32 [M, ld_aux]; out2 as MV_EPI_DGELU.  The
                              fc2 input gradient of the split-operand modes leaves as the operand of fc1's dW / dX products: no fp32
                              tensor, no split pass.  Whole 256 x 256 tiles only (M % 256 == N % 256 == 0) */
  MV_EPI_SPLIT_GELU = 11,  /* likewise: out2 = acc + bias (fp32 pre-activation [M, ld_out2], required), C = the pieces of
                              gelu_erf(acc + bias): fc1 of the split-operand modes */
  MV_EPI_EMBED = 4     /* patch-embedding: row m of the GEMM is patch (m % aux_i) of image (m / aux_i);
                          C row = img*(aux_i+1) + 1 + patch;  C = acc + bias + aux[1 + patch]  (aux: fp32 [aux_i+1, N]) */
};

int mv_version(void);
const char* mv_error_string(int code);
/* number of bytes of workspace mv_gemm_tn_bf16 / mv_layernorm_bwd want for these sizes */
size_t mv_gemm_tn_workspace_bytes(int M, int N, int Kc);
/* Tuning / test hook (no reference counterpart): force a kernel variant for the following mv_gemm_nt_bf16 (0 auto | 128 | 2564 ring |
 * 2568 8-phase | 25680 8-phase without the half-item tail | 3000 / 3002 automatic dispatch with the 8-phase kernel's column bands
 * off / on | 3100 / 3102 the same with the 8-phase kernel forced) and mv_gemm_tn_bf16 (0 auto | 128 | 256 ring) calls of this
 * process; a forced variant that cannot run a shape (alignment of K) falls back to the automatic choice.  Results are identical up
 * to fp32 summation order (bands on / off: bit for bit). */
int mv_gemm_force_variant(int nt_variant, int tn_variant);
size_t mv_layernorm_bwd_workspace_bytes(int rows, int dim);

/* ---- LayerNorm: nn.LayerNorm(dim), eps 1e-5 -- vit.py:37,41 (PreNorm), :332,:353 (decoder norms) ----
 * x: fp32 [rows, dim] with row stride ldx (lets the decoder read x[:,0] / x[:,1:] in place);
 * y: y_dtype [rows, dim] dense; mean/rstd: fp32 [rows] (saved for backward). */
int mv_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, void* y, int y_dtype,
                     float* mean, float* rstd, int rows, int dim, float eps, mv_stream_t stream);
/* The same LayerNorm with the output leaving as the bf16 pieces of the split-operand Linear products (nseg = 6: mv_split3_bf16's
 * role-0 side-by-side layout [rows, 6 * dim], nseg = 3: mv_split2_bf16's [rows, 3 * dim]); bit-identical to mv_layernorm_fwd (fp32)
 * followed by the split, without the fp32 tensor in between.  dim <= 1024. */
int mv_layernorm_fwd_split(const float* x, long ldx, const float* gamma, const float* beta, void* y_split, int nseg, float* mean,
                           float* rstd, int rows, int dim, float eps, mv_stream_t stream);
/* dx[rows, dim] (fp32, row stride lddx) = dLN(dy) (+ dx_add if non-null, same layout as dx; may alias dx);
 * dgamma/dbeta: fp32 [dim], overwritten (accumulate=0) or added to (accumulate=1).
 * Optional fused by-products for the consumer of dx in the backward chain (both may be NULL):
 *   dx_bf16  : bf16 [rows, dim] dense copy of dx (the MFMA operand of the next block's dX/dW products);
 *   dx_colsum: fp32 [dim] = sum over rows of dx (that block's output-projection bias gradient), overwritten. */
int mv_layernorm_bwd(const void* dy, int dy_dtype, const float* x, long ldx, const float* gamma,
                     const float* mean, const float* rstd, const float* dx_add, float* dx, long lddx,
                     float* dgamma, float* dbeta, int accumulate, float* workspace, size_t workspace_bytes,
                     int rows, int dim, void* dx_bf16, float* dx_colsum, mv_stream_t stream);
/* The same backward with dx ALSO leaving as the bf16 pieces of the split-operand products (dx_split [rows, nseg * dim], nseg = 3 | 6,
 * mv_split2_bf16 / mv_split3_bf16 role 0): what the consumer of dx in backward order feeds its dW / dX products, without a split pass. */
int mv_layernorm_bwd_split(const void* dy, int dy_dtype, const float* x, long ldx, const float* gamma, const float* mean,
                           const float* rstd, const float* dx_add, float* dx, long lddx, float* dgamma, float* dbeta, int accumulate,
                           float* workspace, size_t workspace_bytes, int rows, int dim, void* dx_split, int nseg, float* dx_colsum,
                           mv_stream_t stream);

/* ---- dense contractions on MFMA (bf16 in, fp32 accumulate) ----
 * nn.Linear forward  y = x W^T + b : patch_to_embedding :278, to_qkv :86, to_out :98, net.0/net.3 :48-51,
 * decoder.linear :333/:354.   C[M,N] = A[M,K] . B[N,K]^T  (both operands K-contiguous).
 * The same entry point computes the input gradient dX = dY . W with B = W^T[K_in, N_out] (a transposed
 * bf16 copy the host keeps), i.e. autograd's mm for AddmmBackward.
 * Requirements: K % 8 == 0 is NOT required, but rows must be readable (and zero) up to round_up(K, 8). */
int mv_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_dtype, int M, int N,
                    int K, const float* bias, int epilogue, const void* aux, int ld_aux, int aux_i, void* out2,
                    int ld_out2, mv_stream_t stream);
/* the same with C = alpha * (A . B^T) (+ bias, epilogue): the integer-code products of the converted int8 path
 * (mv_quant_affine_codes), where alpha = scale_activation * scale_weight.  Serves QFormat.PyTorchINT8 after convert()
 * (utils/quantize.py:230-251, 329-338: torch.quantization.convert -> quantized::linear, which does not run in the
 * reference; classification/test_quantize.py:26-34,145-156 is the caller). */
int mv_gemm_nt_bf16_scaled(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_dtype, int M, int N,
                           int K, float alpha, const float* bias, int epilogue, const void* aux, int ld_aux, int aux_i,
                           void* out2, int ld_out2, mv_stream_t stream);
/* weight gradient  dW[M,N] (fp32) (+)= A[Kc,M]^T . B[Kc,N]  (A = dY, B = X; contraction over tokens) --
 * autograd's mm(dY^T, X) for every nn.Linear above.  Split over Kc into fp32 slabs in `workspace`,
 * reduced deterministically.  colsum (optional, fp32 [M]) (+)= column sums of A = bias gradient. */
int mv_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int Kc,
                    int accumulate, float* colsum, float* workspace, size_t workspace_bytes, mv_stream_t stream);

/* ---- generic fp32 contraction (exact-parity mode, odd shapes, materialised attention) ----
 * C[b1,b2][m,n] = sum_k A[b1,b2][m,k] * B[b1,b2][k,n] with arbitrary element strides; same epilogues
 * (aux/out2 are fp32).  alpha scales the accumulator before the epilogue; accumulate adds into C. */
int mv_gemm_f32(const float* A, long sa_m, long sa_k, long sa_b1, long sa_b2, const float* B, long sb_k, long sb_n,
                long sb_b1, long sb_b2, float* C, long sc_m, long sc_n, long sc_b1, long sc_b2, int M, int N, int K,
                int nb1, int nb2, float alpha, int accumulate, const float* bias, int epilogue, const float* aux,
                long ld_aux, int aux_i, float* out2, long ld_out2, mv_stream_t stream);
/* out[i] (+)= sum_s slabs[s * stride + i], s in fixed order: the deterministic reduce of a product whose contraction was
 * split over mv_gemm_f32's batch dimension (dW = dY^T X has 50 432 contraction rows and only 36-144 output tiles) */
int mv_sum_slabs(const float* slabs, long stride, int S, float* out, long n, int accumulate, mv_stream_t stream);
/* out[i] = (add ? add[i] : 0) + sum over s of slabs[s * stride + i]; add may alias out */
int mv_sum_slabs_add(const float* slabs, long stride, int S, const float* add, float* out, long n, mv_stream_t stream);
/* Test / tuning hook (no reference counterpart; the second piece of process-global state next to mv_gemm_force_variant):
 * 1 = mv_gemm_f32 runs its FMA kernel, 0 (default; MV_GEMM_F32=fma in the environment starts with 1) = the f32-input MFMA
 * kernels, 2 = matrix cores but only the generic (any stride, any size) kernel.  All compute the same k-ordered fmaf chain
 * per output: results are bit-identical. */
int mv_gemm_f32_force_fma(int on);

/* ---- fused multi-head self-attention core -- Attention.forward vit.py:87-96 ----
 * qkv: bf16 [B, N, 3, H, 64] (the to_qkv output, feature order [q|k|v][head][dh], vit.py:87-90);
 * out: bf16 [B, N, H*64] = (softmax(q k^T * scale) v).transpose(1,2).reshape(B,N,C);  lse: fp32 [B,H,N].
 * dim_head must be 64 (all shipped configs; ViT default dim_head=64, vit.py:178); N <= 320. */
int mv_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, float scale, mv_stream_t stream);
/* dqkv: bf16 [B, N, 3, H, 64]; dout/out: bf16 [B, N, H*64].  colsum (optional): fp32 [B, 3*H*64], row b = column sums
 * of image b's dqkv rows (fp32 accumulators, before the bf16 rounding) -- summed over b they are to_qkv's bias gradient,
 * so no separate pass over dqkv is needed for it. */
int mv_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv, float* colsum,
                     int B, int N, int H, float scale, mv_stream_t stream);
/* The attention core of precision "bf16x3" (vit.py:87-96 between fp32 tensors): the fused kernels above on IEEE-half operands with
 * fp32 accumulation, softmax and OUTPUTS; N <= 288 (N <= 208: the 13-key-tile kernels; above: the two-pass kernels of the 257-token case).  qkv16: half [B, N, 3, H, 64] (mv_cast to MV_F16 of the fp32 to_qkv output);
 * out / lse as mv_attention_fwd but out is fp32.  Backward: mv_attention_bwd_prep_f16 turns the fp32 dout [B, N, H*64] into half
 * scaled, per (image, head), by a power of two s (the slice's largest magnitude -> [2^7, 2^8): gradients lie below half's normal
 * range otherwise), leaves s in gscale[b * H + h] (device, fp32 [B * H]) and delta[b, h, n] = sum_d half(dout * s) * out.
 * mv_attention_bwd_f16 then writes dqkv with s divided out (exactly): nseg = 0: fp32 [B, N, 3, H, 64]; nseg = 3 / 6: the bf16 pieces of
 * the split-operand products ([B * N, nseg * 3 * H * 64], mv_split2_bf16 / mv_split3_bf16 role 0) -- the dY operand of to_qkv's dW
 * and dX products without an fp32 tensor and a split pass; colsum (optional) as mv_attention_bwd. */
int mv_attention_fwd_f16(const void* qkv16, float* out, float* lse, int B, int N, int H, float scale, mv_stream_t stream);
int mv_attention_bwd_prep_f16(const float* dout, const float* out, void* dout16, float* delta, float* gscale, int B, int N, int H,
                              mv_stream_t stream);
int mv_attention_bwd_f16(const void* qkv16, const void* dout16, const float* delta, const float* lse, const float* gscale,
                         void* dqkv, int nseg, float* colsum, int B, int N, int H, float scale, mv_stream_t stream);
/* Test / tuning hook (process-global, atomic, like mv_gemm_force_variant): backward kernel for the following
 * mv_attention_bwd calls -- 0 auto (N <= 208: 4; N <= 288: 2; else 8), 4 = dS exchanged through LDS (N <= 208), 5 = the same
 * with two waves of 512 registers per workgroup (192 < N <= 208; equal results up to the placement of the softmax scale), 2 = two
 * barrier-free passes (N <= 288), 8 = the eight-wave kernel (N <= 320).  A variant that cannot take the length falls back
 * to auto. */
int mv_attention_bwd_force(int variant);
/* The same for mv_attention_fwd: 0 auto (N <= 208: 3, else 1), 3 = 13 key tiles in 53 KB of LDS, three workgroups per CU (N <= 208),
 * 1 = one 16-query tile per wave and pass, 2 = a PAIR of query
 * tiles per wave and pass (every K / V^T fragment read from LDS feeds two MFMAs; N <= 224, else falls back to 1). */
int mv_attention_fwd_force(int variant);
/* The same attention core in EXACT fp32 arithmetic on the f32-input matrix cores, forward only: qkv fp32 [B, N, 3, H, 64],
 * out fp32 [B, N, H*64]; N <= 272.  For the paths that need fp32 values and no gradient (converted PyTorchINT8 model,
 * precision="fp32" evaluation): same products and sums as mv_gemm_f32 + mv_softmax_fwd + mv_gemm_f32 up to summation order,
 * without the [B, H, N, N] probabilities. */
int mv_attention_fwd_f32(const float* qkv, float* out, int B, int N, int H, float scale, mv_stream_t stream);

/* ---- row softmax for the materialised attention path (fp32): attn.softmax(dim=-1) vit.py:93 ---- */
int mv_softmax_fwd(const float* x, float* y, long rows, int cols, float scale, mv_stream_t stream);
/* dx = scale * y * (dy - sum(dy*y)) */
int mv_softmax_bwd(const float* y, const float* dy, float* dx, long rows, int cols, float scale, mv_stream_t stream);

/* ---- patchify: vit.py:271-275  (B,C,H,W) fp32 -> (B*gh*gw, p*p*C) out_dtype, k = (py*p+px)*C + c ---- */
int mv_patchify(const float* img, void* out, int out_dtype, int B, int C, int H, int W, int p, mv_stream_t stream);
/* cls row of the embedding: x[b, 0, :] = cls[:] + pos[0, :]  (vit.py:283-290,305-310); x fp32 [B, T, D] */
int mv_embed_cls(const float* cls, const float* pos, float* x, int B, int T, int D, mv_stream_t stream);
/* backward of the embedding assembly: dpos[t, :] (+)= sum_b dx[b, t, :];  dcls[:] (+)= sum_b dx[b, 0, :] */
int mv_embed_bwd(const float* dx, float* dpos, float* dcls, int accumulate, int B, int T, int D, mv_stream_t stream);
/* mv_embed_bwd and mv_gather_patch_rows in one pass over dx (vit.py:271-311 backward): dpos / dcls as mv_embed_bwd (either may be NULL; written, not
 * accumulated) and dy = rows 1..T-1 of every image in dy_dtype (NULL: skipped) -- the dY operand of the patch GEMM's dW product.
 * D % 4 == 0, pointers 16-byte aligned. */
int mv_embed_bwd_gather(const float* dx, void* dy, int dy_dtype, float* dpos, float* dcls, int B, int T, int D,
                        mv_stream_t stream);
/* gather rows 1..T-1 of every image: dst[b*(T-1)+t-1, :] = (dtype) src[b, t, :]  (dY for the patch GEMM) */
int mv_gather_patch_rows(const float* src, void* dst, int dst_dtype, int B, int T, int D, mv_stream_t stream);

/* ---- casts / layout ---- */
/* dst (dst_dtype) = src (src_dtype), n elements */
int mv_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, mv_stream_t stream);
/* fp32 -> three bf16 pieces for the "bf16x6" fp32 product (the fp32 arithmetic mode of nn.Linear, vit.py:48-51,72-74, on
 * the bf16 matrix cores): x = p0 + p1 + p2 to 2^-26 |x|; writes the six segments of one operand, ``seg`` elements apart,
 * in the order role 0 (left operand): p0 p0 p1 p0 p1 p2, role 1 (right operand): p0 p1 p0 p2 p1 p0, so that a bf16 product
 * contracting over all six segments equals the fp32 product to 2^-25 relative.  out row stride ldo; seg = cols with
 * ldo = 6 * cols lays the segments side by side along the contraction axis of an NT product, seg = rows * ldo stacks
 * them along the rows (TN product).  cols, ldx, ldo, seg multiples of 4; x and out 16-byte aligned. */
int mv_split3_bf16(const float* x, long ldx, void* out, long ldo, long seg, long rows, int cols, int role,
                   mv_stream_t stream);
/* The role-0 side-by-side split (out [rows, 6 * cols] bf16) of v, behind a producer's last elementwise step: op 0: v = x;
 * op 1: v = gelu(x) (nn.GELU, vit.py:49: fc1's activation as the fc2 operand, the fp32 activation never stored); op 2:
 * v = x * gelu'(h) (fc2's dX times the activation derivative).  colsum (optional, fp32 [cols]) receives the column sums of v
 * = the bias gradient of the Linear that v is the output gradient of (deterministic two-stage sum; workspace of
 * mv_split3_ex_workspace_bytes(rows, cols) bytes, needed only with colsum). */
size_t mv_split3_ex_workspace_bytes(long rows, int cols);
int mv_split3_bf16_ex(const float* x, long ldx, const float* h, long ldh, int op, void* out, long rows, int cols,
                      float* colsum, float* workspace, size_t workspace_bytes, mv_stream_t stream);
/* K-split form of mv_gemm_nt_bf16 for products with few output tiles and a long contraction (the bf16x6 products with a
 * 768-wide output at batch 64: 150 tiles of 256x256 on 256 CUs): slabs[s] (fp32 [M, N], dense, s < splits) = A[:, slice s]
 * B[:, slice s]^T, the bias added to slab 0; sum with mv_sum_slabs_add.  K % (128 * splits) == 0, N % 4 == 0. */
int mv_gemm_nt_bf16_ksplit(const void* A, int lda, const void* B, int ldb, float* slabs, int M, int N, int K, int splits,
                           const float* bias, mv_stream_t stream);
/* dW of an fp32 nn.Linear as a bf16x6 product: A6 = the role-0 side-by-side split (seg = cols, ldo = 6 * cols) of dY
 * [rows, M], B6 = that of X [rows, N]; C[M, N] (fp32, row stride ldc) = dY^T X to fp32 accuracy.  The kernel addresses the
 * six piece pairings inside the two buffers itself, so the splits made for the dX / forward products are reused as they
 * are.  M, N multiples of 8, rows a multiple of 32; workspace as mv_gemm_tn_workspace_bytes(M, N, 6 * rows). */
int mv_gemm_tn_bf16_x6(const void* A6, const void* B6, float* C, int ldc, int M, int N, int rows, float* workspace,
                       size_t workspace_bytes, mv_stream_t stream);
/* "bf16x3": the fp32 products of nn.Linear (vit.py:48-51,72-74,86) to 2^-16 relative -- inside BASELINE's 1e-3 end to end -- at
 * half the matrix-core work of bf16x6 (``precision="bf16x3"``).  Each operand as TWO bf16 pieces p0 = bf16(x), p1 = bf16(x - p0),
 * three segments in the order role 0: p0 p0 p1, role 1: p0 p1 p0 (the first three of mv_split3_bf16's six), so a bf16 product
 * over the 3 * K contraction is a0 b0 + a0 b1 + a1 b0.  Same argument meaning as mv_split3_bf16 / mv_split3_bf16_ex (the _ex
 * form writes [rows, 3 * cols]; workspace as mv_split3_ex_workspace_bytes) / mv_gemm_tn_bf16_x6 (A3 [rows, 3 M], B3 [rows, 3 N],
 * workspace as mv_gemm_tn_workspace_bytes(M, N, 3 * rows)). */
int mv_split2_bf16(const float* x, long ldx, void* out, long ldo, long seg, long rows, int cols, int role,
                   mv_stream_t stream);
int mv_split2_bf16_ex(const float* x, long ldx, const float* h, long ldh, int op, void* out, long rows, int cols,
                      float* colsum, float* workspace, size_t workspace_bytes, mv_stream_t stream);
/* The right-operand pieces of an nn.Linear weight w [R = out, C = in] (fp32, dense) for BOTH of its split-operand products from
 * one read: fwd [R, nseg * C] = mv_split2/3_bf16(w, role 1) (y = x W^T, vit.py:48-51,86,98) and dx [C, nseg * R] = the same of w^T
 * (dx = dy W); either may be NULL.  nseg = 3 | 6; R and C even.  Replaces a transposed fp32 copy + two split passes per weight and
 * optimizer step. */
int mv_weight_split(const float* w, void* fwd, void* dx, int R, int C, int nseg, mv_stream_t stream);
/* Round 4, NOT yet used by the model (a measured building block for a faster tolerance-meeting mode,
 * profiles/r04_fp8_correction_study.txt): the bf16x3 product a0 b0 + a0 b1 + a1 b0 of an nn.Linear (vit.py:48-51,86,98) with its two
 * CORRECTION terms on e4m3 operands and the 8-bit matrix instruction.  mv_split_f8c writes an operand's rows of 4 * cols bytes
 * [p0 as bf16 | 8-bit segment 1 | 8-bit segment 2] (role 0: Q(p0 2^e) | Q(p1 2^(e+8)); role 1: Q(p1 2^(e+8)) | Q(p0 2^e)); exp_hi = e:
 * max |x| 2^e < 448.  mv_gemm_nt_f8c: C[M, N] = A0 B0^T + 2^scale_exp (segments 1 and 2 contracted), scale_exp = -(e_a + e_b + 8);
 * K % 128 == 0; epilogues MV_EPI_NONE (fp32 / bf16 C) and MV_EPI_RESIDUAL (fp32 C). */
int mv_split_f8c(const float* x, long ldx, void* out, long ldo_bytes, long rows, int cols, int role, int exp_hi, mv_stream_t stream);
int mv_gemm_nt_f8c(const void* A, long lda_bytes, const void* B, long ldb_bytes, void* C, int ldc, int c_dtype, int M, int N, int K,
                   int scale_exp, const float* bias, int epilogue, const void* aux, int ld_aux, mv_stream_t stream);
int mv_gemm_tn_bf16_x3(const void* A3, const void* B3, float* C, int ldc, int M, int N, int rows, float* workspace,
                       size_t workspace_bytes, mv_stream_t stream);
/* weight prep for the MFMA path: w fp32 [R, C] -> w_bf16 [R, ldw] and wt_bf16 [C, ldt] (transposed), pads zeroed;
 * either output may be NULL */
int mv_weight_prep(const float* w, void* w_bf16, int ldw, void* wt_bf16, int ldt, int R, int C, mv_stream_t stream);
/* the same for many weights in one launch (after an optimizer step every nn.Linear weight of the model is stale:
 * vit.py:72-74,86,48-51 x depth).  ``items_device``: DEVICE array of ``count`` items sorted by first_block; item i covers
 * blocks [first_block, first_block + tiles_x * tiles_y) with tiles_x = ceil(max(C, ldw) / 64), tiles_y = ceil(max(R, ldt)
 * / 64); first_block of item 0 is 0 and total_blocks is the sum.  Both outputs of every item are required; ldw and
 * ldt even, w 8-byte and the outputs 4-byte aligned. */
typedef struct mv_weight_prep_item {
  const float* w;
  void* w_bf16;
  void* wt_bf16;
  int ldw, ldt, R, C;
  int tiles_x, first_block;
} mv_weight_prep_item;
int mv_weight_prep_batch(const mv_weight_prep_item* items_device, int count, int total_blocks, mv_stream_t stream);
/* column sums (bias gradient): out[c] (+)= sum_r x[r, c]; x dtype x_dtype [rows, ld] */
int mv_colsum(const void* x, int x_dtype, long ld, float* out, int accumulate, long rows, int cols, float* workspace,
              size_t workspace_bytes, mv_stream_t stream);
/* y = gelu(x) and dx = dy * gelu'(x) (unfused forms, fp32/bf16) */
int mv_gelu_fwd(const void* x, void* y, int dtype, long n, mv_stream_t stream);
int mv_gelu_bwd(const void* x, const void* dy, void* dx, int dtype, long n, mv_stream_t stream);
/* out = a + b (fp32), n elements: Residual.res_add vit.py:27 in the unfused path */
int mv_add_f32(const float* a, const float* b, float* out, long n, mv_stream_t stream);

/* ---- fake quantisation (QPyTorch path): utils/quantize.py:46-72,84 ----
 * fp32 in/out, nearest rounding; float format (exp, man) or fixed point (wl, fl); in place allowed */
int mv_quant_float(const float* x, float* y, long n, int exp_bits, int man_bits, mv_stream_t stream);
int mv_quant_fixed(const float* x, float* y, long n, int wl, int fl, int clamp, int symmetric, mv_stream_t stream);
/* per-tensor affine fake-quant (MinMaxObserver qparams, utils/quantize.py:242-249) */
int mv_quant_affine(const float* x, float* y, long n, float scale, int zero_point, int qmin, int qmax,
                    mv_stream_t stream);
/* (utils/quantize.py:242-249: MinMaxObserver quint8 affine activations / qint8 symmetric weights)
 * integer codes of the affine quantiser, re-centred: codes[r, c] = clamp(rint(x / scale) + zp, qmin, qmax) - zp as bf16
 * (exact: |code| <= 256), rows ld elements apart with zeroed padding -- an MFMA operand whose products with another
 * code tensor, accumulated in fp32, are the exact integer dot products of real int8 inference.
 * x_dtype: MV_F32 or MV_BF16 (the fused attention kernel's output).  pre_op 1: x is passed through GELU (erf form) first -- FeedForward's nn.GELU between its two quantised Linears
 * (vit.py:48-51) without the fp32 round trip; 0: none. */
int mv_quant_affine_codes(const void* x, int x_dtype, void* codes, long rows, int cols, int ld, float scale, int zero_point,
                          int qmin, int qmax, int pre_op, mv_stream_t stream);
/* float_quantize(exp 5, man 10) written as IEEE half (exact: every value of the format is a half), and the NT product of two
 * such operands on v_mfma_f32_16x16x32_f16 with fp32 accumulation: the FORWARD products of the FP16_16 / FP16_32 formats
 * (quantize.py:253-327: both the activation stub and weight_fake_quant round to (5, 10)), which the reference computes as an
 * fp32 GEMM of the same values.  A [M, lda], B [N, ldb] half, K % 128 == 0; C fp32; epilogues MV_EPI_NONE / MV_EPI_RESIDUAL. */
int mv_quant_float_f16(const float* x, void* y, long n, mv_stream_t stream);
int mv_gemm_nt_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const float* bias,
                   int epilogue, const void* aux, int ld_aux, void* out2, int ld_out2, mv_stream_t stream);
/* the same quantiser writing int8 codes q - 128 (quint8: qmin 0, qmax 255) for mv_gemm_nt_i8; ld = row stride in BYTES
 * (multiple of 16), columns [cols, ld) zero-filled */
int mv_quant_affine_i8(const void* x, int x_dtype, void* codes, long rows, int cols, int ld, float scale, int zero_point,
                       int pre_op, mv_stream_t stream);
/* int8 x int8 -> int32 NT product on v_mfma_i32_16x16x64_i8 with the affine bookkeeping in the epilogue (config 5; build-
 * defined: the reference's converted PyTorchINT8 path, quantize.py:230-251 + classification/test_quantize.py:109, does not run):
 *   C[m][n] = alpha * ( sum_k A8[m][k] * B8[n][k] + icorr[n] ) + bias[n]  (+ aux[m][n] for MV_EPI_RESIDUAL)
 * With A8 = q_x - 128, B8 = q_w and icorr[n] = (128 - zero_point_x) * sum_k q_w[n][k] the bracket is the exact integer dot
 * product of (q_x - zero_point_x) and q_w; alpha = scale_x * scale_w.  A8 [M, lda], B8 [N, ldb] int8, strides in bytes
 * (multiples of 16), K % 256 == 0; C fp32 or bf16; epilogues MV_EPI_NONE / MV_EPI_RESIDUAL (fp32) / MV_EPI_GELU_Q8 (C int8
 * [M, ldc bytes], c_dtype MV_I8, M % 256 == N % 256 == 0; q_scale / q_zero_point: the next layer's quantiser). */
int mv_gemm_nt_i8(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_dtype, int M, int N, int K,
                  float alpha, const float* bias, const int* icorr, int epilogue, const void* aux, int ld_aux,
                  float q_scale, int q_zero_point, mv_stream_t stream);
/* The same fp32 core for TRAINING in precision="fp32" (vit.py:92-96 forward and its autograd backward), still without the
 * [B, H, N, N] tensors: the forward also leaves lse[b, h, n] = log sum_j exp(scale q_n . k_j) (fp32 [B, H, N]); the backward
 * recomputes the probabilities from it and returns dqkv (fp32, qkv's layout) from qkv, out (the forward's result), dout.
 * N <= 272, dim_head 64. */
int mv_attention_fwd_f32_lse(const float* qkv, float* out, float* lse, int B, int N, int H, float scale, mv_stream_t stream);
int mv_attention_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int B, int N,
                         int H, float scale, mv_stream_t stream);
/* Producers with the NEXT layer's quint8 quantiser fused in (converted PyTorchINT8 model; bit-identical to producer +
 * mv_quant_affine_i8): LayerNorm (vit.py:37,41) and the exact-fp32 attention core (vit.py:92-97, quant_out) writing int8
 * codes q - 128 [rows, dim] / [B, N, H*64] directly.  With MV_EPI_GELU_Q8 above they remove every fp32 activation tensor
 * between an int8 model's Linear layers except the q/k/v projections and the residual stream. */
int mv_layernorm_fwd_q8(const float* x, long ldx, const float* gamma, const float* beta, void* codes, int rows, int dim,
                        float eps, float scale, int zero_point, mv_stream_t stream);
int mv_attention_fwd_f32_q8(const float* qkv, void* codes, int B, int N, int H, float scale, float q_scale, int q_zero_point,
                            mv_stream_t stream);
/* running min/max observer: minmax[0] = min(minmax[0], min x), minmax[1] = max(minmax[1], max x);
 * minmax points at FOUR floats: [2..3] are scratch for the reduction */
int mv_minmax(const float* x, long n, float* minmax, mv_stream_t stream);

/* ---- loss: nn.CrossEntropyLoss() mean reduction -- classification/train.py:170,250; segmentation/train.py:188,261 ----
 * logits fp32 viewed as [outer, C, inner] (classification: inner = 1; segmentation: outer = B, inner = H*W);
 * labels int64 [outer*inner]: torch semantics -- ignore_index -100 gives no loss / gradient and is left out of the mean;
 *   any other label outside [0, C) is never dereferenced and turns the loss into NaN (torch device-asserts there);
 * loss_sum: fp32 [4], zeroed by this call: [0] = sum of per-sample losses / count (NaN when count = 0, as torch),
 *   [1] = count of non-ignored labels, [2] = number of out-of-range labels, [3] unused;
 * dlogits (optional, dl_dtype, same layout, row length ld_dl >= C when inner == 1 with zeroed padding)
 *   = (softmax - onehot) * grad_scale / count, zero rows for ignored labels.  argmax (optional int64 [outer*inner]). */
int mv_cross_entropy(const float* logits, const int64_t* labels, float* loss_sum, void* dlogits, int dl_dtype,
                     int ld_dl, int64_t* argmax, long outer, int C, long inner, float grad_scale,
                     mv_stream_t stream);

/* ---- bilinear upsample (align_corners=False): nn.Upsample(size, 'bilinear') vit.py:355,371 ----
 * small[b, c, y, x] is read at  small + b*sb + c*sc + (y*w + x)*sp  (so the [B, h*w, C] output of the decoder GEMM is
 * consumed in place: the reference's transpose/view, vit.py:367-369, costs nothing); big: fp32 [B, C, H, W] dense.
 * backward is the exact adjoint in gather form (deterministic). */
int mv_upsample_bilinear_fwd(const float* small, long sb, long sc, long sp, float* big, int B, int C, int h, int w,
                             int H, int W, mv_stream_t stream);
int mv_upsample_bilinear_bwd(const float* dbig, float* dsmall, long sb, long sc, long sp, int B, int C, int h, int w,
                             int H, int W, mv_stream_t stream);

/* ---- fused segmentation tail: CrossEntropyLoss()(Upsample(size=(H,W),'bilinear')(small), labels) ----
 * replaces vit.py:355,371 (SegmentationDecoder.upsample) + segmentation/train.py:188,261-265 (criterion, argmax, accuracy)
 * in one pass that never materialises the [B, C, H, W] logits.
 * small: fp32 [B, h*w, C] (the decoder GEMM output, consumed in place); labels: int64 [B, H, W] in [0, C), or -100
 *      (ignore_index: no loss, no gradient, not counted in the mean); other out-of-range labels make the loss NaN.
 * fwd: lse fp32 [B,H,W] (log-sum-exp per pixel, kept for the backward), pred uint8 [B,H,W] (first-index arg-max),
 *      partials fp32 [4 * mv_seg_ce_partials(B,H,W)] scratch, stats fp32 [4]: [0] = mean loss over the counted labels,
 *      [1] = pixel accuracy over all pixels, [2] = counted labels, [3] = out-of-range labels.
 * bwd: stats = the forward's (its count is the mean's denominator; NULL: every pixel counts);
 *      dsmall (ds_dtype, [B*h*w, ld_ds], columns [C, ld_ds) zeroed) = d(mean loss)/d(small) * grad_scale; gather form,
 *      deterministic.  MV_ERR_UNSUPPORTED when C > 32 (bwd) or the per-image map does not fit 64 KB of LDS: compose
 *      mv_upsample_bilinear_* with mv_cross_entropy instead. */
long mv_seg_ce_partials(int B, int H, int W);
int mv_seg_ce_fwd(const float* small, const int64_t* labels, float* lse, uint8_t* pred, float* partials, float* stats,
                  int B, int C, int h, int w, int H, int W, mv_stream_t stream);
int mv_seg_ce_bwd(const float* small, const int64_t* labels, const float* lse, const float* stats, void* dsmall,
                  int ds_dtype, int ld_ds, float grad_scale, int B, int C, int h, int w, int H, int W, mv_stream_t stream);

/* ---- image batch preparation: Normalize(ToTensor(hflip?(resize(crop(img, box), size, BILINEAR)))) ----
 * replaces the per-image torchvision/Pillow pipeline of the DataLoader worker (datasets/resisc45.py:40-69,
 * datasets/dlrsd.py:39-66, transforms/segmentation.py) on decoded uint8 frames, bit-exact to Pillow's 8-bit resampler.
 * src: uint8 [B][Hs][Ws][3] (images img_stride bytes apart).  Per image and axis the host supplies Pillow's
 * precompute_coeffs tables with crop offset / CenterCrop window folded in: bounds bh/bv int32 [B, out, 2] = (first
 * source index, tap count <= ks) and 22-bit fixed-point coefficients kh/kv int32 [B, out, ks].  flip: uint8 [B].
 * out: fp32 [B, 3, out_h, out_w] = ((pixel / 255) - mean[c]) / std[c]  (mean 0, std 1: plain ToTensor). */
int mv_image_prepare(const uint8_t* src, long img_stride, int Hs, int Ws, const int32_t* kh, const int32_t* bh,
                     const int32_t* kv, const int32_t* bv, int ks, const uint8_t* flip, float mean0, float mean1,
                     float mean2, float std0, float std1, float std2, float* out, int B, int out_h, int out_w,
                     mv_stream_t stream);
/* segmentation masks: Image.resize(NEAREST) of the crop as a gather through per-axis source index tables yi [B, out_h],
 * xi [B, out_w] (Pillow's ImagingScaleAffine indices, crop offset folded in); out int64 [B, out_h, out_w] = src + add
 * (DLRSD labels are PNG value - 1, datasets/dlrsd.py:80). */
int mv_mask_prepare(const uint8_t* src, long img_stride, int Hs, int Ws, const int32_t* yi, const int32_t* xi,
                    const uint8_t* flip, int add, int64_t* out, int B, int out_h, int out_w, mv_stream_t stream);
/* stage one of a Resize -> RandomResizedCrop chain (segmentation/data_configs/data_config.json transform_ops_train, built
 * by datasets/dlrsd.py:39-66): the same resampling with the result kept as uint8 -- out [B, out_h, out_w, 3] (frames) /
 * [B, out_h, out_w] (masks, NEAREST) -- because every Pillow resize rounds to uint8, so the second resize must start
 * from that image.  Tables as for mv_image_prepare / mv_mask_prepare; no flip, no normalisation. */
int mv_image_resize_u8(const uint8_t* src, long img_stride, int Hs, int Ws, const int32_t* kh, const int32_t* bh,
                       const int32_t* kv, const int32_t* bv, int ks, uint8_t* out, int B, int out_h, int out_w,
                       mv_stream_t stream);
int mv_mask_resize_u8(const uint8_t* src, long img_stride, int Hs, int Ws, const int32_t* yi, const int32_t* xi, uint8_t* out,
                      int B, int out_h, int out_w, mv_stream_t stream);

/* ---- optimizer: AdamW step (timm create_optimizer 'adamw' -> torch.optim.AdamW), classification/train.py:161-166,274-277 ----
 * flat fp32 arrays of n elements; decoupled weight decay; bias corrections passed in (host computes from step) */
int mv_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2, float eps,
             float weight_decay, float bias_corr1, float bias_corr2, float grad_scale, const float* clip_coef,
             mv_stream_t stream);
/* the same step with the scalars that change from step to step read from DEVICE memory: hyper = fp32 [3] = (lr, bias_corr1,
 * bias_corr2).  A launch captured in a HIP graph (the whole training step of classification/train.py:239-279 replayed as one
 * graph) then stays valid while the learning-rate schedule and the step count advance: the host rewrites three floats. */
int mv_adamw_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, float beta1, float beta2, float eps,
                 float weight_decay, float grad_scale, const float* clip_coef, mv_stream_t stream);
/* ---- gradient clipping: torch.nn.utils.clip_grad_norm_(vit.parameters(), clip_grad), classification/train.py:265-270 ----
 * over the flat gradient array g[n]: out[0] = total_norm = ||g * grad_scale||_2, out[1] = min(1, max_norm / (total_norm + 1e-6)).
 * out stays on the device; pass out + 1 as mv_adamw's clip_coef (it multiplies every gradient there: no extra pass).
 * workspace: mv_grad_norm_workspace_bytes() bytes, 16-byte aligned.  Deterministic (fixed summation order). */
size_t mv_grad_norm_workspace_bytes(void);
int mv_grad_norm_clip(const float* g, long n, float grad_scale, float max_norm, float* out, void* workspace,
                      size_t workspace_bytes, mv_stream_t stream);

/* ---- dropout: nn.Dropout(p) in training mode -- vit.py:50,52 (FeedForward), :75 (Attention.to_out), :311 (embedding) ----
 * y[i] = x[i] * keep_i / (1 - p), keep_i = (word (i & 3) of Philox4x32-10(counter = (i >> 2, offset), key = seed) >= p * 2^32).
 * The same call with (seed, offset) on the output gradient IS the backward: no mask is stored.  x == y allowed.
 * (The reference draws its mask from torch's generator: same distribution, different stream -- parity is statistical.) */
int mv_dropout(const void* x, void* y, int dtype, long n, float p, uint64_t seed, uint64_t offset, mv_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MYRTLE_VISION_HIP_H */
