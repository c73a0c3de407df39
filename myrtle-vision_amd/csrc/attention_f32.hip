// Fused fp32 attention core on the f32-input matrix cores (gfx950, v_mfma_f32_16x16x4_f32):
//
//   out[b, n, h, :] = softmax(q k^T * scale) v        (reference: Attention.forward, models/vit.py:92-96)
//
// for the EXACT-arithmetic paths that run without autograd: the converted PyTorchINT8 model (q, k, v come out of an 8-bit
// Linear as fp32, the result goes into the next 8-bit quantiser; BASELINE config 5) and evaluation in precision="fp32".
// Every product and sum is an fp32 fma / add with one rounding, exactly as in the materialised path
// (mv_gemm_f32 + mv_softmax_fwd + mv_gemm_f32), which it replaces there: the two differ only in summation ORDER.  What it
// removes is the [B, H, N, N] probability tensor (1.9 GB per layer at batch 1024) and three launches whose 197 x 197 x 64
// products ran at 15 TFLOP/s: the materialised attention was 2/3 of the converted model's forward pass.
//
// One workgroup (8 waves) per (image, head).  K [keys][64] and V^T [64][keys] of the head sit in LDS (fp32, 2 x 53 KB at
// N <= 208); a wave owns query tiles of 16 rows.  Orientation is chosen so that nothing but V is ever transposed:
//   S^T tile = K_tile (A operand) x Q^T (B operand): a lane holds ONE query (lane & 15) and, per key tile, the four keys
//       4 (lane >> 4) + r -- the softmax statistics of a row are an in-lane reduction plus two shuffles (xor 16, 32);
//   O^T tile = V^T_tile (A) x P^T (B): the B operand of k-block (T, r) is exactly the lane's probability register (T, r)
//       -- P never leaves the registers; the lane ends with four consecutive output features of its query: float4 stores.
// Both operands of a product use the same k-assignment (lane group g supplies k = 16 c + 4 g + kk in step (c, kk)), so any
// assignment gives the same sum; this one makes every fragment a 16-byte access (global for Q, LDS for K and V^T).
// LDS images: K rows of 256 B with the 16-byte slot index XORed with (key & 15); V^T rows of NK floats with the slot index
// XORed with ((d >> 2) & 3): both fragment reads are conflict-free under the ds_read_b128 banking rules.
#include "mv_common.h"

namespace {

constexpr int AF_DH = 64;

// NT = key/query tiles of 16 (13: N <= 208, i.e. 197 tokens; 17: N <= 272, i.e. 257 tokens)
// Q8 = true: the result goes straight into to_out's quint8 quantiser and leaves as int8 codes q - 128 ([B, N, H*64] int8)
template <int NT, bool Q8 = false>
__global__ __launch_bounds__(512) void attn_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N,
                                                           int H, float scale, float q_inv = 0.f, float q_zp = 0.f) {
  constexpr int NK = NT * 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Ks = smem;                    // [NK][64]
  float* const Vt = smem + NK * AF_DH;       // [64][NK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long row = 3L * H * AF_DH;           // floats between consecutive tokens
  const float* const qb = qkv + (long)b * N * row + (long)h * AF_DH;
  const float* const kb = qb + (long)H * AF_DH;
  const float* const vb = kb + (long)H * AF_DH;

  // ---- stage K (row-major, swizzled slots) and V^T; keys >= N are zero rows / columns
  for (int i = tid; i < NK * 16; i += 512) {
    const int key = i >> 4, j = i & 15;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (key < N) v = *reinterpret_cast<const f32x4*>(kb + key * row + 4 * j);
    *reinterpret_cast<f32x4*>(Ks + key * AF_DH + ((j ^ (key & 15)) << 2)) = v;
  }
  for (int i = tid; i < NK * 16; i += 512) {
    const int key = i % NK, j = i / NK;      // lanes walk the keys: the transposed writes below are conflict-free
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (key < N) v = *reinterpret_cast<const f32x4*>(vb + key * row + 4 * j);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int d = 4 * j + e;
      Vt[d * NK + ((((key >> 2) ^ ((d >> 2) & 3))) << 2) + (key & 3)] = v[e];
    }
  }
  __syncthreads();

  const int q16 = lane & 15, g = lane >> 4;
  for (int qt = wave; qt * 16 < N; qt += 8) {
    const int q = qt * 16 + q16;
    const float* qrow = qb + (long)(q < N ? q : N - 1) * row + 4 * g;       // clamped: rows >= N are never stored
    f32x4 qf[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) qf[c] = *reinterpret_cast<const f32x4*>(qrow + 16 * c);

    // ---- S^T = K Q^T: acc[T][r] = score of (query q, key 16 T + 4 g + r).  Fragments are prefetched ONE tile ahead by hand
    // and sched_barriers keep the compiler from hoisting all 52 tile loads to the top (that version spilled 255 VGPRs).
    f32x4 acc[NT];
    f32x4 kf[4], kn[4];
    auto load_k = [&](f32x4 (&dst)[4], int T) {
      const int key = T * 16 + q16;          // this lane's row of the A operand
#pragma unroll
      for (int c = 0; c < 4; ++c)
        dst[c] = *reinterpret_cast<const f32x4*>(Ks + key * AF_DH + (((4 * c + g) ^ (key & 15)) << 2));
    };
    load_k(kf, 0);
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      if (T + 1 < NT) load_k(kn, T + 1);
      acc[T] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c][kk], qf[c][kk], acc[T], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) kf[c] = kn[c];
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---- softmax over the keys (vit.py:92-93: (q k^T) * scale, then softmax): max, exp, sum, normalise
    float mx = -INFINITY;
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = T * 16 + 4 * g + r;
        const float x = key < N ? acc[T][r] * scale : -INFINITY;
        acc[T][r] = x;
        mx = fmaxf(mx, x);
      }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float s = 0.f;
#pragma unroll
    for (int T = 0; T < NT; ++T)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = expf(acc[T][r] - mx);      // exp(-inf) = 0 for the padded keys
        acc[T][r] = e;
        s += e;
      }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float inv = 1.0f / s;
#pragma unroll
    for (int T = 0; T < NT; ++T) acc[T] *= inv;

    // ---- O^T = V^T P^T: o[dt][r] = out(query q, feature 16 dt + 4 g + r)
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 vf[4], vn[4];
    auto load_v = [&](f32x4 (&dst)[4], int T) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int d = dt * 16 + q16;
        dst[dt] = *reinterpret_cast<const f32x4*>(Vt + d * NK + (((4 * T + g) ^ ((d >> 2) & 3)) << 2));
      }
    };
    load_v(vf, 0);
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      if (T + 1 < NT) load_v(vn, T + 1);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[dt][r], acc[T][r], o[dt], 0, 0, 0);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) vf[dt] = vn[dt];
      __builtin_amdgcn_sched_barrier(0);
    }
    if (q < N) {
      if constexpr (Q8) {
        int8_t* op = reinterpret_cast<int8_t*>(out) + ((long)b * N + q) * H * AF_DH + (long)h * AF_DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          *reinterpret_cast<unsigned*>(op + 16 * dt) = affine_i8_pack4<0>(o[dt][0], o[dt][1], o[dt][2], o[dt][3], q_inv, q_zp);
      } else {
        float* op = out + ((long)b * N + q) * H * AF_DH + (long)h * AF_DH + 4 * g;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(op + 16 * dt) = o[dt];
      }
    }
  }
}

template <int NT, bool Q8>
int launch_attn_f32(const float* qkv, void* out, int B, int N, int H, float scale, float q_inv, float q_zp, hipStream_t s) {
  constexpr size_t lds = (size_t)2 * NT * 16 * AF_DH * sizeof(float);
  static const int attr = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_f32_kernel<NT, Q8>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : -1;
  if (attr) return MV_ERR_LAUNCH;
  attn_fwd_f32_kernel<NT, Q8><<<B * H, 512, lds, s>>>(qkv, (float*)out, N, H, scale, q_inv, q_zp);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

}  // namespace

extern "C" int mv_attention_fwd_f32(const float* qkv, float* out, int B, int N, int H, float scale, mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && N > 0 && H > 0 && (long)B * H < (1L << 31), MV_ERR_SHAPE);
  MV_REQUIRE(N <= 272, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(mv_aligned16(qkv) && mv_aligned16(out), MV_ERR_ALIGN);
  if (B == 0) return MV_OK;
  hipStream_t s = (hipStream_t)stream;
  return N <= 208 ? launch_attn_f32<13, false>(qkv, out, B, N, H, scale, 0.f, 0.f, s)
                  : launch_attn_f32<17, false>(qkv, out, B, N, H, scale, 0.f, 0.f, s);
}

extern "C" int mv_attention_fwd_f32_q8(const float* qkv, void* codes, int B, int N, int H, float scale, float q_scale,
                                       int q_zero_point, mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && N > 0 && H > 0 && (long)B * H < (1L << 31) && q_scale > 0.f, MV_ERR_SHAPE);
  MV_REQUIRE(N <= 272 && q_zero_point >= 0 && q_zero_point <= 255, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(mv_aligned16(qkv) && mv_aligned16(codes), MV_ERR_ALIGN);
  if (B == 0) return MV_OK;
  hipStream_t s = (hipStream_t)stream;
  const float inv = 1.0f / q_scale, fz = (float)q_zero_point;
  return N <= 208 ? launch_attn_f32<13, true>(qkv, codes, B, N, H, scale, inv, fz, s)
                  : launch_attn_f32<17, true>(qkv, codes, B, N, H, scale, inv, fz, s);
}
