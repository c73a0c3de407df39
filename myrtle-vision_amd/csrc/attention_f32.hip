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
// XORed with vt_sw(d) = {0, 0, 3, 3}[(d >> 2) & 3]: both fragment reads are conflict-free under the ds_read_b128 banking rules
// (MI355X_MICROARCH.md: a b128 read is served in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... -- a group
// holds every row d & 15 once, with lane group g or g ^ 1 depending on whether (d >> 2) & 3 is 1 or 2; the plain XOR with
// (d >> 2) & 3, right for 16 CONTIGUOUS lanes, was 2-way conflicted: SQ_LDS_BANK_CONFLICT = 1.3 x the kernel's LDS-active cycles).
#include "mv_common.h"

namespace {

constexpr int AF_DH = 64;
__device__ __forceinline__ int vt_sw(int d) { return (0xF0 >> (((d >> 2) & 3) * 2)) & 3; }   // {0, 0, 3, 3}[(d >> 2) & 3]
constexpr float AF_LOG2E = 1.4426950408889634f, AF_LN2 = 0.6931471805599453f;

// NT = key/query tiles of 16 (13: N <= 208, i.e. 197 tokens; 17: N <= 272, i.e. 257 tokens)
// Q8 = true: the result goes straight into to_out's quint8 quantiser and leaves as int8 codes q - 128 ([B, N, H*64] int8)
template <int NT, bool Q8 = false>
__global__ __launch_bounds__(512) void attn_fwd_f32_kernel(const float* __restrict__ qkv, float* __restrict__ out, int N,
                                                           int H, float scale, float q_inv = 0.f, float q_zp = 0.f,
                                                           float* __restrict__ lse = nullptr, int n_items = 0) {
  constexpr int NK = NT * 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const Ks = smem;                    // [NK][64]
  float* const Vt = smem + NK * AF_DH;       // [64][NK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long row = 3L * H * AF_DH;           // floats between consecutive tokens

  // ---- staging of K (row-major, swizzled slots) and V^T of one (image, head); keys >= N are zero rows / columns.
  // The kernel is PERSISTENT (grid = #CUs, one workgroup per CU by LDS): a workgroup walks items blockIdx.x, + gridDim.x, ...
  // and the 16-byte global loads of the NEXT item's K and V (7 + 7 per thread at 197 tokens) are issued before the current
  // item's products and land under them; only the LDS writes (between two barriers) stay exposed.  With one item per
  // workgroup the staging loads were 108 of 511 us at batch 256 (phase ablation, tools/ablate_attn_f32.sh): nothing else
  // runs on a CU whose only workgroup waits for memory.  NT = 17 (257 tokens) has no registers to spare: its next item's
  // loads are issued after the current item's products.
  constexpr int NIT = (NK * 16 + 511) / 512;
  constexpr bool PREFETCH = NT <= 13;
  f32x4 kst[NIT], vst[NIT];
  auto issue_loads = [&](int item) {
    const int ib = item / H, ih = item % H;
    const float* const kb = qkv + (long)ib * N * row + (long)ih * AF_DH + (long)H * AF_DH;
    const float* const vb = kb + (long)H * AF_DH;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + 512 * it;
      const int kkey = i >> 4, kj = i & 15;
      const int vkey = i % NK, vj = i / NK;
      kst[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      vst[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < NK * 16 && kkey < N) kst[it] = *reinterpret_cast<const f32x4*>(kb + kkey * row + 4 * kj);
      if (i < NK * 16 && vkey < N) vst[it] = *reinterpret_cast<const f32x4*>(vb + vkey * row + 4 * vj);
    }
  };
  auto write_lds = [&]() {
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + 512 * it;
      if (i < NK * 16) {
        const int key = i >> 4, j = i & 15;
        *reinterpret_cast<f32x4*>(Ks + key * AF_DH + ((j ^ (key & 15)) << 2)) = kst[it];
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + 512 * it;
      if (i < NK * 16) {
        const int key = i % NK, j = i / NK;    // lanes walk the keys: the transposed writes below are conflict-free
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int d = 4 * j + e;
          Vt[d * NK + ((((key >> 2) ^ vt_sw(d))) << 2) + (key & 3)] = vst[it][e];
        }
      }
    }
  };
  issue_loads(blockIdx.x);
  write_lds();
  __syncthreads();

  const int q16 = lane & 15, g = lane >> 4;
  const float sl2 = scale * AF_LOG2E;
  for (int item = blockIdx.x; item < n_items; item += gridDim.x) {
  const int b = item / H, h = item % H;
  const float* const qb = qkv + (long)b * N * row + (long)h * AF_DH;
  const bool has_next = item + (int)gridDim.x < n_items;      // workgroup-uniform
  if (PREFETCH && has_next) issue_loads(item + gridDim.x);
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
        for (int kk = 0; kk < 4; ++kk) {
          acc[T] = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c][kk], qf[c][kk], acc[T], 0, 0, 0);
        }
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
        const float x = key < N ? acc[T][r] * sl2 : -INFINITY;        // scores in log2 units: exp2 is one instruction
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
        const float e = __builtin_amdgcn_exp2f(acc[T][r] - mx);      // exp2(-inf) = 0 for the padded keys
        acc[T][r] = e;
        s += e;
      }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float inv = 1.0f / s;
#pragma unroll
    for (int T = 0; T < NT; ++T) acc[T] *= inv;
    if (lse && g == 0 && q < N) lse[((long)b * H + h) * N + q] = (mx + __builtin_amdgcn_logf(s)) * AF_LN2;     // ln sum_j exp(scale q.k_j)

    // ---- O^T = V^T P^T: o[dt][r] = out(query q, feature 16 dt + 4 g + r)
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 vf[4], vn[4];
    auto load_v = [&](f32x4 (&dst)[4], int T) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const int d = dt * 16 + q16;
        dst[dt] = *reinterpret_cast<const f32x4*>(Vt + d * NK + (((4 * T + g) ^ vt_sw(d)) << 2));
      }
    };
    load_v(vf, 0);
#pragma unroll
    for (int T = 0; T < NT; ++T) {
      if (T + 1 < NT) load_v(vn, T + 1);
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          o[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[dt][r], acc[T][r], o[dt], 0, 0, 0);
        }
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
  if (has_next) {
    __syncthreads();                          // every wave is done reading this item's K / V^T
    if (!PREFETCH) issue_loads(item + gridDim.x);
    write_lds();
    __syncthreads();
  }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// Backward of the same core in fp32 (training in precision="fp32"): dq, dk, dv from q, k, v, out, d(out) and the saved
// log-sum-exp -- no [B, H, N, N] tensor in either direction (the materialised path wrote and re-read 1.4 GB of scores,
// probabilities and their gradients per layer at batch 64, with 197 x 197 x 64 products at 27-47 TFLOP/s).
// One workgroup (8 waves) per (image, head), two passes over the SAME two LDS regions, no barrier inside a pass:
//   pass A (dQ):    R0 = K, R1 = V (row-major, the forward kernel's K image).  A wave owns 16 queries: S^T = K Q^T and
//                   dP^T = V dO^T (a lane holds one query and four keys per key tile), P^T = exp(scale S^T - lse),
//                   delta = rowsum(dO * out) from the query's own fragments, dS^T = P^T (dP^T - delta) scale is at once the
//                   A operand of dQ[q][d] += dS[q][key] K[key][d] (B operand: 4-byte reads of K rows).
//   pass B (dK/dV): R0 = Q, R1 = dO.  A wave owns 16 keys (K, V fragments in registers): S = Q K^T and dP = dO V^T (a lane
//                   holds one key and four queries per query tile), P and dS are the A operands of dV[key][d] += P[q][key]
//                   dO[q][d] and dK[key][d] += dS[q][key] Q[q][d].
// S and dP are computed in both passes (7 products instead of 5): the f32 matrix pipe is the bound either way, and nothing
// crosses between waves but delta (written in pass A, read after the one barrier pair between the passes).
// Same k-assignment rule as the forward kernel: lane group g supplies k = 16 c + 4 g + kk in step (c, kk) on both operands.
// ---------------------------------------------------------------------------------------------------------------------
template <int NT>
__global__ __launch_bounds__(512) void attn_bwd_f32_kernel(const float* __restrict__ qkv, const float* __restrict__ out,
                                                           const float* __restrict__ dout, const float* __restrict__ lse,
                                                           float* __restrict__ dqkv, int N, int H, float scale) {
  constexpr int NK = NT * 16;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float* const R0 = smem;                    // [NK][64]: K, then Q
  float* const R1 = smem + NK * AF_DH;       // [NK][64]: V, then dO
  float* const sLse = R1 + NK * AF_DH;       // [NK]; +inf for rows >= N (their probabilities are exactly 0)
  float* const sDelta = sLse + NK;           // [NK]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int q16 = lane & 15, g = lane >> 4;
  const float sl2 = scale * AF_LOG2E;
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long row = 3L * H * AF_DH, orow = (long)H * AF_DH;
  const float* const qb = qkv + (long)b * N * row + (long)h * AF_DH;
  const float* const kb = qb + orow;
  const float* const vb = kb + orow;
  const float* const ob = out + (long)b * N * orow + (long)h * AF_DH;
  const float* const dob = dout + (long)b * N * orow + (long)h * AF_DH;
  float* const dqb = dqkv + (long)b * N * row + (long)h * AF_DH;

  // rows of two [N][64] operands -> R0, R1 (swizzled 16-byte slots, rows >= N zero); every load of the thread is issued
  // before the first LDS write (see the forward kernel)
  constexpr int NIT = (NK * 16 + 511) / 512;
  auto stage2 = [&](const float* src0, long ld0, const float* src1, long ld1) {
    f32x4 a[NIT], c[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + 512 * it, r = i >> 4, j = i & 15;
      a[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      c[it] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (i < NK * 16 && r < N) {
        a[it] = *reinterpret_cast<const f32x4*>(src0 + r * ld0 + 4 * j);
        c[it] = *reinterpret_cast<const f32x4*>(src1 + r * ld1 + 4 * j);
      }
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int i = tid + 512 * it, r = i >> 4, j = i & 15;
      if (i < NK * 16) {
        *reinterpret_cast<f32x4*>(R0 + r * AF_DH + ((j ^ (r & 15)) << 2)) = a[it];
        *reinterpret_cast<f32x4*>(R1 + r * AF_DH + ((j ^ (r & 15)) << 2)) = c[it];
      }
    }
  };
  // A-operand fragments of the 16 rows of tile T: lane (row q16, group g) gets floats 16 c + 4 g .. + 3
  auto load_rows = [&](f32x4 (&dst)[4], const float* R, int T) {
    const int r = T * 16 + q16;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = *reinterpret_cast<const f32x4*>(R + r * AF_DH + (((4 * c + g) ^ (r & 15)) << 2));
  };
  // B operand of the products that contract over ROWS: element (row 16 T + 4 g + r, column 16 dt + q16)
  auto elem = [&](const float* R, int T, int r, int dt) -> float {
    const int rr = T * 16 + 4 * g + r, d = 16 * dt + q16;
    return R[rr * AF_DH + ((((d >> 2) ^ (rr & 15)) << 2) | (d & 3))];
  };
  // B-operand fragments of one global row (clamped; masked through lse / key < N)
  auto load_global = [&](f32x4 (&dst)[4], const float* base, long ld, int r) {
    const float* p = base + (long)(r < N ? r : N - 1) * ld + 4 * g;
#pragma unroll
    for (int c = 0; c < 4; ++c) dst[c] = *reinterpret_cast<const f32x4*>(p + 16 * c);
  };

  stage2(kb, row, vb, row);
  for (int i = tid; i < NK; i += 512) {
    sLse[i] = i < N ? lse[((long)b * H + h) * N + i] * AF_LOG2E : INFINITY;     // log2 units, like the scores below
    sDelta[i] = 0.f;
  }
  __syncthreads();

  // ---------------- pass A: dQ ----------------
  for (int qt = wave; qt * 16 < N; qt += 8) {
    const int q = qt * 16 + q16;
    f32x4 qf[4], dof[4], of[4];
    load_global(qf, qb, row, q);
    load_global(dof, dob, orow, q);
    load_global(of, ob, orow, q);
    float dl = 0.f;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) dl += dof[c][e] * of[c][e];
    dl += __shfl_xor(dl, 16, 64);
    dl += __shfl_xor(dl, 32, 64);
    if (g == 0) sDelta[q] = dl;                  // q < NK always; pass B reads it after the barrier
    const float l = sLse[q];
    f32x4 dq[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) dq[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    f32x4 kf[4], vf[4], kn[4], vn[4];
    load_rows(kf, R0, 0);
    load_rows(vf, R1, 0);
#pragma unroll 1
    for (int T = 0; T < NT; ++T) {
      if (T + 1 < NT) {
        load_rows(kn, R0, T + 1);
        load_rows(vn, R1, T + 1);
      }
      f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          st = __builtin_amdgcn_mfma_f32_16x16x4f32(kf[c][kk], qf[c][kk], st, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(vf[c][kk], dof[c][kk], dp, 0, 0, 0);
        }
      f32x4 ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = T * 16 + 4 * g + r;
        const float p = key < N ? __builtin_amdgcn_exp2f(st[r] * sl2 - l) : 0.f;
        ds[r] = p * (dp[r] - dl) * scale;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dq[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[r], elem(R0, T, r, dt), dq[dt], 0, 0, 0);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        kf[c] = kn[c];
        vf[c] = vn[c];
      }
    }
    // dq[dt][r] = dQ(query 16 qt + 4 g + r, feature 16 dt + q16)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = qt * 16 + 4 * g + r;
      if (qq < N) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dqb[(long)qq * row + 16 * dt + q16] = dq[dt][r];
      }
    }
  }

  // ---------------- the same LDS bytes now hold Q and dO ----------------
  __syncthreads();
  stage2(qb, row, dob, orow);
  __syncthreads();

  // ---------------- pass B: dK, dV ----------------
  for (int kt = wave; kt * 16 < N; kt += 8) {
    const int key = kt * 16 + q16;
    f32x4 kf[4], vf[4];
    load_global(kf, kb, row, key);
    load_global(vf, vb, row, key);
    f32x4 dk[4], dv[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
    f32x4 qa[4], da[4], qn[4], dn[4];
    load_rows(qa, R0, 0);
    load_rows(da, R1, 0);
#pragma unroll 1
    for (int T = 0; T < NT; ++T) {
      if (T + 1 < NT) {
        load_rows(qn, R0, T + 1);
        load_rows(dn, R1, T + 1);
      }
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(sLse + T * 16 + 4 * g);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(sDelta + T * 16 + 4 * g);
      f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          sv = __builtin_amdgcn_mfma_f32_16x16x4f32(qa[c][kk], kf[c][kk], sv, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x4f32(da[c][kk], vf[c][kk], dp, 0, 0, 0);
        }
      f32x4 p, ds;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = key < N ? __builtin_amdgcn_exp2f(sv[r] * sl2 - l4[r]) : 0.f;   // lse = +inf for padded queries: 0
        p[r] = e;
        ds[r] = e * (dp[r] - d4[r]) * scale;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dv[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(p[r], elem(R1, T, r, dt), dv[dt], 0, 0, 0);
          dk[dt] = __builtin_amdgcn_mfma_f32_16x16x4f32(ds[r], elem(R0, T, r, dt), dk[dt], 0, 0, 0);
        }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        qa[c] = qn[c];
        da[c] = dn[c];
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int kk = kt * 16 + 4 * g + r;
      if (kk < N) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dqb[(long)kk * row + orow + 16 * dt + q16] = dk[dt][r];
          dqb[(long)kk * row + 2 * orow + 16 * dt + q16] = dv[dt][r];
        }
      }
    }
  }
}

template <int NT>
int launch_attn_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv, int B, int N,
                        int H, float scale, hipStream_t s) {
  constexpr size_t lds = ((size_t)2 * NT * 16 * AF_DH + 2 * NT * 16) * sizeof(float);
  const int attr = MV_ONCE_PER_DEVICE(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_f32_kernel<NT>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : -1);
  if (attr) return MV_ERR_LAUNCH;
  attn_bwd_f32_kernel<NT><<<B * H, 512, lds, s>>>(qkv, out, dout, lse, dqkv, N, H, scale);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

template <int NT, bool Q8>
int launch_attn_f32(const float* qkv, void* out, int B, int N, int H, float scale, float q_inv, float q_zp, hipStream_t s,
                    float* lse = nullptr) {
  constexpr size_t lds = (size_t)2 * NT * 16 * AF_DH * sizeof(float);
  const int attr = MV_ONCE_PER_DEVICE(hipFuncSetAttribute(reinterpret_cast<const void*>(attn_fwd_f32_kernel<NT, Q8>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) == hipSuccess ? 0 : -1);
  if (attr) return MV_ERR_LAUNCH;
  const int n_cu = mv_cu_count();                         // of the current device (one cached value per device)
  const int items = B * H;
  attn_fwd_f32_kernel<NT, Q8><<<items < n_cu ? items : n_cu, 512, lds, s>>>(qkv, (float*)out, N, H, scale, q_inv, q_zp, lse,
                                                                             items);
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

extern "C" int mv_attention_fwd_f32_lse(const float* qkv, float* out, float* lse, int B, int N, int H, float scale,
                                        mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && N > 0 && H > 0 && (long)B * H < (1L << 31), MV_ERR_SHAPE);
  MV_REQUIRE(N <= 272, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(mv_aligned16(qkv) && mv_aligned16(out) && lse, MV_ERR_ALIGN);
  if (B == 0) return MV_OK;
  hipStream_t s = (hipStream_t)stream;
  return N <= 208 ? launch_attn_f32<13, false>(qkv, out, B, N, H, scale, 0.f, 0.f, s, lse)
                  : launch_attn_f32<17, false>(qkv, out, B, N, H, scale, 0.f, 0.f, s, lse);
}

extern "C" int mv_attention_bwd_f32(const float* qkv, const float* out, const float* dout, const float* lse, float* dqkv,
                                    int B, int N, int H, float scale, mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && N > 0 && H > 0 && (long)B * H < (1L << 31), MV_ERR_SHAPE);
  MV_REQUIRE(N <= 272, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(mv_aligned16(qkv) && mv_aligned16(out) && mv_aligned16(dout) && mv_aligned16(dqkv) && lse, MV_ERR_ALIGN);
  if (B == 0) return MV_OK;
  hipStream_t s = (hipStream_t)stream;
  return N <= 208 ? launch_attn_bwd_f32<13>(qkv, out, dout, lse, dqkv, B, N, H, scale, s)
                  : launch_attn_bwd_f32<17>(qkv, out, dout, lse, dqkv, B, N, H, scale, s);
}
