// Fused multi-head self-attention core, forward and backward (reference: Attention.forward vit.py:87-96:
//   qkv.reshape(b,n,3,H,dh).permute(2,0,3,1,4); softmax(q k^T * dh^-0.5) v; .transpose(1,2).reshape(b,n,c)).
//
// ViT sequences are short (N = 197 at 224^2, 257 at 256^2) and dim_head = 64, so one workgroup owns one
// (image, head): its whole K and V (N x 64 bf16 = 25 KB each) sit in LDS, nothing is tiled over the sequence
// and no online-softmax rescaling is needed.  All five/two products run on v_mfma_f32_16x16x32_bf16.
//
// Orientation is chosen so computed tiles never need a transpose through LDS (cdna guide section 3,
// "an accumulator tile as the next MFMA's operand"):
//   forward : S^T[key][query] = K Q^T (keys on accumulator rows).  The softmax reduction over keys is then
//             in-lane (4 regs x tiles) + two wave shuffles (xor 16, 32), and P^T is already the B operand of
//             O^T[d][query] = V^T P^T; V^T fragments come from the row-major V tile by ds_read_b64_tr_b16.
//   backward: S[query][key] and dP[query][key] (keys on lanes).  P and dS are then already the B operands of
//             dV^T += dO^T P and dK^T += Q^T dS (contraction over accumulator rows = queries); only dS crosses
//             LDS once (stored transposed, [key][32 queries]) for dQ^T = K^T dS^T.
// The k order inside such an accumulator-fed MFMA is permuted (slot 8g+j <-> row 4g+j of tile 0 | tile 1);
// the LDS-side fragments are gathered in the same order (rows 4g+q of each 16-row tile), which is also the
// conflict-free order for transposed reads of 128-byte rows in the sw128 image (tools/lds_bank_sim.py).
#include "mv_common.h"

#include <atomic>


namespace {

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

__device__ __forceinline__ int sw128(int row, int ch) { return row * 128 + ((ch ^ (((row >> 1) & 3) << 1)) << 4); }
// dS^T image: [key][32 queries] bf16 = 64-byte rows; the two 32-byte halves swap on rows 4..7 (mod 8)
__device__ __forceinline__ int swds(int key, int half) { return key * 64 + ((half ^ ((key >> 2) & 1)) << 5); }
// attn_bwd4_kernel's image of the same tile: the 8-byte atom (query tile t, query group gq = 4 queries) of row `key` sits at
// slot (4 t + gq) ^ Y((key >> 1) & 7), Y(k2 k1 k0) = (k1 k2 k0).  Banking rules (MI355X_MICROARCH.md, LDS table): ds_write_b64
// goes in 4 groups of 16 contiguous lanes over 32 banks, ds_read_b64_tr_b16 in 2 x 32 lanes over 64.  A store touches
// {16 keys} x {gq = lane >> 4}: within 16 lanes the 8 keys of one parity share a 128-byte window and need 8 different slots
// (Y is a bijection of the key pair index); a transposed read touches {rows 4 g + q} x {gq = lane & 3}: rows r and r + 4 share a
// 256-byte window and need opposite slot halves (bit 2 of Y follows bit 1 of its argument).  swds() above is conflict-free
// for the reads but 4-way conflicted for the stores (16 cycles instead of 4): SQ_LDS_BANK_CONFLICT = half of that kernel's
// LDS-active cycles, all from those stores by phase ablation under the counter; this image: both at their ideal.
__device__ __forceinline__ int dsy(int k) { return (((k >> 1) & 1) << 2) | (((k >> 2) & 1) << 1) | (k & 1); }
__device__ __forceinline__ int swds4(int key, int t, int gq) { return key * 64 + (((4 * t + gq) ^ dsy((key >> 1) & 7)) << 3); }

__device__ __forceinline__ bf16x8 cat8(bf16x4 a, bf16x4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ bf16x4 tr_read(const char* p) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, p));
}
__device__ __forceinline__ bf16x8 pack8(f32x4 a, f32x4 b) {
  bf16x8 r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3], (bf16_t)b[0], (bf16_t)b[1], (bf16_t)b[2], (bf16_t)b[3]};
  return r;
}
__device__ __forceinline__ bf16x4 pack4(f32x4 a) {
  bf16x4 r = {(bf16_t)a[0], (bf16_t)a[1], (bf16_t)a[2], (bf16_t)a[3]};
  return r;
}
// F16 forms (round 4, precision "bf16x3"): the SAME kernels on IEEE-half operands -- 11 significand bits instead of 8, the same
// 2-byte geometry, LDS images, fragment maps and MFMA rate (v_mfma_f32_16x16x32_f16).  Fragments stay in their bf16x8 / bf16x4
// containers (they are only moved); what changes is the matrix instruction and every float -> element conversion.
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8_t;
typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
template <bool F16>
__device__ __forceinline__ f32x4 mma32(bf16x8 a, bf16x8 b, f32x4 c, int, int, int) {
  if constexpr (F16)
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8_t, a), __builtin_bit_cast(f16x8_t, b), c, 0, 0, 0);
  else
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ bf16x8 pack8t(f32x4 a, f32x4 b) {
  if constexpr (F16) {
    const f16x8_t r = {(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3],
                       (_Float16)b[0], (_Float16)b[1], (_Float16)b[2], (_Float16)b[3]};
    return __builtin_bit_cast(bf16x8, r);
  } else {
    return pack8(a, b);
  }
}
template <bool F16>
__device__ __forceinline__ bf16x4 pack4t(f32x4 a) {
  if constexpr (F16) {
    const f16x4_t r = {(_Float16)a[0], (_Float16)a[1], (_Float16)a[2], (_Float16)a[3]};
    return __builtin_bit_cast(bf16x4, r);
  } else {
    return pack4(a);
  }
}
// Outputs leave as 16-byte stores: a lane holds 4 consecutive features (8 bytes) of each 16-feature tile; for an adjacent
// tile pair v_permlane16_swap (lanes l <-> l ^ 16, same row) leaves lane group g with 8 consecutive features of tile
// (g & 1), starting at feature 8 (g >> 1) -- a row's four lanes then cover 64 contiguous bytes per instruction instead of
// two 32-byte pieces in two instructions (the NT epilogue's trick; partial-sector accesses are what hurt, DESIGN finding 23).
// Must be executed by every lane of the wave.
__device__ __forceinline__ u32x4 pair16(f32x4 a, f32x4 b) {
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const u32x2_t pa = __builtin_bit_cast(u32x2_t, pack4(a)), pb = __builtin_bit_cast(u32x2_t, pack4(b));
  const u32x2_t r0 = __builtin_amdgcn_permlane16_swap(pa[0], pb[0], false, false);
  const u32x2_t r1 = __builtin_amdgcn_permlane16_swap(pa[1], pb[1], false, false);
  return (u32x4){r0[0], r1[0], r0[1], r1[1]};
}
// feature offset of that vector inside the tile pair starting at tile j0: 16 (j0 + (g & 1)) + 8 (g >> 1)
__device__ __forceinline__ int pair16_off(int j0, int g) { return 16 * (j0 + (g & 1)) + 8 * (g >> 1); }
// Per-lane LDS offsets of the fragment reads.  Every tile base used below is a multiple of 16 rows, so the swizzle
// term of sw128 -- a function of (row>>1)&3 -- depends on the LANE only, and every fragment address is
// "tile + row_base*128 + lane constant (+ 2048 for the second half of a transposed fragment)".  Computing sw128() per
// read instead cost ~150 VALU instructions per (query pair, key tile) against 16 MFMAs: the kernels were issue-bound.
struct LaneOff {
  int rf[2];   // row fragment, k-step 0 / 1: row (lane&15), chunk (4 ks + lane>>4) ^ swz
  int tr[4];   // transposed fragment for d-tile 0..3: row 4g+q, chunk (2 dt + p>>1) ^ swz, + 8 (p&1)
};
__device__ __forceinline__ LaneOff make_lane_off(int lane) {
  LaneOff L;
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int fr = (((lane & 15) >> 1) & 3) << 1;
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) L.rf[ks] = (lane & 15) * 128 + (((4 * ks + g) ^ fr) << 4);
  const int rt = 4 * g + q, ft = ((rt >> 1) & 3) << 1;
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) L.tr[dt] = rt * 128 + (((2 * dt + (p >> 1)) ^ ft) << 4) + 8 * (p & 1);
  return L;
}
// transposed fragment of d-tile dt (pass L.tr[dt]): column (lane&15) of its 16 columns; k slots 8g+e <-> rows
// row_base + 4g + e (e<4) and row_base + 16 + 4g + (e-4); row_base % 16 == 0
__device__ __forceinline__ bf16x8 tr_frag128(const char* tile, int row_base, int tr_off) {
  const char* p0 = tile + row_base * 128 + tr_off;
  return cat8(tr_read(p0), tr_read(p0 + 2048));
}
// row fragment: 8 consecutive k (k-step ks of 32) of row (row_base + lane&15); row_base % 16 == 0
__device__ __forceinline__ bf16x8 row_frag128(const char* tile, int row_base, int rf_off) {
  return *reinterpret_cast<const bf16x8*>(tile + row_base * 128 + rf_off);
}

// K and V of one head, global -> LDS by DMA ([rows][64] bf16 = 128-byte rows, the sw128 image; one wave-instruction =
// 8 rows).  Every piece is in flight at once and no register is touched.  (The register path -- load, select zero for
// rows >= N, ds_write -- compiled to load / s_waitcnt vmcnt(0) / ds_write once per 16 bytes: ~14 exposed memory round
// trips per thread.)  Rows >= N are CLAMPED to row N-1, not zeroed: padded keys are masked (`key < N`: p = 0 exactly)
// and padded queries are either never stored (forward) or masked through lse = +inf (backward), so their
// contributions are 0 x finite.  The caller waits (s_waitcnt vmcnt(0)) and barriers before reading.
__device__ __forceinline__ void stage_kv_dma(const bf16_t* base, long D, int N, char* sK, int rows_k, char* sV, int rows_v,
                                             int wave, int nwaves, int lane) {
  const int prow = lane >> 3, pch = lane & 7;
  for (int pc = wave; pc < rows_k / 8; pc += nwaves) {
    const int row = 8 * pc + prow;
    const long rr = row < N ? row : N - 1;
    const int ch = pch ^ (((row >> 1) & 3) << 1);
    glds16(base + rr * 3 * D + D + ch * 8, sK + pc * 1024);
    if (pc < rows_v / 8) glds16(base + rr * 3 * D + 2 * D + ch * 8, sV + pc * 1024);
  }
}

// ------------------------------------------------------------------------------------------------
// forward: one 256-thread workgroup per (image, head); waves take 16-query tiles round-robin
// ------------------------------------------------------------------------------------------------
template <int NKT, bool F16 = false>   // F16: half operands, fp32 output (precision "bf16x3h")
__global__ __launch_bounds__(256, (NKT <= 18 ? 2 : 1)) void attn_fwd_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                          float* __restrict__ lse, int N, int H, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NKT * 16;
  char* sK = smem;
  char* sV = smem + NP * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const LaneOff L = make_lane_off(lane);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long D = (long)H * 64;
  const bf16_t* base = qkv + (long)b * N * 3 * D + h * 64;
  stage_kv_dma(base, D, N, sK, NP, sV, NP, wave, 4, lane);
  const int nqt = (N + 15) >> 4;
  // Q fragments straight from global memory, one 16-query tile ahead: the next tile's loads are in flight while this
  // one computes (rows >= N clamped: those outputs are not stored)
  auto load_q = [&](int qt, int ks) -> u32x4 {
    const int qrow = qt * 16 + (lane & 15);
    return *reinterpret_cast<const u32x4*>(base + (long)(qrow < N ? qrow : N - 1) * 3 * D + 32 * ks + 8 * g);
  };
  u32x4 qn0 = load_q(wave < nqt ? wave : 0, 0), qn1 = load_q(wave < nqt ? wave : 0, 1);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's K/V DMA pieces (and its first Q tile) have landed
  __syncthreads();

  for (int qt = wave; qt < nqt; qt += 4) {
    const int qrow = qt * 16 + (lane & 15);
    bf16x8 qf[2];
    qf[0] = __builtin_bit_cast(bf16x8, qn0);
    qf[1] = __builtin_bit_cast(bf16x8, qn1);
    if (qt + 4 < nqt) {
      qn0 = load_q(qt + 4, 0);
      qn1 = load_q(qt + 4, 1);
    }
    f32x4 st[NKT];
    float mx = -INFINITY;
    // key tiles in PAIRS with their two-step accumulation chains interleaved (a0 b0 a1 b1): issued tile by tile, every
    // second MFMA waited for the one before it (this phase took 41 us against 20 for the equally large P.V products,
    // whose four accumulators are independent)
#pragma unroll
    for (int kp = 0; kp < NKT; kp += 2) {
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      {
        const bf16x8 ka0 = row_frag128(sK, kp * 16, L.rf[0]), kb0 = row_frag128(sK, (kp + 1) * 16, L.rf[0]);
        const bf16x8 ka1 = row_frag128(sK, kp * 16, L.rf[1]), kb1 = row_frag128(sK, (kp + 1) * 16, L.rf[1]);
        acc[0] = mma32<F16>(ka0, qf[0], acc[0], 0, 0, 0);
        acc[1] = mma32<F16>(kb0, qf[0], acc[1], 0, 0, 0);
        acc[0] = mma32<F16>(ka1, qf[1], acc[0], 0, 0, 0);
        acc[1] = mma32<F16>(kb1, qf[1], acc[1], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = (kp + t) * 16 + 4 * g + r;
          const float v = key < N ? acc[t][r] * scale_log2e : -INFINITY;
          acc[t][r] = v;
          mx = fmaxf(mx, v);
        }
        st[kp + t] = acc[t];
      }
      // left alone the compiler hoists every tile's K reads above the first MFMA (and, at 18 key tiles, spills 16 VGPRs):
      // 98.7 -> 95 us at 197 tokens, 188 -> 157 us at 257
      __builtin_amdgcn_sched_barrier(0);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(st[kt][r] - mx);
        st[kt][r] = p;
        sum += p;
      }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);

    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NKT / 2; ++u) {
      const bf16x8 pf = pack8t<F16>(st[2 * u], st[2 * u + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        o[dt] = mma32<F16>(tr_frag128(sV, 32 * u, L.tr[dt]), pf, o[dt], 0, 0, 0);
      }
    }
    {
      const float inv = 1.0f / sum;
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[dt] *= inv;
      if constexpr (F16) {
        if (qrow < N) {
          float* orow = reinterpret_cast<float*>(out) + ((long)b * N + qrow) * D + h * 64;
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(orow + 16 * dt + 4 * g) = o[dt];
          if (g == 0) lse[((long)b * H + h) * N + qrow] = (mx + __builtin_amdgcn_logf(sum)) * LN2;
        }
      } else {
      const u32x4 w0 = pair16(o[0], o[1]), w1 = pair16(o[2], o[3]);
      if (qrow < N) {
        bf16_t* orow = out + ((long)b * N + qrow) * D + h * 64;
        *reinterpret_cast<u32x4*>(orow + pair16_off(0, g)) = w0;
        *reinterpret_cast<u32x4*>(orow + pair16_off(2, g)) = w1;
        if (g == 0) lse[((long)b * H + h) * N + qrow] = (mx + __builtin_amdgcn_logf(sum)) * LN2;
      }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward, two 16-query tiles per wave and pass: every K row fragment and every V^T fragment read from LDS feeds TWO MFMAs
// ------------------------------------------------------------------------------------------------
// attn_fwd_kernel reads one 1 KB fragment from LDS per MFMA.  The LDS delivers 128 B/clk per CU, i.e. a quarter of that per
// SIMD when all four read: 32 cycles per fragment against the 16 cycles its MFMA takes -- by phase ablation the S and P.V
// phases (61 of 101 us) ran at the fragment rate, not the matrix rate.  Here a wave owns a PAIR of query tiles: the same
// four K fragments of a key-tile pair feed eight MFMAs, the same V^T fragment two, so the fragment traffic per FLOP is
// halved (the effect of a 32 x 32 x 16 tile, with the fragment maps of the 16 x 16 x 32 one); 13 query tiles = 7 pairs over
// 4 waves (2, 2, 2, 1: 87 % balanced against 81 % for 4, 3, 3, 3 single tiles).  Scores of both tiles stay in registers
// (2 x NKT x 4 fp32), so this form is for NKT <= 14 (N <= 224); longer sequences keep attn_fwd_kernel.
template <int NKT>
__global__ __launch_bounds__(256, 2) void attn_fwd2_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                           float* __restrict__ lse, int N, int H, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NKT * 16;
  char* sK = smem;
  char* sV = smem + NP * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const LaneOff L = make_lane_off(lane);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long D = (long)H * 64;
  const bf16_t* base = qkv + (long)b * N * 3 * D + h * 64;
  stage_kv_dma(base, D, N, sK, NP, sV, NP, wave, 4, lane);
  const int nqt = (N + 15) >> 4, npair = (nqt + 1) >> 1;
  auto load_q = [&](int qt, int ks) -> u32x4 {           // rows >= N (and the odd tile of the last pair) clamped: never stored
    const int qrow = qt * 16 + (lane & 15);
    return *reinterpret_cast<const u32x4*>(base + (long)(qrow < N ? qrow : N - 1) * 3 * D + 32 * ks + 8 * g);
  };
  const int qp0 = wave < npair ? wave : 0;
  u32x4 qn[2][2] = {{load_q(2 * qp0, 0), load_q(2 * qp0, 1)}, {load_q(2 * qp0 + 1, 0), load_q(2 * qp0 + 1, 1)}};
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's K/V DMA pieces (and its first Q pair) have landed
  __syncthreads();

  for (int qp = wave; qp < npair; qp += 4) {
    bf16x8 qf[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      qf[t][0] = __builtin_bit_cast(bf16x8, qn[t][0]);
      qf[t][1] = __builtin_bit_cast(bf16x8, qn[t][1]);
    }
    if (qp + 4 < npair) {
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        qn[t][0] = load_q(2 * (qp + 4) + t, 0);
        qn[t][1] = load_q(2 * (qp + 4) + t, 1);
      }
    }
    f32x4 st[2][NKT];
    float mx[2] = {-INFINITY, -INFINITY};
#pragma unroll
    for (int kp = 0; kp < NKT; kp += 2) {
      const bf16x8 ka0 = row_frag128(sK, kp * 16, L.rf[0]), kb0 = row_frag128(sK, (kp + 1) * 16, L.rf[0]);
      const bf16x8 ka1 = row_frag128(sK, kp * 16, L.rf[1]), kb1 = row_frag128(sK, (kp + 1) * 16, L.rf[1]);
      f32x4 acc[2][2];
#pragma unroll
      for (int t = 0; t < 2; ++t) acc[t][0] = acc[t][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // four independent two-step chains (query tile t x key tile a | b), interleaved
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka0, qf[t][0], acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb0, qf[t][0], acc[t][1], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        acc[t][0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ka1, qf[t][1], acc[t][0], 0, 0, 0);
        acc[t][1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kb1, qf[t][1], acc[t][1], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = (kp + kk) * 16 + 4 * g + r;
            const float v = key < N ? acc[t][kk][r] * scale_log2e : -INFINITY;
            acc[t][kk][r] = v;
            mx[t] = fmaxf(mx[t], v);
          }
          st[t][kp + kk] = acc[t][kk];
        }
      __builtin_amdgcn_sched_barrier(0);               // keep the K reads of later key tiles below these MFMAs (as attn_fwd_kernel)
    }
    float sum[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 16, 64));
      mx[t] = fmaxf(mx[t], __shfl_xor(mx[t], 32, 64));
      float sm = 0.f;
#pragma unroll
      for (int kt = 0; kt < NKT; ++kt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float p = __builtin_amdgcn_exp2f(st[t][kt][r] - mx[t]);
          st[t][kt][r] = p;
          sm += p;
        }
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      sum[t] = sm;
    }
    f32x4 o[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < NKT / 2; ++u) {
      const bf16x8 pf0 = pack8(st[0][2 * u], st[0][2 * u + 1]), pf1 = pack8(st[1][2 * u], st[1][2 * u + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        const bf16x8 vf = tr_frag128(sV, 32 * u, L.tr[dt]);
        o[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf0, o[0][dt], 0, 0, 0);
        o[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf1, o[1][dt], 0, 0, 0);
      }
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int qrow = (2 * qp + t) * 16 + (lane & 15);
      const float inv = 1.0f / sum[t];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) o[t][dt] *= inv;
      const u32x4 w0 = pair16(o[t][0], o[t][1]), w1 = pair16(o[t][2], o[t][3]);     // executed by every lane (lane exchange)
      if (qrow < N) {
        bf16_t* orow = out + ((long)b * N + qrow) * D + h * 64;
        *reinterpret_cast<u32x4*>(orow + pair16_off(0, g)) = w0;
        *reinterpret_cast<u32x4*>(orow + pair16_off(2, g)) = w1;
        if (g == 0) lse[((long)b * H + h) * N + qrow] = (mx[t] + __builtin_amdgcn_logf(sum[t])) * LN2;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// forward, N <= 208: K and V in 2 x 208 rows = 53 248 B of LDS, THREE workgroups per CU
// ------------------------------------------------------------------------------------------------
// attn_fwd_kernel<14> rounds 197 keys up to 14 tiles (the P.V product walks key-tile PAIRS) = 57 344 B, and 160 KB of LDS then
// hold two workgroups per CU; its 125 VGPRs would allow four.  The kernel's time is load + compute + store per workgroup with
// little overlap inside one (phase ablation: 52 us of memory phases + 61 us of latency-bound compute against 96 us in all), so
// what hides one workgroup's loads is ANOTHER workgroup's compute: residency is the lever.  Here the S phase walks the 13 real
// key tiles (six pairs + one), V sits FIRST in LDS and K behind it: the P.V product's last pair reads "V rows 208..223" out of
// K's first 16 rows -- finite values times p = 0 exactly (keys >= N are masked to -inf before the exponential), the same
// argument as the clamped padding rows.  3 072 workgroups then take 4 rounds of 768 instead of 6 of 512.
template <bool F16 = false>   // F16: half operands, fp32 output (precision "bf16x3")
__global__ __launch_bounds__(256, 3) void attn_fwd13_kernel(const bf16_t* __restrict__ qkv, bf16_t* __restrict__ out,
                                                            float* __restrict__ lse, int N, int H, float scale_log2e) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKS = 13, NP = NKS * 16;
  char* sV = smem;
  char* sK = smem + NP * 128;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const LaneOff L = make_lane_off(lane);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long D = (long)H * 64;
  const bf16_t* base = qkv + (long)b * N * 3 * D + h * 64;
  // K first, then this wave's first Q tile, then V: the first S phase needs K and Q only, so V (25 KB) is still in flight
  // under it.  Every wave issues exactly 7 pieces of each (26 pieces of 8 rows over 4 waves: the last two waves repeat piece
  // 25 -- the same bytes to the same place), so one counted wait serves all waves.
  const int prow = lane >> 3, pch = lane & 7;
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int pc = (wave + 4 * i) < NP / 8 ? (wave + 4 * i) : NP / 8 - 1;
    const int row = 8 * pc + prow;
    const long rr = row < N ? row : N - 1;
    glds16(base + rr * 3 * D + D + (pch ^ (((row >> 1) & 3) << 1)) * 8, sK + pc * 1024);
  }
  const int nqt = (N + 15) >> 4;
  auto load_q = [&](int qt, int ks) -> u32x4 {
    const int qrow = qt * 16 + (lane & 15);
    return *reinterpret_cast<const u32x4*>(base + (long)(qrow < N ? qrow : N - 1) * 3 * D + 32 * ks + 8 * g);
  };
  // the first Q tile through inline asm: a load hipcc can see would be waited for with vmcnt(0) while LDS-DMA is in flight
  // (cdna guide 5, "three .s-level traps" (b)), draining the V pieces this order exists to keep flying
  u32x4 qn0, qn1;
  {
    const int q0 = (wave < nqt ? wave : 0) * 16 + (lane & 15);
    const bf16_t* qp = base + (long)(q0 < N ? q0 : N - 1) * 3 * D + 8 * g;
    asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:64"
                 : "=&v"(qn0), "=&v"(qn1) : "v"(qp) : "memory");
  }
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int pc = (wave + 4 * i) < NP / 8 ? (wave + 4 * i) : NP / 8 - 1;
    const int row = 8 * pc + prow;
    const long rr = row < N ? row : N - 1;
    glds16(base + rr * 3 * D + 2 * D + (pch ^ (((row >> 1) & 3) << 1)) * 8, sV + pc * 1024);
  }
  asm volatile("s_waitcnt vmcnt(7)" : "+v"(qn0), "+v"(qn1) : : "memory");     // K and Q landed; the 7 V pieces may still fly
  __builtin_amdgcn_s_barrier();                        // raw: __syncthreads() would fence with vmcnt(0) and drain them
  asm volatile("" ::: "memory");
  bool v_ready = false;

  for (int qt = wave; qt < nqt; qt += 4) {
    const int qrow = qt * 16 + (lane & 15);
    bf16x8 qf[2];
    qf[0] = __builtin_bit_cast(bf16x8, qn0);
    qf[1] = __builtin_bit_cast(bf16x8, qn1);
    if (qt + 4 < nqt) {
      qn0 = load_q(qt + 4, 0);
      qn1 = load_q(qt + 4, 1);
    }
    f32x4 st[NKS + 1];
    float mx = -INFINITY;
#pragma unroll
    for (int kp = 0; kp < NKS; kp += 2) {
      const bool two = kp + 1 < NKS;                   // compile-time after unrolling
      f32x4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
      const bf16x8 ka0 = row_frag128(sK, kp * 16, L.rf[0]), ka1 = row_frag128(sK, kp * 16, L.rf[1]);
      if (two) {
        const bf16x8 kb0 = row_frag128(sK, (kp + 1) * 16, L.rf[0]), kb1 = row_frag128(sK, (kp + 1) * 16, L.rf[1]);
        acc[0] = mma32<F16>(ka0, qf[0], acc[0], 0, 0, 0);
        acc[1] = mma32<F16>(kb0, qf[0], acc[1], 0, 0, 0);
        acc[0] = mma32<F16>(ka1, qf[1], acc[0], 0, 0, 0);
        acc[1] = mma32<F16>(kb1, qf[1], acc[1], 0, 0, 0);
      } else {
        acc[0] = mma32<F16>(ka0, qf[0], acc[0], 0, 0, 0);
        acc[0] = mma32<F16>(ka1, qf[1], acc[0], 0, 0, 0);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        if (t == 1 && !two) break;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = (kp + t) * 16 + 4 * g + r;
          const float v = key < N ? acc[t][r] * scale_log2e : -INFINITY;
          acc[t][r] = v;
          mx = fmaxf(mx, v);
        }
        st[kp + t] = acc[t];
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int kt = 0; kt < NKS; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float p = __builtin_amdgcn_exp2f(st[kt][r] - mx);
        st[kt][r] = p;
        sum += p;
      }
    st[NKS] = (f32x4){0.f, 0.f, 0.f, 0.f};             // the partner of key tile 12 in the last P.V pair: p = 0 for keys 208..223
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);

    if (!v_ready) {                                    // first pass only (wave-uniform): V must have landed in every wave
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      v_ready = true;
    }
    f32x4 o[4];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int u = 0; u < (NKS + 1) / 2; ++u) {
      const bf16x8 pf = pack8t<F16>(st[2 * u], st[2 * u + 1]);
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
        o[dt] = mma32<F16>(tr_frag128(sV, 32 * u, L.tr[dt]), pf, o[dt], 0, 0, 0);
    }
    const float inv = 1.0f / sum;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] *= inv;
    if constexpr (F16) {                                 // fp32 output: the lane's four features of each 16-feature tile
      if (qrow < N) {
        float* orow = reinterpret_cast<float*>(out) + ((long)b * N + qrow) * D + h * 64;
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) *reinterpret_cast<f32x4*>(orow + 16 * dt + 4 * g) = o[dt];
        if (g == 0) lse[((long)b * H + h) * N + qrow] = (mx + __builtin_amdgcn_logf(sum)) * LN2;
      }
    } else {
    const u32x4 w0 = pair16(o[0], o[1]), w1 = pair16(o[2], o[3]);
    if (qrow < N) {
      bf16_t* orow = out + ((long)b * N + qrow) * D + h * 64;
      *reinterpret_cast<u32x4*>(orow + pair16_off(0, g)) = w0;
      *reinterpret_cast<u32x4*>(orow + pair16_off(2, g)) = w1;
      if (g == 0) lse[((long)b * H + h) * N + qrow] = (mx + __builtin_amdgcn_logf(sum)) * LN2;
    }
    }
  }
  if (!v_ready) {                                      // a wave without a query tile (N < 64) still owes the workgroup its barrier
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
}

// Column sums of dQ / dK / dV of one (image, head) = that head's slice of the to_qkv bias gradient for this image: the
// wave-local sums (16 lanes with equal lane>>4 hold the 16 rows of a tile) go through LDS, and 192 threads write
// colsum[b][which][h][d].  Deterministic: fixed shuffle tree, waves added in order.  Padded rows contribute exact zeros
// (their p is 0), so nothing is masked here.  red: LDS, NW * 192 floats, free at this point.
template <int NW>
__device__ __forceinline__ void attn_colsum_zero(float* red, int tid, int nthreads) {
  for (int i = tid; i < NW * 192; i += nthreads) red[i] = 0.f;
}
__device__ __forceinline__ float rowsum16(float v) {
  v += __shfl_xor(v, 1, 64);
  v += __shfl_xor(v, 2, 64);
  v += __shfl_xor(v, 4, 64);
  v += __shfl_xor(v, 8, 64);
  return v;
}
template <int NW>
__device__ __forceinline__ void attn_colsum_store(const float* red, float* __restrict__ gout, long D, int tid) {
  for (int c = tid; c < 192; c += 64 * NW) {              // (64 NW threads call this)
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) t += red[w * 192 + c];
    gout[(long)(c >> 6) * D + (c & 63)] = t;              // which = c / 64, d = c % 64
  }
}

// ------------------------------------------------------------------------------------------------
// backward: one 512-thread workgroup per (image, head); wave w owns key tiles w, w+8, (w+16)
// ------------------------------------------------------------------------------------------------
template <int NKT, int KPW>  // NKT 16-key tiles (even), KPW key tiles per wave
__global__ __launch_bounds__(512, 1) void attn_bwd_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                          const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                          bf16_t* __restrict__ dqkv, float* __restrict__ colsum, int N, int H,
                                                          float scale) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NP = NKT * 16;
  char* sK = smem;                         // [NP][64] bf16, sw128
  char* sV = sK + NP * 128;
  char* sPair = sV + NP * 128;             // 2 buffers x { Q[32][64], dO[32][64] }, sw128 : 2 x 8 KiB
  char* sDS = sPair + 2 * 8192;            // 2 buffers x [NP][32] bf16 (dS^T), swds
  float* sLse = reinterpret_cast<float*>(sDS + 2 * NP * 64);  // [NP] lse * log2(e), +inf on padding rows
  float* sDelta = sLse + NP;                                  // [NP] rowsum(dO * O)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const LaneOff L = make_lane_off(lane);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long D = (long)H * 64;
  const bf16_t* base = qkv + (long)b * N * 3 * D + h * 64;
  const bf16_t* dobase = dout + (long)b * N * D + h * 64;
  const bf16_t* obase = out + (long)b * N * D + h * 64;
  bf16_t* dbase = dqkv + (long)b * N * 3 * D + h * 64;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const float c2 = scale * LOG2E;

  stage_kv_dma(base, D, N, sK, NP, sV, NP, wave, 8, lane);
  // dS^T rows of key tiles that are entirely padding are never written: keep them finite (0 * K-pad-row = 0)
  for (int idx = tid; idx < 2 * NP * 4; idx += 512) reinterpret_cast<u32x4*>(sDS)[idx] = zero4;
  for (int row = tid; row < NP; row += 512) {
    float dl = 0.f, l2 = INFINITY;
    if (row < N) {
      l2 = lse[((long)b * H + h) * N + row] * LOG2E;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const u32x4 a = *reinterpret_cast<const u32x4*>(dobase + (long)row * D + c * 8);
        const u32x4 o = *reinterpret_cast<const u32x4*>(obase + (long)row * D + c * 8);
        const bf16x8 av = __builtin_bit_cast(bf16x8, a), ov = __builtin_bit_cast(bf16x8, o);
#pragma unroll
        for (int e = 0; e < 8; ++e) dl += (float)av[e] * (float)ov[e];
      }
    }
    sLse[row] = l2;
    sDelta[row] = dl;
  }

  // pair staging: 512 threads = 2 tiles x 32 rows x 8 chunks: thread -> (which = tid>>8, row = (tid>>3)&31, ch = tid&7)
  const int p_which = tid >> 8, p_row = (tid >> 3) & 31, p_ch = tid & 7;
  const bf16_t* pair_src = p_which == 0 ? base + p_ch * 8 : dobase + p_ch * 8;
  const long pair_ld = p_which == 0 ? 3 * D : D;
  auto load_pair = [&](int u) -> u32x4 {
    const int q = 32 * u + p_row;
    // rows >= N clamped, not zeroed (masked through lse = +inf): a select would make the compiler wait for the load at once
    return *reinterpret_cast<const u32x4*>(pair_src + (long)(q < N ? q : N - 1) * pair_ld);
  };
  auto store_pair = [&](int buf, u32x4 v) {
    *reinterpret_cast<u32x4*>(sPair + buf * 8192 + p_which * 4096 + sw128(p_row, p_ch)) = v;
  };

  f32x4 adk[KPW][4], adv[KPW][4];
#pragma unroll
  for (int i = 0; i < KPW; ++i)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      adk[i][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      adv[i][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  constexpr int NQP = NKT / 2;
  f32x4 dqs = {0.f, 0.f, 0.f, 0.f};                      // running column sums of this wave's dQ tile (over query pairs)
  store_pair(0, load_pair(0));
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's K/V DMA pieces have landed
  __syncthreads();

  const int nkt_valid = (N + 15) >> 4;
  for (int u = 0; u < NQP; ++u) {
    const int cur = u & 1;
    u32x4 nxt = zero4;
    if (u + 1 < NQP) nxt = load_pair(u + 1);
    const char* sQ = sPair + cur * 8192;
    const char* sDO = sQ + 4096;
    char* sds = sDS + cur * (NP * 64);

    if (32 * u < N) {
#pragma unroll
      for (int i = 0; i < KPW; ++i) {
        const int kt = wave + 8 * i;
        if (kt >= nkt_valid) continue;
        const int key = kt * 16 + (lane & 15);
        f32x4 s[2], dp[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          dp[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            s[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag128(sQ, t * 16, L.rf[ks]),
                                                           row_frag128(sK, kt * 16, L.rf[ks]), s[t], 0, 0, 0);
            dp[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(row_frag128(sDO, t * 16, L.rf[ks]),
                                                            row_frag128(sV, kt * 16, L.rf[ks]), dp[t], 0, 0, 0);
          }
        }
        // P = exp2(S*c2 - lse2[q]),  dS = P * (dP - delta[q]) * scale ; q = 32u + 16t + 4g + r, key on the lane
        f32x4 pp[2], ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ql = 32 * u + 16 * t + 4 * g + r;
            float p = __builtin_amdgcn_exp2f(s[t][r] * c2 - sLse[ql]);
            p = key < N ? p : 0.f;
            pp[t][r] = p;
            ds[t][r] = p * (dp[t][r] - sDelta[ql]) * scale;
          }
        const bf16x8 pf = pack8(pp[0], pp[1]);
        const bf16x8 dsf = pack8(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          adv[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag128(sDO, 0, L.tr[dt]), pf, adv[i][dt], 0, 0, 0);
          adk[i][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag128(sQ, 0, L.tr[dt]), dsf, adk[i][dt], 0, 0, 0);
        }
        // dS^T -> LDS: row = key, 4 consecutive queries per lane per q-tile
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<bf16x4*>(sds + swds(key, t) + 8 * g) = pack4(ds[t]);
      }
    }
    if (u + 1 < NQP) store_pair(cur ^ 1, nxt);
    __syncthreads();

    // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q] : 2 q-tiles x 4 d-tiles = 8 output tiles, one per wave
    if (32 * u < N) {
      const int t = wave >> 2, dt = wave & 3;
      const int trk = dt == 0 ? L.tr[0] : dt == 1 ? L.tr[1] : dt == 2 ? L.tr[2] : L.tr[3];   // static indices (no scratch)
      const int dsoff = (4 * g + ((lane >> 2) & 3)) * 64 + ((t ^ (g & 1)) << 5) + 8 * (lane & 3);
      f32x4 dq = {0.f, 0.f, 0.f, 0.f};
      for (int v = 0; v < NQP; ++v) {
        if (32 * v >= N) break;
        const char* dsp = sds + 2048 * v + dsoff;               // swds(32 v + 4g + q4, t) + 8 p ; the +16-row block is +1024
        const bf16x8 dsf = cat8(tr_read(dsp), tr_read(dsp + 1024));
        dq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(tr_frag128(sK, 32 * v, trk), dsf, dq, 0, 0, 0);
      }
      const int q = 32 * u + 16 * t + (lane & 15);
      if (q < N) *reinterpret_cast<bf16x4*>(dbase + (long)q * 3 * D + dt * 16 + 4 * g) = pack4(dq);
      dqs += dq;
    }
  }

#pragma unroll
  for (int i = 0; i < KPW; ++i) {
    const int kt = wave + 8 * i;
    const int key = kt * 16 + (lane & 15);
    if (kt < nkt_valid && key < N) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        *reinterpret_cast<bf16x4*>(dbase + (long)key * 3 * D + D + dt * 16 + 4 * g) = pack4(adk[i][dt]);
        *reinterpret_cast<bf16x4*>(dbase + (long)key * 3 * D + 2 * D + dt * 16 + 4 * g) = pack4(adv[i][dt]);
      }
    }
  }
  if (colsum) {                                          // to_qkv bias-gradient partials of this (image, head)
    float* red = reinterpret_cast<float*>(sDS);
    __syncthreads();
    attn_colsum_zero<8>(red, tid, 512);
    __syncthreads();
    const int dtq = wave & 3;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v = rowsum16(dqs[r]);
      if ((lane & 15) == 0) red[wave * 192 + dtq * 16 + 4 * g + r] = v;
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float vk = 0.f, vv = 0.f;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
          vk += adk[i][dt][r];
          vv += adv[i][dt][r];
        }
        vk = rowsum16(vk);
        vv = rowsum16(vv);
        if ((lane & 15) == 0) {
          red[wave * 192 + 64 + dt * 16 + 4 * g + r] = vk;
          red[wave * 192 + 128 + dt * 16 + 4 * g + r] = vv;
        }
      }
    __syncthreads();
    attn_colsum_store<8>(red, colsum + (long)b * 3 * D + h * 64, D, tid);
  }
}

// ------------------------------------------------------------------------------------------------
// backward, N <= 208: one 256-thread workgroup per (image, head), TWO workgroups per CU
// ------------------------------------------------------------------------------------------------
// The 8-wave kernel below needs 104 KB of LDS (one workgroup per CU) and measured 313 us per ViT-B layer: its short
// phases between barriers have nothing to overlap with.  This variant keeps the same mathematics and fragment maps but
// fits two workgroups on a CU: single-buffered Q/dO pair and dS^T (two barriers per query pair instead of one), V
// trimmed to the 13 real key tiles -> 79,616 B.  Wave w owns key tiles w, w+4, w+8, w+12 (128 accumulator VGPRs).
// F16 (precision "bf16x3"): half operands; dout arrives multiplied by a power of two s = *gscale that brings the gradient into
// half's range (mv_attention_bwd_prep_f16), delta_in = s * rowsum(dO . O) comes precomputed (no O loads), and dQ / dK / dV leave as
// fp32 times 1 / s.
// SPLIT = 3 / 6 (F16 only): dQ / dK / dV leave as the bf16 PIECES of the split-operand products (mv_split2_bf16 / mv_split3_bf16 role 0,
// rows of SPLIT * 3 D, segments 3 D apart) instead of fp32 -- the dY operand of to_qkv's dW and dX products, without an fp32 dqkv and
// a split pass over it (8 B per element of traffic per layer); the bias gradient comes from the kernel's own column sums.
template <int NW, bool F16 = false, int SPLIT = 0>   // waves per workgroup: 4 (256 registers per lane) or 2 (one wave per SIMD: 512)
__global__ __launch_bounds__(64 * NW, NW / 2) void attn_bwd4_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                           const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                           bf16_t* __restrict__ dqkv, float* __restrict__ colsum, int N, int H,
                                                           float scale, const float* __restrict__ delta_in = nullptr,
                                                           const float* __restrict__ gscale = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NKT = 13, NPK = 224, NPV = 208, NQP = 7, KPW = (NKT + NW - 1) / NW, NT = 64 * NW;
  char* sK = smem;                           // [224][64] bf16 (rows >= N zero; rows 208..223 exist for the key-pair reads)
  char* sV = sK + NPK * 128;                 // [208][64]
  char* sPair = sV + NPV * 128;              // Q[32][64], dO[32][64]
  char* sDS = sPair + 8192;                  // [224 keys][32 queries] bf16 (dS^T), swds4
  float* sLse = reinterpret_cast<float*>(sDS + NPK * 64);
  float* sDelta = sLse + NPK;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const LaneOff L = make_lane_off(lane);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long D = (long)H * 64;
  const bf16_t* base = qkv + (long)b * N * 3 * D + h * 64;
  const bf16_t* dobase = dout + (long)b * N * D + h * 64;
  const bf16_t* obase = out + (long)b * N * D + h * 64;
  bf16_t* dbase = dqkv + (long)b * N * 3 * D + h * 64;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const float c2 = scale * LOG2E;

  // K, V by DMA (the serialized register path was 112 of this kernel's 313 us: stage_kv_dma)
  stage_kv_dma(base, D, N, sK, NPK, sV, NPV, wave, NW, lane);
  // dS^T rows no wave ever writes must read as zero in the dQ phase.  Two waves (192 < N): the 14 key tiles are all written (the
  // masked keys and wave 1's tile 13 as exact zeros) -- nothing to clear.
  if (NW == 4)
    for (int idx = tid; idx < NPK * 4; idx += NT) reinterpret_cast<u32x4*>(sDS)[idx] = zero4;
  for (int row = tid; row < NPK; row += NT) {
    float l2 = INFINITY;
    if (row < N) l2 = lse[((long)b * H + h) * N + row] * LOG2E;
    sLse[row] = l2;
    if constexpr (F16) sDelta[row] = row < N ? delta_in[((long)b * H + h) * N + row] : 0.f;
  }
  // (one power of two per (image, head): heads and images whose gradient is small get their own range)
  [[maybe_unused]] const float inv_s = F16 ? 1.0f / gscale[blockIdx.x] : 1.0f;
  [[maybe_unused]] const float inv_ss = inv_s * scale;      // F16: dS^T leaves without the softmax scale (3 more bits above half's underflow)
  [[maybe_unused]] float* const dbase32 = reinterpret_cast<float*>(dqkv) + (long)b * N * 3 * D + h * 64;
  // F16 output of four consecutive features at (token row, column col of this head's slice of q | k | v)
  auto put4 = [&](long row, int col, f32x4 v) __attribute__((always_inline)) {
    if constexpr (SPLIT == 0) {
      *reinterpret_cast<f32x4*>(dbase32 + row * 3 * D + col) = v;
    } else {
      bf16_t* o = dqkv + ((long)b * N + row) * (SPLIT * 3 * D) + h * 64 + col;
      bf16x4 p[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bf16_t p0 = (bf16_t)v[e];
        const float r1 = v[e] - (float)p0;
        const bf16_t p1 = (bf16_t)r1;
        p[0][e] = p0;
        p[1][e] = p1;
        p[2][e] = (bf16_t)(r1 - (float)p1);
      }
      constexpr int order[6] = {0, 0, 1, 0, 1, 2};
#pragma unroll
      for (int sg = 0; sg < SPLIT; ++sg) *reinterpret_cast<bf16x4*>(o + (long)sg * 3 * D) = p[order[sg]];
    }
  };

  // pair staging: 2 tiles x 32 rows x 8 chunks = 512 chunks, PE = 512 / NT per thread (the first half Q, the second dO)
  constexpr int PE = 512 / NT;
  auto load_pair = [&](int u, int e) -> u32x4 {
    const int idx = tid + NT * e;
    const int which = idx >> 8, row = (idx >> 3) & 31, ch = idx & 7;
    const int q = 32 * u + row;
    const bf16_t* src = which == 0 ? base + ch * 8 : dobase + ch * 8;
    const long ld = which == 0 ? 3 * D : D;
    // rows >= N: clamped, not zeroed (masked through lse = +inf, see above) -- a select here forces the compiler to wait
    // for the load at once, which turned this prefetch into a stall
    return *reinterpret_cast<const u32x4*>(src + (long)(q < N ? q : N - 1) * ld);
  };
  auto store_pair = [&](int e, u32x4 v) {
    const int idx = tid + NT * e;
    const int which = idx >> 8, row = (idx >> 3) & 31, ch = idx & 7;
    *reinterpret_cast<u32x4*>(sPair + which * 4096 + sw128(row, ch)) = v;
  };
  // delta[q] = sum_d dO[q][d] O[q][d] rides on the pair prefetch: thread (row = tid >> 3, chunk = tid & 7) already holds its
  // 8 elements of dO (load_pair(u, 1)), loads the same 8 of O, and the row's 8 threads (consecutive lanes) add up.  (Computed
  // for all 208 rows before the loop, this cost 53 of the kernel's 282 us: 16 dependent-latency loads per thread with
  // nothing to hide under.)
  // (the thread's dO chunk e = PE / 2 + j is row (tid >> 3) + (NT / 8) j, chunk tid & 7: the same row and chunk of O)
  auto load_o = [&](int u, int j) -> u32x4 {
    const int q = 32 * u + (tid >> 3) + (NT / 8) * j;
    return *reinterpret_cast<const u32x4*>(obase + (long)(q < N ? q : N - 1) * D + (tid & 7) * 8);
  };
  auto put_delta = [&](int u, int j, u32x4 dov, u32x4 ov) {
    const bf16x8 av = __builtin_bit_cast(bf16x8, dov), bv = __builtin_bit_cast(bf16x8, ov);
    float dl = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) dl += (float)av[e] * (float)bv[e];
    dl += __shfl_xor(dl, 1);
    dl += __shfl_xor(dl, 2);
    dl += __shfl_xor(dl, 4);
    if ((tid & 7) == 0) sDelta[32 * u + (tid >> 3) + (NT / 8) * j] = dl;
  };

  f32x4 adk[KPW][4], adv[KPW][4];
#pragma unroll
  for (int i = 0; i < KPW; ++i)
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      adk[i][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      adv[i][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

  {
    u32x4 x0[PE], o0[PE / 2];
#pragma unroll
    for (int e = 0; e < PE; ++e) x0[e] = load_pair(0, e);
#pragma unroll
    for (int j = 0; j < PE / 2; ++j) o0[j] = F16 ? zero4 : load_o(0, j);
#pragma unroll
    for (int e = 0; e < PE; ++e) store_pair(e, x0[e]);
#pragma unroll
    for (int j = 0; j < PE / 2; ++j)
      if constexpr (!F16) put_delta(0, j, x0[PE / 2 + j], o0[j]);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // this wave's K/V DMA pieces have landed
  __syncthreads();

  const char* sQ = sPair;
  const char* sDO = sPair + 4096;
  const int nkt_valid = (N + 15) >> 4;
  f32x4 dqs0 = {0.f, 0.f, 0.f, 0.f}, dqs1 = {0.f, 0.f, 0.f, 0.f};   // running column sums of this wave's two dQ tiles
  for (int u = 0; u < NQP; ++u) {
    u32x4 nx[PE], nxo[PE / 2];
#pragma unroll
    for (int e = 0; e < PE; ++e) nx[e] = zero4;
#pragma unroll
    for (int j = 0; j < PE / 2; ++j) nxo[j] = zero4;
    if (u + 1 < NQP) {
#pragma unroll
      for (int e = 0; e < PE; ++e) nx[e] = load_pair(u + 1, e);
#pragma unroll
      for (int j = 0; j < PE / 2; ++j) nxo[j] = F16 ? zero4 : load_o(u + 1, j);
    }
    if constexpr (NW == 2) {
      if (32 * u < N) {
        // Two waves of 512 registers: EVERY fragment of the query pair -- rows of Q and dO (S, dP), their transposes (dK, dV) --
        // and the pair's lse / delta are read once per pair and wave, 16 KB, and only the K / V fragments (4 KB) per key tile;
        // the four-wave form below reads 13 KB per key tile, 8 of them the pair's row fragments again (it has no registers
        // to keep them in), and its S phase is bound by LDS bandwidth and by the length of its dependent chain.
        bf16x8 dotr[4], qtr[4], qrow[2][2], dorow[2][2];
        f32x4 lse4[2], dl4[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          lse4[t] = *reinterpret_cast<const f32x4*>(sLse + 32 * u + 16 * t + 4 * g);
          dl4[t] = -*reinterpret_cast<const f32x4*>(sDelta + 32 * u + 16 * t + 4 * g);       // (negated: see stage A)
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            qrow[t][ks] = row_frag128(sQ, t * 16, L.rf[ks]);
            dorow[t][ks] = row_frag128(sDO, t * 16, L.rf[ks]);
          }
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          dotr[dt] = tr_frag128(sDO, 0, L.tr[dt]);
          qtr[dt] = tr_frag128(sQ, 0, L.tr[dt]);
        }
        // One wave per SIMD is bound by what it can ISSUE: a 16x16x32 MFMA holds the vector issue port for 8 of its 16 cycles, a
        // plain VALU operation costs 4, a transcendental 8 (MI355X_MICROARCH.md constants) -- so the 16 MFMAs of a key tile hide
        // 128 cycles of VALU and everything beyond that is added time.  B is therefore cut to the minimum -- per element one fma
        // and one exp2 (P), one multiply (dS: the MFMA accumulator of dP starts at -delta, and the softmax scale moves to the dK
        // and dQ stores), half a bf16 pack each for P and dS -- and the file is compiled without SLP vectorisation (a packed
        // f32 operation beside MFMAs costs more than the two scalar ones it replaces, same table).
        // Software pipeline over the wave's seven key tiles (one wave per SIMD: nothing else hides a dependent chain).  Per tile
        //   A: S = Q K^T, dP = dO V^T (8 MFMAs)   B: P, dS (exp and ~60 more VALU operations)   C: dV^T += dO^T P, dK^T += Q^T dS (8)
        // and step i issues C(i-1) and A(i+1) -- sixteen MFMAs that depend on nothing in flight -- between the VALU operations
        // of B(i); in source order each tile is A, B, C and the matrix unit idles through every B.  Both waves run seven tiles:
        // wave 1's seventh is key tile 13 (K rows 208..223 are the padding rows, "V" rows the first of the pair buffer: finite
        // bits), every key of it is masked, it accumulates and stores exact zeros and is never written out.
        f32x4 sv[2][2], dpv[2][2];
        bf16x8 kf[2][2], vf[2][2], pf[2], dsf[2];
        auto read_kv = [&](int i) __attribute__((always_inline)) {
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            kf[i & 1][ks] = row_frag128(sK, (wave + NW * i) * 16, L.rf[ks]);
            vf[i & 1][ks] = row_frag128(sV, (wave + NW * i) * 16, L.rf[ks]);
          }
        };
        auto stage_a = [&](int i) __attribute__((always_inline)) {
#pragma unroll
          for (int t = 0; t < 2; ++t) {
            sv[i & 1][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
            dpv[i & 1][t] = dl4[t];                       // the accumulator starts at -delta[q]: dP - delta leaves the MFMA
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
              sv[i & 1][t] = mma32<F16>(qrow[t][ks], kf[i & 1][ks], sv[i & 1][t], 0, 0, 0);
              dpv[i & 1][t] = mma32<F16>(dorow[t][ks], vf[i & 1][ks], dpv[i & 1][t], 0, 0, 0);
            }
          }
        };
        auto stage_b = [&](int i) __attribute__((always_inline)) {
          const int key = (wave + NW * i) * 16 + (lane & 15);
          f32x4 pp[2], ds[2];
#pragma unroll
          for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
              float pv = __builtin_amdgcn_exp2f(fmaf(sv[i & 1][t][r], c2, -lse4[t][r]));
              if (i == KPW - 1) pv = key < N ? pv : 0.f;           // (192 < N: only key tile 12 holds padded keys; 13 all)
              pp[t][r] = pv;
              ds[t][r] = pv * dpv[i & 1][t][r];                    // dS / scale: dK and dQ take the factor at their stores
            }
          pf[i & 1] = pack8t<F16>(pp[0], pp[1]);
          const bf16x4 d0 = pack4t<F16>(ds[0]), d1 = pack4t<F16>(ds[1]);
          dsf[i & 1] = cat8(d0, d1);
          *reinterpret_cast<bf16x4*>(sDS + swds4(key, 0, g)) = d0;
          *reinterpret_cast<bf16x4*>(sDS + swds4(key, 1, g)) = d1;
        };
        auto stage_c = [&](int i) __attribute__((always_inline)) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            adv[i][dt] = mma32<F16>(dotr[dt], pf[i & 1], adv[i][dt], 0, 0, 0);
            adk[i][dt] = mma32<F16>(qtr[dt], dsf[i & 1], adk[i][dt], 0, 0, 0);
          }
        };
        read_kv(0);
        stage_a(0);
        read_kv(1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
          // this step's instructions: [C(i-1): 8 MFMA] [A(i+1): 8 MFMA] [reads of K/V(i+2)] [B(i): VALU, 2 LDS stores]
          if (i >= 1) stage_c(i - 1);
          if (i + 1 < KPW) stage_a(i + 1);
          stage_b(i);
          if (i + 2 < KPW) read_kv(i + 2);
          // and the order they are to be issued in: one MFMA, four VALU, ...; the LDS traffic at the end
          const int nm = (i >= 1 ? 8 : 0) + (i + 1 < KPW ? 8 : 0);
#pragma unroll
          for (int k = 0; k < nm; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
            __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);      // VALU
          }
          __builtin_amdgcn_sched_group_barrier(0x002, 64, 0);       // whatever VALU is left
          __builtin_amdgcn_sched_group_barrier(0x200, 2, 0);        // the two dS^T stores
          __builtin_amdgcn_sched_group_barrier(0x100, 4, 0);        // K/V fragment reads of tile i + 2
          __builtin_amdgcn_sched_barrier(0);
        }
        stage_c(KPW - 1);
      }
    } else if (32 * u < N) {
      // The query pair's TRANSPOSED fragments (operands of the dV / dK products) do not depend on the key tile: read them
      // once per pair (8 fragments, 32 VGPRs) instead of once per key tile.  (Hoisting the row fragments as well spills.)
      bf16x8 dotr[4], qtr[4];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dotr[dt] = tr_frag128(sDO, 0, L.tr[dt]);
        qtr[dt] = tr_frag128(sQ, 0, L.tr[dt]);
      }
#pragma unroll
      for (int i = 0; i < KPW; ++i) {
        const int kt = wave + NW * i;
        if (kt >= nkt_valid || kt >= NKT) continue;
        const int key = kt * 16 + (lane & 15);
        f32x4 s[2], dp[2];
        bf16x8 kf[2], vf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          kf[ks] = row_frag128(sK, kt * 16, L.rf[ks]);
          vf[ks] = row_frag128(sV, kt * 16, L.rf[ks]);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
          dp[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            s[t] = mma32<F16>(row_frag128(sQ, t * 16, L.rf[ks]), kf[ks], s[t], 0, 0, 0);
            dp[t] = mma32<F16>(row_frag128(sDO, t * 16, L.rf[ks]), vf[ks], dp[t], 0, 0, 0);
          }
        }
        f32x4 pp[2], ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int ql = 32 * u + 16 * t + 4 * g + r;
            float p = __builtin_amdgcn_exp2f(s[t][r] * c2 - sLse[ql]);
            p = key < N ? p : 0.f;
            pp[t][r] = p;
            if constexpr (F16) ds[t][r] = __builtin_amdgcn_fmed3f(p * (dp[t][r] - sDelta[ql]), -65000.f, 65000.f);
            else ds[t][r] = p * (dp[t][r] - sDelta[ql]) * scale;
          }
        const bf16x8 pf = pack8t<F16>(pp[0], pp[1]);
        const bf16x8 dsf = pack8t<F16>(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          adv[i][dt] = mma32<F16>(dotr[dt], pf, adv[i][dt], 0, 0, 0);
          adk[i][dt] = mma32<F16>(qtr[dt], dsf, adk[i][dt], 0, 0, 0);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) *reinterpret_cast<bf16x4*>(sDS + swds4(key, t, g)) = pack4t<F16>(ds[t]);
      }
    }
    __syncthreads();                                   // S-phase done everywhere: dS^T complete, Q/dO pair no longer read
    if (u + 1 < NQP) {
#pragma unroll
      for (int e = 0; e < PE; ++e) store_pair(e, nx[e]);
#pragma unroll
      for (int j = 0; j < PE / 2; ++j)
        if constexpr (!F16) put_delta(u + 1, j, nx[PE / 2 + j], nxo[j]);   // read by the next S-phase, after the barrier below
    }
    // dQ^T[d][q] = sum_key K^T[d][key] dS^T[key][q]: 2 q-tiles x 4 d-tiles = 8 output tiles.  Four waves: two each (one q-tile,
    // a d-tile pair); two waves: four each (both q-tiles of d-tile pair `wave`: the K^T fragments serve both)
    if (32 * u < N) {
      constexpr int QT = NW == 4 ? 1 : 2;
      const int t0 = NW == 4 ? wave >> 1 : 0, dh = NW == 4 ? (wave & 1) : wave;
      const int trk0 = dh ? L.tr[2] : L.tr[0], trk1 = dh ? L.tr[3] : L.tr[1];
      int dsoff[QT];                                  // rows 32 v + 4 g + q: (row >> 1) & 7 is the same for every v
#pragma unroll
      for (int tq = 0; tq < QT; ++tq) dsoff[tq] = swds4(4 * g + ((lane >> 2) & 3), t0 + tq, lane & 3);
      f32x4 dq[QT][2];
#pragma unroll
      for (int tq = 0; tq < QT; ++tq) dq[tq][0] = dq[tq][1] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // All seven 32-key groups, unconditionally (groups beyond N hold zero dS^T rows -- zero-filled at the start, never
      // written -- against clamped finite K rows), two register sets: the six transposed reads of group v + 1 are in flight
      // while group v's two MFMAs run.  (As a rolled loop with a break each group was read -> wait -> MFMA: 16 % MFMA duty.)
      // ring depth: two waves have the registers (the S phase's fragments are dead here) to request ALL seven groups at once --
      // one wave per SIMD hides an LDS round trip only behind its own MFMAs, and two groups in flight left it waiting
      constexpr int RD = NW == 4 ? 2 : NQP;
      bf16x8 dsv[RD][QT], kv0[RD], kv1[RD];
#define ATTN_DQ_READ(buf_, v_)                                                  \
  {                                                                             \
    _Pragma("unroll") for (int tq = 0; tq < QT; ++tq) {                         \
      const char* dsp_ = sDS + 2048 * (v_) + dsoff[tq];                         \
      dsv[buf_][tq] = cat8(tr_read(dsp_), tr_read(dsp_ + 1024));                \
    }                                                                           \
    kv0[buf_] = tr_frag128(sK, 32 * (v_), trk0);                                \
    kv1[buf_] = tr_frag128(sK, 32 * (v_), trk1);                                \
  }
#pragma unroll
      for (int v = 0; v < RD - 1; ++v) ATTN_DQ_READ(v, v)
#pragma unroll
      for (int v = 0; v < NQP; ++v) {
        if (v + RD - 1 < NQP) ATTN_DQ_READ((v + RD - 1) % RD, v + RD - 1)
#pragma unroll
        for (int tq = 0; tq < QT; ++tq) {
          dq[tq][0] = mma32<F16>(kv0[v % RD], dsv[v % RD][tq], dq[tq][0], 0, 0, 0);
          dq[tq][1] = mma32<F16>(kv1[v % RD], dsv[v % RD][tq], dq[tq][1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
#undef ATTN_DQ_READ
#pragma unroll
      for (int tq = 0; tq < QT; ++tq) {
        const int q = 32 * u + 16 * (t0 + tq) + (lane & 15);
        if constexpr (NW == 2) {                             // dS^T came without the softmax scale
          dq[tq][0] *= scale;
          dq[tq][1] *= scale;
        }
        if constexpr (F16) {
          if (q < N) {
            put4(q, 16 * (2 * dh) + 4 * g, dq[tq][0] * inv_ss);
            put4(q, 16 * (2 * dh + 1) + 4 * g, dq[tq][1] * inv_ss);
          }
        } else {
        const u32x4 dqw = pair16(dq[tq][0], dq[tq][1]);      // (every lane executes the exchange)
        if (q < N) *reinterpret_cast<u32x4*>(dbase + (long)q * 3 * D + pair16_off(2 * dh, g)) = dqw;
        }
        dqs0 += dq[tq][0];
        dqs1 += dq[tq][1];
      }
    }
    __syncthreads();                                   // dS^T reads done; next Q/dO pair visible
  }

  if constexpr (NW == 2) {                                   // dS^T came without the softmax scale
#pragma unroll
    for (int i = 0; i < KPW; ++i)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) adk[i][dt] *= scale;
  }
#pragma unroll
  for (int i = 0; i < KPW; ++i) {
    const int kt = wave + NW * i;
    const int key = kt * 16 + (lane & 15);
    if (kt < nkt_valid && kt < NKT) {                     // wave-uniform: the lane exchange below runs on whole waves
      if constexpr (F16) {
        if (key < N) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            put4(key, (int)D + 16 * dt + 4 * g, adk[i][dt] * inv_ss);
            put4(key, 2 * (int)D + 16 * dt + 4 * g, adv[i][dt] * inv_s);
          }
        }
      } else {
#pragma unroll
      for (int dp = 0; dp < 4; dp += 2) {
        const u32x4 wk = pair16(adk[i][dp], adk[i][dp + 1]), wv = pair16(adv[i][dp], adv[i][dp + 1]);
        if (key < N) {
          *reinterpret_cast<u32x4*>(dbase + (long)key * 3 * D + D + pair16_off(dp, g)) = wk;
          *reinterpret_cast<u32x4*>(dbase + (long)key * 3 * D + 2 * D + pair16_off(dp, g)) = wv;
        }
      }
      }
    }
  }
  if (colsum) {                                          // to_qkv bias-gradient partials of this (image, head)
    float* red = reinterpret_cast<float*>(sDS);          // the loop's last barrier freed dS^T
    attn_colsum_zero<NW>(red, tid, NT);
    __syncthreads();
    const int dq_d0 = 32 * (NW == 4 ? (wave & 1) : wave);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float v0 = rowsum16(dqs0[r]) * (F16 ? inv_ss : 1.0f), v1 = rowsum16(dqs1[r]) * (F16 ? inv_ss : 1.0f);
      if ((lane & 15) == 0) {
        red[wave * 192 + dq_d0 + 4 * g + r] = v0;
        red[wave * 192 + dq_d0 + 16 + 4 * g + r] = v1;
      }
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float vk = 0.f, vv = 0.f;
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
          vk += adk[i][dt][r];
          vv += adv[i][dt][r];
        }
        vk = rowsum16(vk) * (F16 ? inv_ss : 1.0f);
        vv = rowsum16(vv) * (F16 ? inv_s : 1.0f);
        if ((lane & 15) == 0) {
          red[wave * 192 + 64 + dt * 16 + 4 * g + r] = vk;
          red[wave * 192 + 128 + dt * 16 + 4 * g + r] = vv;
        }
      }
    __syncthreads();
    attn_colsum_store<NW>(red, colsum + (long)b * 3 * D + h * 64, D, tid);
  }
}

// ------------------------------------------------------------------------------------------------
// backward, two passes without barriers inside: one 256-thread workgroup per (image, head), two per CU
// ------------------------------------------------------------------------------------------------
// attn_bwd4_kernel exchanges dS through LDS for dQ: two barriers per 32 queries, and between them four serial
// load -> MFMA -> exp -> MFMA chains per wave (the key tiles sit behind execmask branches): 18 % MFMA duty in its loop
// (phase ablation: S-phase 90 us, dQ phase 48 us, of 259).  Here the work is cut so that no wave ever needs another wave's
// data inside a loop, at the price of computing S and dP twice (7 products instead of 5):
//   pass A (dQ):    K, V in LDS.  A wave owns 16 QUERIES: S^T = K Q^T and dP^T = V dO^T (keys on accumulator rows, as in
//                   the forward kernel, Q / dO / O fragments straight from global memory one tile ahead), P^T from the saved
//                   lse, delta = rowsum(dO * O) from the fragments in registers, dS^T is at once the B operand of
//                   dQ^T = K^T dS^T.  Streams over the keys: 12 MFMAs per 32 keys, 16 accumulator registers.
//   pass B (dK/dV): the SAME LDS bytes re-staged with Q and dO (one barrier pair per workgroup).  A wave owns 32 KEYS (K, V
//                   fragments in registers, from global memory) and walks the queries: S, dP (keys on lanes), P / dS are the
//                   B operands of dV^T += dO^T P and dK^T += Q^T dS -- the S-phase of attn_bwd4_kernel without its dS store.
// LDS: 2 x [32 NP32 rows][64] bf16 + lse + delta + column sums = 62,208 B (N <= 224) / 79,104 B (N <= 288): two workgroups per CU for the
// 257-token case as well, which the 104 KB eight-wave kernel above could not do.
__device__ __forceinline__ void stage_rows_dma(const bf16_t* src, long ld, int N, char* dst, int rows, int wave, int nwaves,
                                               int lane) {
  const int prow = lane >> 3, pch = lane & 7;
  for (int pc = wave; pc < rows / 8; pc += nwaves) {
    const int row = 8 * pc + prow;
    const long rr = row < N ? row : N - 1;
    const int ch = pch ^ (((row >> 1) & 3) << 1);
    glds16(src + rr * ld + ch * 8, dst + pc * 1024);
  }
}

// F16 / SPLIT: as attn_bwd4_kernel -- half operands, dout scaled per (image, head) by gscale, delta precomputed (no O loads),
// outputs fp32 (SPLIT = 0) or the bf16 pieces of the split-operand products (SPLIT = 3 / 6) with the scale divided out
template <int NP32, bool F16 = false, int SPLIT = 0>
__global__ __launch_bounds__(256, 2) void attn_bwd2p_kernel(const bf16_t* __restrict__ qkv, const bf16_t* __restrict__ out,
                                                            const bf16_t* __restrict__ dout, const float* __restrict__ lse,
                                                            bf16_t* __restrict__ dqkv, float* __restrict__ colsum, int N, int H,
                                                            float scale, const float* __restrict__ delta_in = nullptr,
                                                            const float* __restrict__ gscale = nullptr) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NR = NP32 * 32;
  char* sA = smem;                           // pass A: K    pass B: Q      ([NR][64] bf16, sw128 image, rows >= N clamped)
  char* sB = smem + NR * 128;                // pass A: V    pass B: dO
  float* sLse = reinterpret_cast<float*>(sB + NR * 128);   // lse * log2(e); +inf for rows >= N (their p is exactly 0)
  float* sDelta = sLse + NR;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4;
  const LaneOff L = make_lane_off(lane);
  const int b = blockIdx.x / H, h = blockIdx.x % H;
  const long D = (long)H * 64;
  const bf16_t* base = qkv + (long)b * N * 3 * D + h * 64;
  const bf16_t* dobase = dout + (long)b * N * D + h * 64;
  const bf16_t* obase = out + (long)b * N * D + h * 64;
  bf16_t* dbase = dqkv + (long)b * N * 3 * D + h * 64;
  const float c2 = scale * LOG2E;
  [[maybe_unused]] const float inv_s = F16 ? 1.0f / gscale[blockIdx.x] : 1.0f;
  [[maybe_unused]] const float inv_ss = inv_s * scale;      // F16: dS leaves without the softmax scale
  auto put4 = [&](long row, int col, f32x4 v) __attribute__((always_inline)) {     // F16 outputs: four consecutive features
    if constexpr (SPLIT == 0) {
      *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(dqkv) + ((long)b * N + row) * 3 * D + h * 64 + col) = v;
    } else {
      bf16_t* o = dqkv + ((long)b * N + row) * (SPLIT * 3 * D) + h * 64 + col;
      bf16x4 p[3];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bf16_t p0 = (bf16_t)v[e];
        const float r1 = v[e] - (float)p0;
        const bf16_t p1 = (bf16_t)r1;
        p[0][e] = p0;
        p[1][e] = p1;
        p[2][e] = (bf16_t)(r1 - (float)p1);
      }
      constexpr int order[6] = {0, 0, 1, 0, 1, 2};
#pragma unroll
      for (int sg = 0; sg < SPLIT; ++sg) *reinterpret_cast<bf16x4*>(o + (long)sg * 3 * D) = p[order[sg]];
    }
  };

  stage_rows_dma(base + D, 3 * D, N, sA, NR, wave, 4, lane);
  stage_rows_dma(base + 2 * D, 3 * D, N, sB, NR, wave, 4, lane);
  for (int row = tid; row < NR; row += 256) {
    sLse[row] = row < N ? lse[((long)b * H + h) * N + row] * LOG2E : INFINITY;
    sDelta[row] = 0.f;
  }
  // fragments of a 16-row tile from global memory: B operands (n = row lane&15, 8 consecutive d at 32 ks + 8 g)
  auto frag = [&](const bf16_t* src, long ld, int row, int ks) -> u32x4 {
    return *reinterpret_cast<const u32x4*>(src + (long)(row < N ? row : N - 1) * ld + 32 * ks + 8 * g);
  };
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---------------- pass A: dQ ----------------
  // a task = 32 queries (two 16-query tiles share every K / V / K^T fragment read: 24 MFMAs per 16 LDS reads -- with one
  // tile per task this pass was bound by LDS bandwidth)
  f32x4 dqs[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) dqs[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (int task = wave; task < NP32; task += 4) {
    if (32 * task >= N) break;
    bf16x8 qf[2][2], dof[2][2];
    float dl[2], l2[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int qrow = 32 * task + 16 * t + (lane & 15);
      float d = 0.f;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        qf[t][ks] = __builtin_bit_cast(bf16x8, frag(base, 3 * D, qrow, ks));
        dof[t][ks] = __builtin_bit_cast(bf16x8, frag(dobase, D, qrow, ks));
        if constexpr (!F16) {
          const bf16x8 of = __builtin_bit_cast(bf16x8, frag(obase, D, qrow, ks));
#pragma unroll
          for (int e = 0; e < 8; ++e) d += (float)dof[t][ks][e] * (float)of[e];
        }
      }
      if constexpr (F16) {
        d = delta_in[((long)b * H + h) * N + (qrow < N ? qrow : N - 1)];
      } else {
        d += __shfl_xor(d, 16, 64);
        d += __shfl_xor(d, 32, 64);
      }
      if (g == 0) sDelta[qrow] = d;                      // pass B reads it after the barrier between the passes
      dl[t] = d;
      l2[t] = sLse[qrow];
    }
    f32x4 dq[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dq[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    // the row fragments of key pair u + 1 are read while pair u computes (two register sets); K^T of pair u is read at the
    // top of its own iteration (first needed after the S / dP products and the exponentials).  The scheduling barrier
    // keeps the compiler from hoisting every pair's reads to the top, which spilled.
    bf16x8 kr[2][2][2], vr[2][2][2], ktr[4];
#define ATTN_A_READ(buf_, u_)                                                              \
  {                                                                                        \
    _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) _Pragma("unroll") for (int ks = 0; ks < 2; ++ks) { \
      kr[buf_][kk][ks] = row_frag128(sA, (2 * (u_) + kk) * 16, L.rf[ks]);                  \
      vr[buf_][kk][ks] = row_frag128(sB, (2 * (u_) + kk) * 16, L.rf[ks]);                  \
    }                                                                                      \
  }
    ATTN_A_READ(0, 0)
#pragma unroll
    for (int u = 0; u < NP32; ++u) {
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) ktr[dt] = tr_frag128(sA, 32 * u, L.tr[dt]);
      if (u + 1 < NP32) ATTN_A_READ((u + 1) & 1, u + 1)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        f32x4 ds[2];
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
          f32x4 st = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            st = mma32<F16>(kr[u & 1][kk][ks], qf[t][ks], st, 0, 0, 0);
            dp = mma32<F16>(vr[u & 1][kk][ks], dof[t][ks], dp, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const int key = (2 * u + kk) * 16 + 4 * g + r;
            const float p = key < N ? __builtin_amdgcn_exp2f(st[r] * c2 - l2[t]) : 0.f;
            ds[kk][r] = F16 ? __builtin_amdgcn_fmed3f(p * (dp[r] - dl[t]), -65000.f, 65000.f) : p * (dp[r] - dl[t]) * scale;
          }
        }
        const bf16x8 dsf = pack8t<F16>(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt)
          dq[t][dt] = mma32<F16>(ktr[dt], dsf, dq[t][dt], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#undef ATTN_A_READ
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int qrow = 32 * task + 16 * t + (lane & 15);
      if constexpr (F16) {
        if (qrow < N) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) put4(qrow, 16 * dt + 4 * g, dq[t][dt] * inv_ss);
        }
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) dqs[dt] += dq[t][dt] * inv_ss;
      } else {
      const u32x4 w0 = pair16(dq[t][0], dq[t][1]), w1 = pair16(dq[t][2], dq[t][3]);
      if (qrow < N) {
        bf16_t* drow = dbase + (long)qrow * 3 * D;
        *reinterpret_cast<u32x4*>(drow + pair16_off(0, g)) = w0;
        *reinterpret_cast<u32x4*>(drow + pair16_off(2, g)) = w1;
      }
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) dqs[dt] += dq[t][dt];   // padded queries: lse = +inf -> p = 0 -> exact zeros
      }
    }
  }

  // column sums of this wave's dQ tiles -> LDS now (16 registers that pass B needs)
  float* sCs = sDelta + NR;                    // [4 waves][192]: dq | dk | dv column sums of this (image, head)
  if (colsum) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float vq = rowsum16(dqs[dt][r]);
        if ((lane & 15) == 0) sCs[wave * 192 + dt * 16 + 4 * g + r] = vq;
      }
  }
  // ---------------- the same LDS bytes now hold Q and dO ----------------
  __syncthreads();
  stage_rows_dma(base, 3 * D, N, sA, NR, wave, 4, lane);
  stage_rows_dma(dobase, D, N, sB, NR, wave, 4, lane);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  // ---------------- pass B: dK, dV ----------------
  if (colsum && (lane & 15) == 0) {
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        sCs[wave * 192 + 64 + dt * 16 + 4 * g + r] = 0.f;
        sCs[wave * 192 + 128 + dt * 16 + 4 * g + r] = 0.f;
      }
  }
  const int nkt_valid = (N + 15) >> 4;
  for (int task = wave; task < NP32; task += 4) {
    if (32 * task >= N) break;
    bf16x8 kf[2][2], vf[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int krow = (2 * task + i) * 16 + (lane & 15);
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) {
        kf[i][ks] = __builtin_bit_cast(bf16x8, frag(base + D, 3 * D, krow, ks));
        vf[i][ks] = __builtin_bit_cast(bf16x8, frag(base + 2 * D, 3 * D, krow, ks));
      }
    }
    f32x4 adk[2][4], adv[2][4];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        adk[i][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
        adv[i][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    // query groups beyond N are computed too (their rows are clamped copies and lse = +inf: p = 0 exactly): NP32 is chosen so
    // that at the production lengths (197, 257) every group is real
#pragma unroll 1
    for (int u = 0; u < NP32; ++u) {
      bf16x8 dotr[4], qtr[4], qrf[2][2], dorf[2][2];
#pragma unroll
      for (int dt = 0; dt < 4; ++dt) {
        dotr[dt] = tr_frag128(sB, 32 * u, L.tr[dt]);
        qtr[dt] = tr_frag128(sA, 32 * u, L.tr[dt]);
      }
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
          qrf[t][ks] = row_frag128(sA, 32 * u + 16 * t, L.rf[ks]);
          dorf[t][ks] = row_frag128(sB, 32 * u + 16 * t, L.rf[ks]);
        }
      f32x4 l2v[2], dlv[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        l2v[t] = *reinterpret_cast<const f32x4*>(sLse + 32 * u + 16 * t + 4 * g);
        dlv[t] = *reinterpret_cast<const f32x4*>(sDelta + 32 * u + 16 * t + 4 * g);
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int key = (2 * task + i) * 16 + (lane & 15);
        f32x4 pp[2], ds[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          f32x4 sv = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
          for (int ks = 0; ks < 2; ++ks) {
            sv = mma32<F16>(qrf[t][ks], kf[i][ks], sv, 0, 0, 0);
            dp = mma32<F16>(dorf[t][ks], vf[i][ks], dp, 0, 0, 0);
          }
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float p = __builtin_amdgcn_exp2f(sv[r] * c2 - l2v[t][r]);
            p = key < N ? p : 0.f;
            pp[t][r] = p;
            ds[t][r] = F16 ? __builtin_amdgcn_fmed3f(p * (dp[r] - dlv[t][r]), -65000.f, 65000.f) : p * (dp[r] - dlv[t][r]) * scale;
          }
        }
        const bf16x8 pf = pack8t<F16>(pp[0], pp[1]);
        const bf16x8 dsf = pack8t<F16>(ds[0], ds[1]);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          adv[i][dt] = mma32<F16>(dotr[dt], pf, adv[i][dt], 0, 0, 0);
          adk[i][dt] = mma32<F16>(qtr[dt], dsf, adk[i][dt], 0, 0, 0);
        }
      }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int kt = 2 * task + i;
      const int key = kt * 16 + (lane & 15);
      if constexpr (F16) {
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
          adk[i][dt] *= inv_ss;
          adv[i][dt] *= inv_s;
        }
        if (kt < nkt_valid && key < N) {
#pragma unroll
          for (int dt = 0; dt < 4; ++dt) {
            put4(key, (int)D + 16 * dt + 4 * g, adk[i][dt]);
            put4(key, 2 * (int)D + 16 * dt + 4 * g, adv[i][dt]);
          }
        }
      } else {
#pragma unroll
      for (int dp = 0; dp < 4; dp += 2) {
        const u32x4 wk = pair16(adk[i][dp], adk[i][dp + 1]), wv = pair16(adv[i][dp], adv[i][dp + 1]);
        if (kt < nkt_valid && key < N) {
          *reinterpret_cast<u32x4*>(dbase + (long)key * 3 * D + D + pair16_off(dp, g)) = wk;
          *reinterpret_cast<u32x4*>(dbase + (long)key * 3 * D + 2 * D + pair16_off(dp, g)) = wv;
        }
      }
      }
    }
    if (colsum) {                                        // padded keys: p = 0 -> exact zeros
#pragma unroll
      for (int dt = 0; dt < 4; ++dt)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float vk = rowsum16(adk[0][dt][r] + adk[1][dt][r]), vv = rowsum16(adv[0][dt][r] + adv[1][dt][r]);
          if ((lane & 15) == 0) {
            sCs[wave * 192 + 64 + dt * 16 + 4 * g + r] += vk;
            sCs[wave * 192 + 128 + dt * 16 + 4 * g + r] += vv;
          }
        }
    }
  }

  if (colsum) {                                          // to_qkv bias-gradient partials of this (image, head)
    __syncthreads();
    attn_colsum_store<4>(sCs, colsum + (long)b * 3 * D + h * 64, D, tid);
  }
}

template <typename K>
int set_smem(K kernel, int bytes) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) ==
                 hipSuccess
             ? 0
             : -1;
}

constexpr int fwd_smem(int nkt) { return nkt * 16 * 128 * 2; }
constexpr int bwd_smem(int nkt) { return nkt * 16 * 128 * 2 + 2 * 8192 + 2 * nkt * 16 * 64 + 2 * nkt * 16 * 4; }

}  // namespace

namespace {
std::atomic<int> g_bwd_variant{0};
std::atomic<int> g_fwd_variant{getenv("MV_ATTN_FWD") ? atoi(getenv("MV_ATTN_FWD")) : 0};   // 0 auto | 1 one query tile per wave pass | 2 pairs
}
extern "C" int mv_attention_fwd_force(int variant) {
  if (variant != 0 && variant != 1 && variant != 2 && variant != 3) return MV_ERR_UNSUPPORTED;
  g_fwd_variant.store(variant, std::memory_order_relaxed);
  return MV_OK;
}
extern "C" int mv_attention_bwd_force(int variant) {
  if (variant != 0 && variant != 2 && variant != 4 && variant != 5 && variant != 8) return MV_ERR_UNSUPPORTED;
  g_bwd_variant.store(variant, std::memory_order_relaxed);
  return MV_OK;
}

extern "C" int mv_attention_fwd(const void* qkv, void* out, float* lse, int B, int N, int H, float scale,
                                mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && N > 0 && H > 0 && N <= 320, MV_ERR_SHAPE);
  MV_REQUIRE(mv_aligned16(qkv) && mv_aligned16(out), MV_ERR_ALIGN);
  if (B == 0) return MV_OK;
  hipStream_t s = (hipStream_t)stream;
  const float sl = scale * LOG2E;
  const int fv = g_fwd_variant.load(std::memory_order_relaxed);
  if (N <= 208 && (fv == 3 || fv == 0)) {    // 13 key tiles, 53 248 B: three workgroups per CU
    constexpr int smem13 = 2 * 13 * 16 * 128;
    const int a = MV_ONCE_PER_DEVICE(set_smem(attn_fwd13_kernel<false>, smem13));
    if (a) return MV_ERR_LAUNCH;
    attn_fwd13_kernel<false><<<B * H, 256, smem13, s>>>((const bf16_t*)qkv, (bf16_t*)out, lse, N, H, sl);
  } else if (N <= 224 && fv == 2) {          // query-tile pairs per wave: half the LDS fragment traffic per FLOP
    const int a = MV_ONCE_PER_DEVICE(set_smem(attn_fwd2_kernel<14>, fwd_smem(14)));
    if (a) return MV_ERR_LAUNCH;
    attn_fwd2_kernel<14><<<B * H, 256, fwd_smem(14), s>>>((const bf16_t*)qkv, (bf16_t*)out, lse, N, H, sl);
  } else if (N <= 224) {
    const int a = MV_ONCE_PER_DEVICE(set_smem(attn_fwd_kernel<14>, fwd_smem(14)));
    if (a) return MV_ERR_LAUNCH;
    attn_fwd_kernel<14><<<B * H, 256, fwd_smem(14), s>>>((const bf16_t*)qkv, (bf16_t*)out, lse, N, H, sl);
  } else if (N <= 288) {                   // 257 tokens at 256^2: 18 key tiles = 73.7 KB of K/V, still two workgroups per CU
    const int a = MV_ONCE_PER_DEVICE(set_smem(attn_fwd_kernel<18>, fwd_smem(18)));
    if (a) return MV_ERR_LAUNCH;
    attn_fwd_kernel<18><<<B * H, 256, fwd_smem(18), s>>>((const bf16_t*)qkv, (bf16_t*)out, lse, N, H, sl);
  } else {
    const int a = MV_ONCE_PER_DEVICE(set_smem(attn_fwd_kernel<20>, fwd_smem(20)));
    if (a) return MV_ERR_LAUNCH;
    attn_fwd_kernel<20><<<B * H, 256, fwd_smem(20), s>>>((const bf16_t*)qkv, (bf16_t*)out, lse, N, H, sl);
  }
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// ------------------------------------------------------------------------------------------------
// precision "bf16x3": the fused attention core on IEEE-half operands (N <= 208, i.e. the 197-token case)
// ------------------------------------------------------------------------------------------------
// The Linear products of that mode are 2^-16 accurate; an fp32 attention core on the f32 MFMA (1/16 of the half rate) was 23 % of
// its step.  Half operands (11 significand bits: every rounding 8x below bf16's) with fp32 accumulation, fp32 softmax and fp32
// outputs keep the end-to-end error inside 1e-3 (tests/test_vit_parity.py) at the bf16 kernels' speed -- they ARE those kernels,
// instantiated with F16 = true.  Forward: q, k, v are O(1) and p in [0, 1]: no range problem.  Backward: gradients can sit far
// below half's normal range (2^-14), so each (image, head) slice of dO is multiplied by a power of two s chosen from its largest
// magnitude (mv_attention_bwd_prep_f16: max |dO| -> 2^8, which leaves 2^8 of headroom for dP = dO V^T and dS); s multiplies everything
// linear in dO -- delta, dP, dS, dQ, dK, dV -- and the kernel divides it out of its fp32 outputs, exactly (a power of two).
// s = 2^(8 - ceil(log2(amax))) (1 for an all-zero or non-finite gradient)
__device__ __forceinline__ float attn_grad_scale(float a) {
  if (!(a > 0.f) || a > 3.0e38f) return 1.0f;
  int e;
  frexpf(a, &e);                                          // a = f * 2^e, f in [0.5, 1)  ->  a * 2^(8 - e) in [2^7, 2^8)
  e = 8 - e;
  e = e > 100 ? 100 : (e < -100 ? -100 : e);
  return ldexpf(1.0f, e);
}
// One workgroup per (image, head): pass 1 = the largest |dO| of the head's [N, 64] slice -> its scale s; pass 2 (the slice is
// 50 KB: still in L2) = dO * s -> half, delta[b, h, n] = sum_d half(dO s)_d O_d.  Thread = (row tid >> 4 of 16, four features).
// delta uses the ROUNDED gradient: the kernel's dP = dO16 V16^T is exact in fp32 and the forward's O = sum_k P16_k V16_k, so
// sum_d dO16_d O_d = sum_k P16_k dP_k and the cancellation in dS = P (dP - delta) is consistent (with the unrounded dO the two
// sides of that difference carried different roundings).
__global__ __launch_bounds__(256) void attn_bwd_prep_f16_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                                _Float16* __restrict__ dout16, float* __restrict__ delta,
                                                                float* __restrict__ gscale, int N, int H) {
  __shared__ float red[4];
  const int b = blockIdx.x / H, h = blockIdx.x % H, tid = threadIdx.x;
  const long D = (long)H * 64;
  const float* db = dout + (long)b * N * D + h * 64 + 4 * (tid & 15);
  const float* ob = out + (long)b * N * D + h * 64 + 4 * (tid & 15);
  _Float16* wb = dout16 + (long)b * N * D + h * 64 + 4 * (tid & 15);
  float m = 0.f;
  for (int r = tid >> 4; r < N; r += 16) {
    const float4 v = *reinterpret_cast<const float4*>(db + (long)r * D);
    m = fmaxf(fmaxf(fmaxf(fabsf(v.x), fabsf(v.y)), fmaxf(fabsf(v.z), fabsf(v.w))), m);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
  if ((tid & 63) == 0) red[tid >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  const float s = attn_grad_scale(m == m ? m : 0.f);
  if (tid == 0) gscale[blockIdx.x] = s;
  for (int r = tid >> 4; r < N; r += 16) {
    const float4 v = *reinterpret_cast<const float4*>(db + (long)r * D);
    const float4 o = *reinterpret_cast<const float4*>(ob + (long)r * D);
    const f16x4_t w = {(_Float16)(v.x * s), (_Float16)(v.y * s), (_Float16)(v.z * s), (_Float16)(v.w * s)};
    *reinterpret_cast<f16x4_t*>(wb + (long)r * D) = w;
    float dl = (float)w[0] * o.x + (float)w[1] * o.y + (float)w[2] * o.z + (float)w[3] * o.w;
    dl += __shfl_xor(dl, 1, 64);
    dl += __shfl_xor(dl, 2, 64);
    dl += __shfl_xor(dl, 4, 64);
    dl += __shfl_xor(dl, 8, 64);
    if ((tid & 15) == 0) delta[((long)b * H + h) * N + r] = dl;
  }
}

extern "C" int mv_attention_fwd_f16(const void* qkv16, float* out, float* lse, int B, int N, int H, float scale,
                                    mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && N > 0 && H > 0, MV_ERR_SHAPE);
  MV_REQUIRE(N <= 288, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(mv_aligned16(qkv16) && mv_aligned16(out), MV_ERR_ALIGN);
  if (B == 0) return MV_OK;
  if (N <= 208) {
    constexpr int smem13 = 2 * 13 * 16 * 128;
    const int a = MV_ONCE_PER_DEVICE(set_smem(attn_fwd13_kernel<true>, smem13));
    if (a) return MV_ERR_LAUNCH;
    attn_fwd13_kernel<true><<<B * H, 256, smem13, (hipStream_t)stream>>>((const bf16_t*)qkv16, (bf16_t*)out, lse, N, H, scale * LOG2E);
  } else {                                   // the 257-token case (256^2 inputs): 18 key tiles, two workgroups per CU
    const int a = MV_ONCE_PER_DEVICE(set_smem(attn_fwd_kernel<18, true>, fwd_smem(18)));
    if (a) return MV_ERR_LAUNCH;
    attn_fwd_kernel<18, true><<<B * H, 256, fwd_smem(18), (hipStream_t)stream>>>((const bf16_t*)qkv16, (bf16_t*)out, lse, N, H, scale * LOG2E);
  }
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_attention_bwd_prep_f16(const float* dout, const float* out, void* dout16, float* delta, float* gscale,
                                         int B, int N, int H, mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && N > 0 && H > 0, MV_ERR_SHAPE);
  MV_REQUIRE(mv_aligned16(dout) && mv_aligned16(out) && mv_aligned16(dout16) && gscale && delta, MV_ERR_ALIGN);
  if (B == 0) return MV_OK;
  attn_bwd_prep_f16_kernel<<<B * H, 256, 0, (hipStream_t)stream>>>(dout, out, (_Float16*)dout16, delta, gscale, N, H);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_attention_bwd_f16(const void* qkv16, const void* dout16, const float* delta, const float* lse,
                                    const float* gscale, void* dqkv, int nseg, float* colsum, int B, int N, int H, float scale,
                                    mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && N > 0 && H > 0, MV_ERR_SHAPE);
  MV_REQUIRE(N <= 288 && (nseg == 0 || nseg == 3 || nseg == 6), MV_ERR_UNSUPPORTED);
  MV_REQUIRE(mv_aligned16(qkv16) && mv_aligned16(dout16) && mv_aligned16(dqkv) && delta && lse && gscale, MV_ERR_ALIGN);
  if (B == 0) return MV_OK;
  constexpr int smem4 = 224 * 128 + 208 * 128 + 8192 + 224 * 64 + 2 * 224 * 4;
  hipStream_t s = (hipStream_t)stream;
#define MV_BWD_F16(SPLIT_)                                                                                                   \
  {                                                                                                                          \
    const int a = MV_ONCE_PER_DEVICE(set_smem(attn_bwd4_kernel<4, true, SPLIT_>, smem4));                                    \
    if (a) return MV_ERR_LAUNCH;                                                                                             \
    attn_bwd4_kernel<4, true, SPLIT_><<<B * H, 256, smem4, s>>>((const bf16_t*)qkv16, nullptr, (const bf16_t*)dout16, lse,   \
                                                                (bf16_t*)dqkv, colsum, N, H, scale, delta, gscale);           \
  }
#define MV_BWD2P_F16(SPLIT_)                                                                                                  \
  {                                                                                                                          \
    constexpr int smem2 = 2 * 288 * 128 + 2 * 288 * 4 + 4 * 192 * 4;                                                         \
    const int a = MV_ONCE_PER_DEVICE(set_smem(attn_bwd2p_kernel<9, true, SPLIT_>, smem2));                                   \
    if (a) return MV_ERR_LAUNCH;                                                                                             \
    attn_bwd2p_kernel<9, true, SPLIT_><<<B * H, 256, smem2, s>>>((const bf16_t*)qkv16, nullptr, (const bf16_t*)dout16, lse,  \
                                                                 (bf16_t*)dqkv, colsum, N, H, scale, delta, gscale);          \
  }
  if (N <= 208) {
    if (nseg == 0) MV_BWD_F16(0) else if (nseg == 3) MV_BWD_F16(3) else MV_BWD_F16(6)
  } else {                                   // 209 .. 288 tokens: the two-pass kernel (the 257-token case)
    if (nseg == 0) MV_BWD2P_F16(0) else if (nseg == 3) MV_BWD2P_F16(3) else MV_BWD2P_F16(6)
  }
#undef MV_BWD2P_F16
#undef MV_BWD_F16
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_attention_bwd(const void* qkv, const void* out, const void* dout, const float* lse, void* dqkv,
                                float* colsum, int B, int N, int H, float scale, mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && N > 0 && H > 0 && N <= 320, MV_ERR_SHAPE);
  MV_REQUIRE(mv_aligned16(qkv) && mv_aligned16(out) && mv_aligned16(dout) && mv_aligned16(dqkv), MV_ERR_ALIGN);
  if (B == 0) return MV_OK;
  hipStream_t s = (hipStream_t)stream;
  // measured (tools/bench_attn.py, ViT-B, 12 heads): N = 197, batch 256: bwd4 251 us, two-pass 287 us; N = 257, batch 64:
  // two-pass 101 us, eight-wave 115 us.  mv_attention_bwd_force overrides (tests run every variant at every length it takes).
  const int forced = g_bwd_variant.load(std::memory_order_relaxed);
  const bool force8 = forced == 8;
  const bool two_pass = forced == 2 || (forced == 0 && N > 208);
  if (N <= 288 && !force8 && two_pass) {
    if (N <= 224) {
      constexpr int smem = 2 * 224 * 128 + 2 * 224 * 4 + 4 * 192 * 4;
      const int a = MV_ONCE_PER_DEVICE(set_smem(attn_bwd2p_kernel<7>, smem));
      if (a) return MV_ERR_LAUNCH;
      attn_bwd2p_kernel<7><<<B * H, 256, smem, s>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse,
                                                   (bf16_t*)dqkv, colsum, N, H, scale);
    } else {
      constexpr int smem = 2 * 288 * 128 + 2 * 288 * 4 + 4 * 192 * 4;
      const int a = MV_ONCE_PER_DEVICE(set_smem(attn_bwd2p_kernel<9>, smem));
      if (a) return MV_ERR_LAUNCH;
      attn_bwd2p_kernel<9><<<B * H, 256, smem, s>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse,
                                                   (bf16_t*)dqkv, colsum, N, H, scale);
    }
  } else if (N <= 208 && !force8) {
    constexpr int smem4 = 224 * 128 + 208 * 128 + 8192 + 224 * 64 + 2 * 224 * 4;   // 79,616 B: two workgroups per CU
    const int a = MV_ONCE_PER_DEVICE(set_smem(attn_bwd4_kernel<4>, smem4) | set_smem(attn_bwd4_kernel<2>, smem4));
    if (a) return MV_ERR_LAUNCH;
    if (forced == 5 && N > 192)                   // two waves of 512 registers per workgroup (7 + 6 key tiles)
      attn_bwd4_kernel<2><<<B * H, 128, smem4, s>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse,
                                                   (bf16_t*)dqkv, colsum, N, H, scale);
    else
      attn_bwd4_kernel<4><<<B * H, 256, smem4, s>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout, lse,
                                                   (bf16_t*)dqkv, colsum, N, H, scale);
  } else if (N <= 224) {
    const int a = MV_ONCE_PER_DEVICE(set_smem(attn_bwd_kernel<14, 2>, bwd_smem(14)));
    if (a) return MV_ERR_LAUNCH;
    attn_bwd_kernel<14, 2><<<B * H, 512, bwd_smem(14), s>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout,
                                                          lse, (bf16_t*)dqkv, colsum, N, H, scale);
  } else {
    const int a = MV_ONCE_PER_DEVICE(set_smem(attn_bwd_kernel<20, 3>, bwd_smem(20)));
    if (a) return MV_ERR_LAUNCH;
    attn_bwd_kernel<20, 3><<<B * H, 512, bwd_smem(20), s>>>((const bf16_t*)qkv, (const bf16_t*)out, (const bf16_t*)dout,
                                                          lse, (bf16_t*)dqkv, colsum, N, H, scale);
  }
  MV_CHECK_LAUNCH();
  return MV_OK;
}
