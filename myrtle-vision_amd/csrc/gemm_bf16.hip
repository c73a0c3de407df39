// bf16 MFMA contractions for the ViT Linear layers (gfx950, v_mfma_f32_16x16x32_bf16).
//
//   mv_gemm_nt_bf16 : C[M,N] = A[M,K] . B[N,K]^T (+ fused epilogue)   -- nn.Linear forward (vit.py:278,86,98,
//                     48-51,333,354) and, with B = W^T, the input gradient dX = dY . W.
//   mv_gemm_tn_bf16 : C[M,N] (+)= A[Kc,M]^T . B[Kc,N]                  -- weight gradient dW = dY^T . X.
//
// Structure (both): 128x128 output tile per 256-thread workgroup (4 waves as 2x2, 64x64 per wave =
// 4x4 MFMA tiles, 64 fp32 accumulator VGPRs), K-step 64, LDS double buffer (2 x 32 KiB, two workgroups
// per CU).  Two staging variants per kernel:
//   *_glds_kernel (contraction length % 64 == 0: every ViT-B/Tiny layer at the benchmark sizes): tiles go
//       HBM -> LDS directly with global_load_lds_dwordx4 (no VGPR round trip, no ds_write: the register-staged
//       form measured LDS-WRITE bound, ds_write_b128 moves only ~79 B/clk/CU).  The LDS destination of one
//       wave-instruction is lane-linear (1 KiB), so the XOR swizzle is applied to each lane's SOURCE address
//       and to the fragment reads (same involution).  The load of K-step t+1 is issued before step t computes
//       and retired by a COUNTED s_waitcnt vmcnt(8) + raw s_barrier, so it stays in flight across the MFMAs.
//   *_kernel (register-staged fallback, any shape): global->VGPR loads issued one K-step ahead, zero-filled
//       by select (never a conditional load), written to LDS after the barrier, one barrier per K-step.
//
// LDS images (checked with tools/lds_bank_sim.py against the gfx950 banking rules):
//   NT: tiles are [128 rows][64 k] bf16 = 128-byte rows, 16-byte chunk index XORed with ((row>>1)&3)<<1.
//       Fragment reads are ds_read_b128 (8 consecutive k of one row): conflict-free.
//   TN: tiles are [64 kc][128 cols] bf16 = 256-byte rows, chunk ^= ((row&3)<<2)|((row>>2)&3).  Both MFMA
//       operands need 8 consecutive kc of one column = a column read: two ds_read_b64_tr_b16 (hardware
//       transpose) per fragment, conflict-free in this image.  No transposed copies of activations exist.
//
// The MFMA is issued as D' = B_frag x A_frag, i.e. it produces the TRANSPOSE of the output tile, so a lane
// holds 4 consecutive output columns of one row: epilogue loads/stores are 8-/16-byte vectors.
//
// Workgroup ids are remapped so the 8 XCDs each walk a contiguous run of tiles (n fastest): the blocks
// sharing one A row-panel run on one XCD and hit its L2.
#include "mv_common.h"

#include <atomic>
#include <type_traits>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int STAGE_BYTES = (BM + BN) * BK * 2;  // 32 KiB
constexpr int SMEM_BYTES = 2 * STAGE_BYTES;      // 64 KiB

__device__ __forceinline__ int sw128(int row, int ch) { return row * 128 + ((ch ^ (((row >> 1) & 3) << 1)) << 4); }
__device__ __forceinline__ int sw256(int row, int ch) {
  return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4);
}

// bijective XCD-aware remap of a 1-D grid (cdna guide T1): XCD x (= bid % 8) owns a contiguous run of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

struct EpiArgs {
  float alpha;          // C = alpha * acc (+ bias ...): 1 except for the integer-code GEMMs of the int8 path
  const float* bias;
  const void* aux;
  int ld_aux;
  int aux_i;
  void* out2;
  int ld_out2;
  const int* icorr;     // int8 GEMM only: per-output-column integer added to the int32 dot product before scaling
  float q_inv, q_zp;    // MV_EPI_GELU_Q8: 1 / scale and zero point of the next layer's quint8 quantiser
  // K-split launches of gemm_nt_8phase_kernel<.., KSPLIT = true>: work item t covers output tile t % ks_tiles over the
  // contraction slice t / ks_tiles (ks_len elements long) and writes slab t / ks_tiles (ks_slab elements apart)
  int ks_tiles, ks_len;
  long ks_slab;
  int band;             // 8-phase kernel: > 0 = an XCD walks column BANDS of this many tile columns (its B slice stays in its L2)
  // OPK = NT_F8C (mv_gemm_nt_f8c): the first f8_tiles16 K-tiles of a row are bf16, the rest e4m3 whose products carry 2^(f8_scale - 127)
  int f8_tiles16, f8_scale;
};

// 16-byte output store of the NT epilogues
__device__ __forceinline__ void st16(void* p, u32x4 v) { *reinterpret_cast<u32x4*>(p) = v; }

template <typename CT>
__device__ __forceinline__ void store4(CT* p, const float v[4], bool vec, int nvalid) {
  if (vec) {
    if constexpr (sizeof(CT) == 4) {
      *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      typedef CT ct4 __attribute__((ext_vector_type(4)));       // bf16 or IEEE half
      const ct4 o = {(CT)v[0], (CT)v[1], (CT)v[2], (CT)v[3]};
      *reinterpret_cast<ct4*>(p) = o;
    }
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (r < nvalid) p[r] = (CT)v[r];
  }
}

// Column sums of one wave's 64x64 quadrant (bias-gradient partials): cs[j][r] holds this lane's sum over its 4 rows
// per MFMA tile; the 16 lanes with equal lane>>4 hold the other rows of the same columns -> xor-shuffle reduce, then
// lane&15 == 0 writes partial[row_block][n .. n+3] for j = 0..3 (each (row_block, column) has exactly one writer).
__device__ __forceinline__ void nt_colsum_flush(float (&cs)[4][4], float* __restrict__ partial, int ld, int row_block, int nb,
                                                int N, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float v = cs[j][r];
      v += __shfl_xor(v, 1, 64);
      v += __shfl_xor(v, 2, 64);
      v += __shfl_xor(v, 4, 64);
      v += __shfl_xor(v, 8, 64);
      cs[j][r] = v;
    }
  if ((lane & 15) == 0) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + j * 16 + r;
        if (n < N) partial[(long)row_block * ld + n] = cs[j][r];
      }
  }
}

// ---- shared epilogue of the NT kernels.  Lane holds, for MFMA tile (i,j): row m = .. + (lane&15), 4 consecutive
// columns n = .. + 4*(lane>>4) + 0..3 (the MFMA was issued transposed), so bias/aux/out are 8-/16-byte vectors.
// erf to |err| <= 1.5e-7 (Abramowitz-Stegun 7.1.26) on v_rcp_f32 / v_exp_f32: ~13 VALU ops instead of libm erff's
// branchy ~40.  Used only where the result is rounded to bf16 (2^-9) anyway; the fp32 parity path keeps erff.
// Also returns e = exp(-u^2), which GELU' needs as its Gaussian factor.
__device__ __forceinline__ float erf_fast(float u, float& e) { return mv_erf_fast(u, e); }
__device__ __forceinline__ float gelu_fast(float x) {
  float e;
  return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f, e));
}
// gelu(x) and gelu'(x) together (they share the erf and the Gaussian factor): the forward epilogue of fc1 can leave
// gelu'(h) for the backward pass, whose epilogue is then a plain multiply (MV_EPI_GELU_GRAD / MV_EPI_MUL)
__device__ __forceinline__ void gelu_both_fast(float x, float& gl, float& dg) {
  float e;
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f, e));
  gl = x * cdf;
  dg = fmaf(x * 0.39894228040143267794f, e, cdf);
}
constexpr bool epi_is_gelu(int e) { return e == MV_EPI_GELU || e == MV_EPI_GELU_GRAD || e == MV_EPI_GELU_GRAD8; }
constexpr bool epi_is_dgelu(int e) { return e == MV_EPI_DGELU || e == MV_EPI_MUL || e == MV_EPI_MUL8; }
constexpr bool epi_is_gelugrad(int e) { return e == MV_EPI_GELU_GRAD || e == MV_EPI_GELU_GRAD8; }
constexpr bool epi_is_split(int e) { return e == MV_EPI_SPLIT_DGELU || e == MV_EPI_SPLIT_GELU; }
// four consecutive columns of a split-output epilogue: the bf16 pieces of v (split3_kernel's arithmetic), nseg segments `seg` apart
__device__ __forceinline__ void store_pieces4(bf16_t* o, const float (&v)[4], int nseg, long seg) {
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
  *reinterpret_cast<bf16x4*>(o) = p[0];
  *reinterpret_cast<bf16x4*>(o + seg) = p[0];
  *reinterpret_cast<bf16x4*>(o + 2 * seg) = p[1];
  if (nseg == 6) {
    *reinterpret_cast<bf16x4*>(o + 3 * seg) = p[0];
    *reinterpret_cast<bf16x4*>(o + 4 * seg) = p[1];
    *reinterpret_cast<bf16x4*>(o + 5 * seg) = p[2];
  }
}
// gelu'(x) lies in [-0.1290, 1.1290]: MV_EPI_GELU_GRAD8 leaves it as an 8-bit code on a fixed grid, MV_EPI_MUL8 reads it
// back: one byte per hidden element instead of two in fc1's epilogue and in fc2-dX's.  The grid is step 0.005 from -0.13:
// code 26 IS 0 and code 226 IS 1 (a saturated unit's gradient passes unchanged and a dead unit leaks nothing -- a grid
// without those two points gives the two most common values of gelu' a systematic, not zero-mean, error), code 255 = 1.145
// covers the maximum 1.129; |error| <= 0.0025 elsewhere.  Decoding is (code - 26) * 0.005f, NOT fma(code, step, lo): the
// subtraction is exact and 200 * 0.005f rounds to exactly 1.0f.
constexpr float GQ_STEP = 0.005f, GQ_INV = 200.0f, GQ_ZERO = 26.0f;
__device__ __forceinline__ unsigned gq_code(float g) {
  return (unsigned)__builtin_amdgcn_fmed3f(fmaf(g, GQ_INV, GQ_ZERO + 0.5f), 0.f, 255.f);
}
__device__ __forceinline__ unsigned gq_pack4(const float (&g)[4]) {
  return gq_code(g[0]) | (gq_code(g[1]) << 8) | (gq_code(g[2]) << 16) | (gq_code(g[3]) << 24);
}
__device__ __forceinline__ float gq_value(float code) { return (code - GQ_ZERO) * GQ_STEP; }
__device__ __forceinline__ float gq_decode(unsigned w, int k) { return gq_value((float)((w >> (8 * k)) & 255u)); }
__device__ __forceinline__ float dgelu_fast(float x) {
  float e;
  const float cdf = 0.5f * (1.0f + erf_fast(x * 0.70710678118654752440f, e));
  return fmaf(x * 0.39894228040143267794f, e, cdf);          // e = exp(-x^2 / 2)
}

// ---- shared epilogue of the NT kernels.  Lane holds, for MFMA tile (i,j): row m = .. + (lane&15), 4 consecutive
// columns n = .. + 4*(lane>>4) + 0..3 (the MFMA was issued transposed), so bias/aux/out are 8-/16-byte vectors.
// Interior tiles take the fast path: ALL bias/aux vectors are loaded before the first store.  (aux and C are
// different buffers by contract, but the compiler cannot know, and interleaving "load aux, store C" serialises
// sixteen HBM round trips per lane: the +residual GEMM measured 242 TFLOP/s that way.)
// bf16 outputs: a lane holds 4 consecutive columns (8 bytes) of each 16-column MFMA tile.  For an adjacent tile pair
// (j0, j1) v_permlane16_swap exchanges, per dword, tile j0's data in lanes with odd lane>>4 against tile j1's data in
// lanes with even lane>>4 (l <-> l^16).  Afterwards lane g = lane>>4 owns 8 consecutive columns (16 bytes) of ONE tile:
//   tile j0 + (g&1), columns 8 (g>>1) .. +7   ->   half as many, twice as wide stores (the store tail of a 256x256
// bf16 tile is issue-bound: 32 dwordx2 per lane before, 16 dwordx4 now; cdna guide T21).
template <typename CT = bf16_t>
__device__ __forceinline__ u32x4 pair_swap_bf16(const float (&va)[4], const float (&vb)[4]) {
  typedef CT ct4 __attribute__((ext_vector_type(4)));         // bf16 (default) or IEEE half (MV_F16 outputs, round 4)
  const ct4 pa = {(CT)va[0], (CT)va[1], (CT)va[2], (CT)va[3]};
  const ct4 pb = {(CT)vb[0], (CT)vb[1], (CT)vb[2], (CT)vb[3]};
  typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
  const u32x2_t a = __builtin_bit_cast(u32x2_t, pa), b = __builtin_bit_cast(u32x2_t, pb);
  const u32x2_t r0 = __builtin_amdgcn_permlane16_swap(a[0], b[0], false, false);
  const u32x2_t r1 = __builtin_amdgcn_permlane16_swap(a[1], b[1], false, false);
  return (u32x4){r0[0], r1[0], r0[1], r1[1]};
}

// The aux operand of one 64 x 64 quadrant, loaded AHEAD of the epilogue that consumes it.  The 8-phase kernel finishes its two
// quadrant-rows one after the other; each nt_epilogue call used to begin with its own aux loads, so the second call's loads
// (a full memory round trip while every CU of the chip is in its store burst) sat exposed between the first call's stores and
// the second call's arithmetic.  With both quadrants' loads issued before the first call, the second quadrant's data arrives
// under the first one's arithmetic and stores.  RESIDUAL: the fp32 residual tile (64 VGPRs per quadrant; the second
// accumulator half is parked in LDS meanwhile); MUL8: the gelu' codes (4 x 16 bytes per lane).
template <int EPI>
struct EpiPre {
  float4 ax[EPI == MV_EPI_RESIDUAL ? 4 : 1][EPI == MV_EPI_RESIDUAL ? 4 : 1];
  u32x4 xq[EPI == MV_EPI_MUL8 ? 4 : 1];
};
template <int EPI>
__device__ __forceinline__ void nt_epi_prefetch(EpiPre<EPI>& P, int m0, int n0, int wm, int wn, int lane, const EpiArgs& ep) {
  const int mb = m0 + wm * 64 + (lane & 15), nb = n0 + wn * 64 + 4 * (lane >> 4);
  if constexpr (EPI == MV_EPI_RESIDUAL) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        P.ax[i][j] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ep.aux) + (long)(mb + i * 16) * ep.ld_aux + nb + j * 16);
  } else if constexpr (EPI == MV_EPI_MUL8) {
    const unsigned char* a8 = reinterpret_cast<const unsigned char*>(ep.aux);
#pragma unroll
    for (int i = 0; i < 4; ++i)
      P.xq[i] = *reinterpret_cast<const u32x4*>(a8 + (long)(mb + i * 16) * ep.ld_aux + n0 + wn * 64 + 16 * (lane >> 4));
  }
}
// wave-uniform: may this quadrant's aux be prefetched (the conditions of nt_epilogue's fast paths that read it that way)?
template <int EPI>
__device__ __forceinline__ bool nt_epi_prefetch_ok(int M, int N, int m0, int n0, int ldc, const EpiArgs& ep) {
  const bool interior = (m0 + BM <= M) && (n0 + BN <= N) && ((ldc & 3) == 0) && ((ep.ld_aux & 3) == 0) &&
                        ((ep.ld_out2 & 3) == 0) && ((reinterpret_cast<uintptr_t>(ep.bias) & 15) == 0);
  if constexpr (EPI == MV_EPI_MUL8)
    return interior && (ep.ld_aux & 15) == 0 && (reinterpret_cast<uintptr_t>(ep.aux) & 15) == 0;
  return interior && EPI == MV_EPI_RESIDUAL;
}

template <int EPI, typename CT, bool PRE = false>
__device__ __forceinline__ void nt_epilogue(f32x4 (&acc)[4][4], CT* __restrict__ C, int ldc, int M, int N, int m0, int n0,
                                            int wm, int wn, int lane, const EpiArgs& ep, const float* bias_lds = nullptr,
                                            const EpiPre<EPI>& pre = EpiPre<EPI>{}) {
  // PRE: this quadrant's aux tile is already in ``pre`` (nt_epi_prefetch; the caller checked nt_epi_prefetch_ok, i.e. the
  // interior fast path below is the one that runs)
  // bias_lds (persistent kernel): the bias of this 128-column region staged in LDS (zeros when there is none); indexed
  // relative to the region, so the epilogue issues no global load at all
  // (m0, n0) = origin of the 128x128 region this call covers; wave (wm, wn) owns its 64x64 quadrant
  const bool interior = (m0 + BM <= M) && (n0 + BN <= N) && ((ldc & 3) == 0) && ((ep.ld_aux & 3) == 0) &&
                        ((ep.ld_out2 & 3) == 0) && ((reinterpret_cast<uintptr_t>(ep.bias) & 15) == 0);
  if (interior) {
    const int mb = m0 + wm * 64 + (lane & 15), nb = n0 + wn * 64 + 4 * (lane >> 4);
    float4 bv[4];                                      // (the input-gradient epilogues never carry a bias: constants there)
#pragma unroll
    for (int j = 0; j < 4; ++j)
      bv[j] = bias_lds ? *reinterpret_cast<const float4*>(bias_lds + wn * 64 + 4 * (lane >> 4) + j * 16)
              : (!epi_is_dgelu(EPI) && ep.bias) ? *reinterpret_cast<const float4*>(ep.bias + nb + j * 16)
                                                : make_float4(0.f, 0.f, 0.f, 0.f);
    long crow[4];
    int prow[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int m = mb + i * 16;
      crow[i] = m;
      prow[i] = 0;
      if constexpr (EPI == MV_EPI_EMBED) {
        const int img = m / ep.aux_i;
        prow[i] = 1 + (m - img * ep.aux_i);
        crow[i] = (long)img * (ep.aux_i + 1) + prow[i];
      }
    }
    if constexpr (EPI == MV_EPI_GELU_Q8) {
      // Linear -> GELU -> the next Linear's quint8 quantiser, on the accumulators: v is the fp32 Linear output exactly as the
      // MV_EPI_NONE epilogue would store it, affine_i8_pack4<1> is the standalone quantiser's own function (erf GELU, rint,
      // clamp): same int8 codes, and the [M, N] fp32 hidden tensor (2.5 GB at batch 1024) never exists
      const bool wide = (ldc & 15) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0;     // wave-uniform
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        unsigned w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float v0 = fmaf(acc[i][j][0], ep.alpha, bv[j].x), v1 = fmaf(acc[i][j][1], ep.alpha, bv[j].y);
          const float v2 = fmaf(acc[i][j][2], ep.alpha, bv[j].z), v3 = fmaf(acc[i][j][3], ep.alpha, bv[j].w);
          w[j] = affine_i8_pack4<1>(v0, v1, v2, v3, ep.q_inv, ep.q_zp);
        }
        char* crow8 = reinterpret_cast<char*>(C) + (long)(mb + i * 16) * ldc;
        if (wide) {
          // 4 x 4 transpose over (column tile, lane group), as in MV_EPI_GELU_GRAD8: lane group g ends with the 16 codes of
          // tile g -- one 16-byte store per 16 rows (64 contiguous bytes per row) instead of four 4-byte ones
          typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
          const u32x2_t s01 = __builtin_amdgcn_permlane16_swap(w[0], w[1], false, false);
          const u32x2_t s23 = __builtin_amdgcn_permlane16_swap(w[2], w[3], false, false);
          const u32x2_t t02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
          const u32x2_t t13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
          st16(crow8 + n0 + wn * 64 + 16 * (lane >> 4), (u32x4){t02[0], t13[0], t02[1], t13[1]});
        } else {
#pragma unroll
          for (int j = 0; j < 4; ++j) *reinterpret_cast<unsigned*>(crow8 + nb + j * 16) = w[j];
        }
      }
      return;
    }
    float4 ax[4][4];
    if constexpr (EPI == MV_EPI_RESIDUAL || EPI == MV_EPI_EMBED || EPI == MV_EPI_SPLIT_DGELU) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          if constexpr (EPI == MV_EPI_RESIDUAL && PRE) {    // loaded ahead by the kernel (nt_epi_prefetch)
            ax[i][j] = pre.ax[i][j];
            continue;
          }
          const long arow = (EPI == MV_EPI_EMBED) ? prow[i] : (mb + i * 16);
          ax[i][j] = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(ep.aux) + arow * ep.ld_aux + nb + j * 16);
        }
    }
    bf16x4 hx[4][4];                                   // DGELU: the saved pre-activation, kept packed (32 VGPRs, not 64)
    unsigned hq[4][4];                                 // MUL8: four 8-bit codes of gelu'
    if constexpr (EPI == MV_EPI_MUL8) {
      const unsigned char* a8 = reinterpret_cast<const unsigned char*>(ep.aux);
      if ((ep.ld_aux & 15) == 0 && (reinterpret_cast<uintptr_t>(a8) & 15) == 0) {
        // one 16-byte load per 16 rows (lane group g takes the 16 codes of column tile g), then the 4 x 4 transpose over
        // (word, lane group) of MV_EPI_GELU_GRAD8's store, which is its own inverse: word j = tile j, columns 4g .. 4g+3
        typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
        u32x4 x[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
          x[i] = PRE ? pre.xq[i]
                     : *reinterpret_cast<const u32x4*>(a8 + (long)(mb + i * 16) * ep.ld_aux + n0 + wn * 64 + 16 * (lane >> 4));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const u32x2_t s01 = __builtin_amdgcn_permlane16_swap(x[i][0], x[i][1], false, false);
          const u32x2_t s23 = __builtin_amdgcn_permlane16_swap(x[i][2], x[i][3], false, false);
          const u32x2_t t02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
          const u32x2_t t13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
          hq[i][0] = t02[0]; hq[i][1] = t13[0]; hq[i][2] = t02[1]; hq[i][3] = t13[1];
        }
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            hq[i][j] = *reinterpret_cast<const unsigned*>(a8 + (long)(mb + i * 16) * ep.ld_aux + nb + j * 16);
      }
    } else if constexpr (epi_is_dgelu(EPI)) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          hx[i][j] = *reinterpret_cast<const bf16x4*>(reinterpret_cast<const bf16_t*>(ep.aux) +
                                                      (long)(mb + i * 16) * ep.ld_aux + nb + j * 16);
    }
    float cs[4][4];                                   // DGELU: column sums of this wave's 64x64 quadrant
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[j][r] = 0.f;
    if constexpr (sizeof(CT) == 2 && (EPI == MV_EPI_NONE || epi_is_gelu(EPI) || epi_is_dgelu(EPI))) {
      if (((ldc & 7) == 0) && ((ep.ld_out2 & (EPI == MV_EPI_GELU_GRAD8 ? 15 : 7)) == 0 || !epi_is_gelu(EPI))) {
        const int g = lane >> 4;
        const int nw = n0 + wn * 64 + 16 * (g & 1) + 8 * (g >> 1);      // + 32 jp: first of this lane's 8 columns
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          unsigned gw[4] = {0u, 0u, 0u, 0u};                 // GELU_GRAD8: this row tile's codes, one word per column tile
#pragma unroll
          for (int jp = 0; jp < 2; ++jp) {
            float v[2][4], hpre[2][4];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
              const int j = 2 * jp + q;
              v[q][0] = fmaf(acc[i][j][0], ep.alpha, bv[j].x); v[q][1] = fmaf(acc[i][j][1], ep.alpha, bv[j].y);
              v[q][2] = fmaf(acc[i][j][2], ep.alpha, bv[j].z); v[q][3] = fmaf(acc[i][j][3], ep.alpha, bv[j].w);
              if constexpr (epi_is_gelu(EPI)) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                  if constexpr (epi_is_gelugrad(EPI)) {
                    float gl, dg;
                    gelu_both_fast(v[q][r], gl, dg);
                    hpre[q][r] = dg;                       // out2 <- gelu'(pre-activation)
                    v[q][r] = gl;
                  } else {
                    hpre[q][r] = v[q][r];
                    v[q][r] = gelu_fast(v[q][r]);
                  }
                }
              } else if constexpr (epi_is_dgelu(EPI)) {
#pragma unroll
                for (int r = 0; r < 4; ++r)
                  v[q][r] *= (EPI == MV_EPI_MUL8) ? gq_decode(hq[i][j], r)
                             : (EPI == MV_EPI_MUL) ? (float)hx[i][j][r] : dgelu_fast((float)hx[i][j][r]);
                cs[j][0] += v[q][0]; cs[j][1] += v[q][1]; cs[j][2] += v[q][2]; cs[j][3] += v[q][3];
              }
            }
            if constexpr (EPI == MV_EPI_GELU_GRAD8) {
              gw[2 * jp] = gq_pack4(hpre[0]);
              gw[2 * jp + 1] = gq_pack4(hpre[1]);
            } else if constexpr (epi_is_gelu(EPI)) {
              if (ep.out2)
                st16(reinterpret_cast<bf16_t*>(ep.out2) + (long)(mb + i * 16) * ep.ld_out2 + nw + 32 * jp,
                     pair_swap_bf16(hpre[0], hpre[1]));
            }
            st16(C + crow[i] * ldc + nw + 32 * jp, pair_swap_bf16<CT>(v[0], v[1]));
          }
          if constexpr (EPI == MV_EPI_GELU_GRAD8) {
            // The lane holds one word (4 codes = columns 4g .. 4g+3) of each of the four 16-column tiles.  A 4 x 4 transpose
            // over (tile, lane group) -- two swap stages -- leaves lane group g with all 16 codes of tile g: ONE 16-byte
            // store per 16 rows instead of two 8-byte ones (the tail of this epilogue is bound by store issue, not bytes:
            // with 8-byte stores the byte-wide gelu' bought 0.14 ms per step instead of the 0.8 its bytes promised).
            if (ep.out2) {
              typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
              const u32x2_t s01 = __builtin_amdgcn_permlane16_swap(gw[0], gw[1], false, false);
              const u32x2_t s23 = __builtin_amdgcn_permlane16_swap(gw[2], gw[3], false, false);
              const u32x2_t t02 = __builtin_amdgcn_permlane32_swap(s01[0], s23[0], false, false);
              const u32x2_t t13 = __builtin_amdgcn_permlane32_swap(s01[1], s23[1], false, false);
              st16(reinterpret_cast<unsigned char*>(ep.out2) + (long)(mb + i * 16) * ep.ld_out2 + n0 + wn * 64 + 16 * (lane >> 4),
                   (u32x4){t02[0], t13[0], t02[1], t13[1]});
            }
          }
        }
        if constexpr (epi_is_dgelu(EPI)) {
          if (ep.out2) nt_colsum_flush(cs, reinterpret_cast<float*>(ep.out2), ep.ld_out2, (m0 + wm * 64) >> 6, nb, N, lane);
        }
        return;
      }
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v[4] = {fmaf(acc[i][j][0], ep.alpha, bv[j].x), fmaf(acc[i][j][1], ep.alpha, bv[j].y),
                      fmaf(acc[i][j][2], ep.alpha, bv[j].z), fmaf(acc[i][j][3], ep.alpha, bv[j].w)};
        const int n = nb + j * 16;
        if constexpr (epi_is_gelu(EPI)) {
          float o2[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            if constexpr (epi_is_gelugrad(EPI)) gelu_both_fast(o2[r], v[r], o2[r]);
            else v[r] = gelu_fast(v[r]);
          }
          if constexpr (EPI == MV_EPI_GELU_GRAD8) {
            if (ep.out2)
              *reinterpret_cast<unsigned*>(reinterpret_cast<unsigned char*>(ep.out2) + (long)(mb + i * 16) * ep.ld_out2 + n) =
                  gq_pack4(o2);
          } else {
            if (ep.out2) store4(reinterpret_cast<bf16_t*>(ep.out2) + (long)(mb + i * 16) * ep.ld_out2 + n, o2, true, 4);
          }
        } else if constexpr (EPI == MV_EPI_RESIDUAL || EPI == MV_EPI_EMBED) {
          v[0] += ax[i][j].x; v[1] += ax[i][j].y; v[2] += ax[i][j].z; v[3] += ax[i][j].w;
        } else if constexpr (EPI == MV_EPI_SPLIT_DGELU) {         // the exact gelu' of the split pass it replaces (split3_ex_kernel<2>)
          v[0] *= dgelu_f(ax[i][j].x); v[1] *= dgelu_f(ax[i][j].y); v[2] *= dgelu_f(ax[i][j].z); v[3] *= dgelu_f(ax[i][j].w);
          cs[j][0] += v[0]; cs[j][1] += v[1]; cs[j][2] += v[2]; cs[j][3] += v[3];
        } else if constexpr (EPI == MV_EPI_SPLIT_GELU) {          // pre-activation out (fp32), then the exact GELU (split3_ex_kernel<1>)
          st16(reinterpret_cast<float*>(ep.out2) + (long)(mb + i * 16) * ep.ld_out2 + n,
               __builtin_bit_cast(u32x4, (f32x4){v[0], v[1], v[2], v[3]}));
          v[0] = gelu_f(v[0]); v[1] = gelu_f(v[1]); v[2] = gelu_f(v[2]); v[3] = gelu_f(v[3]);
        } else if constexpr (epi_is_dgelu(EPI)) {
#pragma unroll
          for (int r = 0; r < 4; ++r)
            v[r] *= (EPI == MV_EPI_MUL8) ? gq_decode(hq[i][j], r)
                    : (EPI == MV_EPI_MUL) ? (float)hx[i][j][r] : dgelu_fast((float)hx[i][j][r]);
          cs[j][0] += v[0]; cs[j][1] += v[1]; cs[j][2] += v[2]; cs[j][3] += v[3];
        }
        if constexpr (epi_is_split(EPI))
          store_pieces4(reinterpret_cast<bf16_t*>(C) + crow[i] * ldc + n, v, ep.aux_i, N);
        else if constexpr (sizeof(CT) == 4)
          st16(C + crow[i] * ldc + n, __builtin_bit_cast(u32x4, (f32x4){v[0], v[1], v[2], v[3]}));
        else
          store4(C + crow[i] * ldc + n, v, true, 4);
      }
    if constexpr (epi_is_dgelu(EPI) || EPI == MV_EPI_SPLIT_DGELU) {
      if (ep.out2) nt_colsum_flush(cs, reinterpret_cast<float*>(ep.out2), ep.ld_out2, (m0 + wm * 64) >> 6, nb, N, lane);
    }
    return;
  }
  // ---- edge tiles: bounds-checked, element-wise where needed
  const bool ldc_vec = (ldc & 3) == 0;
  float cs[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int r = 0; r < 4; ++r) cs[j][r] = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= M) continue;
    long crow = m;
    int patch = 0;
    if constexpr (EPI == MV_EPI_EMBED) {
      const int img = m / ep.aux_i;
      patch = m - img * ep.aux_i;
      crow = (long)img * (ep.aux_i + 1) + 1 + patch;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
      if (n >= N) continue;
      const int nvalid = (N - n) >= 4 ? 4 : (N - n);
      const bool vec = ldc_vec && nvalid == 4;
      float v[4] = {acc[i][j][0] * ep.alpha, acc[i][j][1] * ep.alpha, acc[i][j][2] * ep.alpha, acc[i][j][3] * ep.alpha};
      if (ep.bias) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < nvalid) v[r] += ep.bias[n + r];
      }
      if constexpr (epi_is_gelu(EPI)) {
        float o2[4] = {v[0], v[1], v[2], v[3]};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (epi_is_gelugrad(EPI)) gelu_both_fast(o2[r], v[r], o2[r]);
          else v[r] = gelu_fast(v[r]);
        }
        if constexpr (EPI == MV_EPI_GELU_GRAD8) {
          if (ep.out2) {
            unsigned char* o8 = reinterpret_cast<unsigned char*>(ep.out2) + (long)m * ep.ld_out2 + n;
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (r < nvalid) o8[r] = (unsigned char)gq_code(o2[r]);
          }
        } else {
          if (ep.out2) store4(reinterpret_cast<bf16_t*>(ep.out2) + (long)m * ep.ld_out2 + n, o2, vec && (ep.ld_out2 & 3) == 0, nvalid);
        }
      } else if constexpr (EPI == MV_EPI_RESIDUAL) {
        const float* ax = reinterpret_cast<const float*>(ep.aux) + (long)m * ep.ld_aux + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < nvalid) v[r] += ax[r];
      } else if constexpr (EPI == MV_EPI_MUL8) {
        const unsigned char* ax = reinterpret_cast<const unsigned char*>(ep.aux) + (long)m * ep.ld_aux + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < nvalid) {
            v[r] *= gq_value((float)ax[r]);
            cs[j][r] += v[r];
          }
      } else if constexpr (epi_is_dgelu(EPI)) {
        const bf16_t* ax = reinterpret_cast<const bf16_t*>(ep.aux) + (long)m * ep.ld_aux + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < nvalid) {
            v[r] *= (EPI == MV_EPI_MUL) ? (float)ax[r] : dgelu_fast((float)ax[r]);
            cs[j][r] += v[r];
          }
      } else if constexpr (EPI == MV_EPI_EMBED) {
        const float* ax = reinterpret_cast<const float*>(ep.aux) + (long)(1 + patch) * ep.ld_aux + n;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (r < nvalid) v[r] += ax[r];
      }
      store4(C + crow * ldc + n, v, vec, nvalid);
    }
  }
  if constexpr (epi_is_dgelu(EPI)) {
    if (ep.out2 && m0 + wm * 64 < M)
      nt_colsum_flush(cs, reinterpret_cast<float*>(ep.out2), ep.ld_out2, (m0 + wm * 64) >> 6, n0 + wn * 64 + 4 * (lane >> 4), N, lane);
  }
}

// ------------------------------------------------------------------------------------------------
// NT kernel
// ------------------------------------------------------------------------------------------------
template <int EPI, typename CT>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(const bf16_t* __restrict__ A, int lda,
                                                         const bf16_t* __restrict__ B, int ldb, CT* __restrict__ C,
                                                         int ldc, int M, int N, int K, int tiles_n, EpiArgs ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;

  // staging map: 8 lanes cover one 128-byte row (8 chunks of 16 B), 32 rows per pass, 4 passes per tile.
  // Loads are UNCONDITIONAL from clamped (always valid) addresses and zeroed by a select afterwards: a
  // conditional load makes hipcc branch around it and drain vmcnt per element (cdna guide, trap (c)).
  const int sc = tid & 7, sr = tid >> 3;
  u32x4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const bf16_t* pa[4];
  const bf16_t* pb[4];
  bool va[4], vb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int ar = m0 + sr + 32 * i, br = n0 + sr + 32 * i;
    va[i] = ar < M;
    vb[i] = br < N;
    pa[i] = A + (long)(va[i] ? ar : M - 1) * lda + sc * 8;
    pb[i] = B + (long)(vb[i] ? br : N - 1) * ldb + sc * 8;
  }
#define NT_LOAD_TILE(kt_)                                                                 \
  {                                                                                       \
    const int k_ = (kt_) * BK;                                                            \
    const bool kin_ = (k_ + sc * 8) < K;                                                  \
    const int ko_ = kin_ ? k_ : -(sc * 8);                                                \
    ra0 = *reinterpret_cast<const u32x4*>(pa[0] + ko_);                                   \
    ra1 = *reinterpret_cast<const u32x4*>(pa[1] + ko_);                                   \
    ra2 = *reinterpret_cast<const u32x4*>(pa[2] + ko_);                                   \
    ra3 = *reinterpret_cast<const u32x4*>(pa[3] + ko_);                                   \
    rb0 = *reinterpret_cast<const u32x4*>(pb[0] + ko_);                                   \
    rb1 = *reinterpret_cast<const u32x4*>(pb[1] + ko_);                                   \
    rb2 = *reinterpret_cast<const u32x4*>(pb[2] + ko_);                                   \
    rb3 = *reinterpret_cast<const u32x4*>(pb[3] + ko_);                                   \
    ra0 = (kin_ && va[0]) ? ra0 : zero4; ra1 = (kin_ && va[1]) ? ra1 : zero4;             \
    ra2 = (kin_ && va[2]) ? ra2 : zero4; ra3 = (kin_ && va[3]) ? ra3 : zero4;             \
    rb0 = (kin_ && vb[0]) ? rb0 : zero4; rb1 = (kin_ && vb[1]) ? rb1 : zero4;             \
    rb2 = (kin_ && vb[2]) ? rb2 : zero4; rb3 = (kin_ && vb[3]) ? rb3 : zero4;             \
  }
#define NT_STORE_TILE(stage_)                                                             \
  {                                                                                       \
    char* sa_ = smem + (stage_) * STAGE_BYTES;                                            \
    char* sb_ = sa_ + BM * BK * 2;                                                        \
    *reinterpret_cast<u32x4*>(sa_ + sw128(sr, sc)) = ra0;                                 \
    *reinterpret_cast<u32x4*>(sa_ + sw128(sr + 32, sc)) = ra1;                            \
    *reinterpret_cast<u32x4*>(sa_ + sw128(sr + 64, sc)) = ra2;                            \
    *reinterpret_cast<u32x4*>(sa_ + sw128(sr + 96, sc)) = ra3;                            \
    *reinterpret_cast<u32x4*>(sb_ + sw128(sr, sc)) = rb0;                                 \
    *reinterpret_cast<u32x4*>(sb_ + sw128(sr + 32, sc)) = rb1;                            \
    *reinterpret_cast<u32x4*>(sb_ + sw128(sr + 64, sc)) = rb2;                            \
    *reinterpret_cast<u32x4*>(sb_ + sw128(sr + 96, sc)) = rb3;                            \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (K + BK - 1) / BK;
  NT_LOAD_TILE(0)
  NT_STORE_TILE(0)
  if (nk > 1) NT_LOAD_TILE(1)
  __syncthreads();

  const int frow = lane & 15, fch = lane >> 4;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const char* sa = smem + cur * STAGE_BYTES;
    const char* sb = sa + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(sa + sw128(wm * 64 + i * 16 + frow, 4 * ks + fch));
        bfr[i] = *reinterpret_cast<const bf16x8*>(sb + sw128(wn * 64 + i * 16 + frow, 4 * ks + fch));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) NT_STORE_TILE(cur ^ 1)
    if (kt + 2 < nk) NT_LOAD_TILE(kt + 2)
    __syncthreads();
  }
#undef NT_LOAD_TILE
#undef NT_STORE_TILE

  nt_epilogue<EPI, CT>(acc, C, ldc, M, N, m0, n0, wm, wn, lane, ep);
}

// ------------------------------------------------------------------------------------------------
// NT kernel, direct-to-LDS staging (K % 64 == 0)
// ------------------------------------------------------------------------------------------------
// glds16 (LDS-DMA, 16 B per lane): mv_common.h

template <int EPI, typename CT, bool BUFDMA = true>
__global__ __launch_bounds__(256, 2) void gemm_nt_glds_kernel(const bf16_t* __restrict__ A, int lda,
                                                              const bf16_t* __restrict__ B, int ldb, CT* __restrict__ C,
                                                              int ldc, int M, int N, int K, int tiles_n, EpiArgs ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;

  // Wave w stages rows [32w, 32w+32) of both tiles: 4 wave-instructions of 8 rows x 128 B each.  Lane L lands at
  // LDS (row 8i + L/8, physical chunk L%8) and therefore fetches LOGICAL chunk (L%8) ^ swz(row) of that row.
  // Rows beyond M / N are clamped to the last valid row (their products only reach outputs that are never stored).
  // BUFDMA (round 3; the dispatcher checks that the operands' byte extents fit 32 bits): buffer_load ... lds with a wave-uniform
  // descriptor, the lane's 32-bit byte offset and the K offset in an SGPR, as in gemm_nt_8phase_kernel; else 64-bit lane pointers
  unsigned pa[4], pb[4];                       // BUFDMA: element offsets from A / B
  const bf16_t* qa[4];                         // else: lane pointers
  const bf16_t* qb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 32 * wave + 8 * i + (lane >> 3);
    const int ch = (lane & 7) ^ (((row >> 1) & 3) << 1);
    const int ar = m0 + row < M ? m0 + row : M - 1, br = n0 + row < N ? n0 + row : N - 1;
    pa[i] = (unsigned)ar * (unsigned)lda + ch * 8;
    pb[i] = (unsigned)br * (unsigned)ldb + ch * 8;
    qa[i] = A + (long)ar * lda + ch * 8;
    qb[i] = B + (long)br * ldb + ch * 8;
  }
  char* const wave_lds = smem + 32 * __builtin_amdgcn_readfirstlane(wave) * 128;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), (short)0, (int)((unsigned)M * (unsigned)lda * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), (short)0, (int)((unsigned)N * (unsigned)ldb * 2u), 0x00020000);
#define NT_ISSUE(stage_, kt_)                                                          \
  {                                                                                    \
    char* la_ = wave_lds + (stage_) * STAGE_BYTES;                                     \
    char* lb_ = la_ + BM * BK * 2;                                                     \
    if constexpr (BUFDMA) {                                                            \
      const int so_ = (kt_) * BK * 2;                                                  \
      _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                               \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(void, la_ + 1024 * i_), 16, (int)(pa[i_] * 2u), so_, 0, 0); \
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(void, lb_ + 1024 * i_), 16, (int)(pb[i_] * 2u), so_, 0, 0); \
      }                                                                                \
    } else {                                                                           \
      const int ko_ = (kt_) * BK;                                                      \
      glds16(qa[0] + ko_, la_);          glds16(qa[1] + ko_, la_ + 1024);              \
      glds16(qa[2] + ko_, la_ + 2048);   glds16(qa[3] + ko_, la_ + 3072);              \
      glds16(qb[0] + ko_, lb_);          glds16(qb[1] + ko_, lb_ + 1024);              \
      glds16(qb[2] + ko_, lb_ + 2048);   glds16(qb[3] + ko_, lb_ + 3072);              \
    }                                                                                  \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / BK;
  const int frow = lane & 15, fch = lane >> 4;
  NT_ISSUE(0, 0)
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      NT_ISSUE(cur ^ 1, kt + 1)                                  // stays in flight across this step's MFMAs
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");         // all but the 8 youngest: this step's tile has landed
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();                                // every wave's share of the tile is visible
    asm volatile("" ::: "memory");
    const char* sa = smem + cur * STAGE_BYTES;
    const char* sb = sa + BM * BK * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        af[i] = *reinterpret_cast<const bf16x8*>(sa + sw128(wm * 64 + i * 16 + frow, 4 * ks + fch));
        bfr[i] = *reinterpret_cast<const bf16x8*>(sb + sw128(wn * 64 + i * 16 + frow, 4 * ks + fch));
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");           // this wave's LDS reads have returned ...
    __builtin_amdgcn_s_barrier();                                // ... before any wave's next DMA overwrites the stage
  }
#undef NT_ISSUE
  nt_epilogue<EPI, CT>(acc, C, ldc, M, N, m0, n0, wm, wn, lane, ep);
}

// 256 x 256 tiles (the ring and 8-phase kernels).  The first kernel of this tile size (two 64 KiB stages) is retired:
// csrc/diag/gemm_nt_glds256.inc.
constexpr int BM2 = 256, BN2 = 256;

// ------------------------------------------------------------------------------------------------
// NT kernel, 256x256 tile, LDS RING of S stages of BK = 32 (K % 32 == 0)
// ------------------------------------------------------------------------------------------------
// Ablation of the 2-stage 256x256 kernel (K = 3072): compute alone 1130 TFLOP/s-equivalent, DMA alone 1206, both 782:
// with ONE 64 KiB stage in flight the DMA side is latency-bound (64 KiB per ~2,700 cycles = 24 B/clk/CU) and overlaps
// the MFMA side poorly.  Here the same LDS holds a ring of S stages of 32 KiB (A 256x32 + B 256x32): S-1 stages
// (96-128 KiB) are in flight while one is consumed, retired by a counted s_waitcnt vmcnt(4 (S-2)), and ONE barrier per
// stage both publishes stage t and frees stage t-1 for the next DMA.  Rows are 64 bytes here; the conflict-free
// swizzle for ds_read_b128 fragment reads of 64-byte rows is chunk ^= ((row>>3)&1)<<1 (tools/lds_bank_sim.py).
constexpr int BKR = 32;
constexpr int RSTAGE_BYTES = (BM2 + BN2) * BKR * 2;  // 32 KiB

template <int EPI, typename CT, int S>
__global__ __launch_bounds__(512, 2) void gemm_nt_ring_kernel(const bf16_t* __restrict__ A, int lda,
                                                              const bf16_t* __restrict__ B, int ldb, CT* __restrict__ C,
                                                              int ldc, int M, int N, int K, int tiles_n, EpiArgs ep) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int t = xcd_remap(blockIdx.x, gridDim.x);
  const int m0 = (t / tiles_n) * BM2, n0 = (t % tiles_n) * BN2;

  // wave w stages rows [32w, 32w+32) of both tiles: 2 + 2 wave-instructions (16 rows x 64 B each) per stage
  const bf16_t* pa[2];
  const bf16_t* pb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 32 * wave + 16 * i + (lane >> 2);
    const int ch = (lane & 3) ^ (((row >> 3) & 1) << 1);
    const int ar = m0 + row < M ? m0 + row : M - 1, br = n0 + row < N ? n0 + row : N - 1;
    pa[i] = A + (long)ar * lda + ch * 8;
    pb[i] = B + (long)br * ldb + ch * 8;
  }
  char* const wave_lds = smem + 32 * wave * 64;
#define RING_ISSUE(slot_, kt_)                                                         \
  {                                                                                    \
    char* la_ = wave_lds + (slot_) * RSTAGE_BYTES;                                     \
    char* lb_ = la_ + BM2 * BKR * 2;                                                   \
    const int ko_ = (kt_) * BKR;                                                       \
    glds16(pa[0] + ko_, la_);   glds16(pa[1] + ko_, la_ + 1024);                       \
    glds16(pb[0] + ko_, lb_);   glds16(pb[1] + ko_, lb_ + 1024);                       \
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = K / BKR;
  // fragment read offset of this lane inside a 16-row x 64-byte block (constant: the swizzle bit is (row>>3)&1)
  const int frag_off = (lane & 15) * 64 + (((lane >> 4) ^ (((lane >> 3) & 1) << 1)) << 4);
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) RING_ISSUE(s, s)
  int slot = 0, fill = S - 1;                  // slot of stage kt; slot that the next DMA goes to
#define RING_SYNC(kt_)                                                                                     \
  {                                                                                                        \
    if ((kt_) + S - 2 < nk) {                                                                              \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (S - 2)) : "memory"); /* stage kt landed */            \
    } else {                                                                                               \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    }                                                                                                      \
    __builtin_amdgcn_s_barrier(); /* stage kt visible to all; every wave is done READING stage kt-1 */     \
    asm volatile("" ::: "memory");                                                                         \
  }
#define RING_REFILL(kt_)                                                                                   \
  {                                                                                                        \
    if ((kt_) + S - 1 < nk) RING_ISSUE(fill, (kt_) + S - 1) /* refill the slot stage kt-1 lived in */      \
  }
#define RING_READ()                                                                                        \
  {                                                                                                        \
    const char* sa_ = smem + slot * RSTAGE_BYTES + wm * 128 * 64 + frag_off;                               \
    const char* sb_ = smem + slot * RSTAGE_BYTES + BM2 * BKR * 2 + wn * 64 * 64 + frag_off;                \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sb_ + j * 1024); \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) af[i] = *reinterpret_cast<const bf16x8*>(sa_ + i * 1024);  \
  }
#define RING_MFMA()                                                                                        \
  {                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                          \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);            \
  }
#define RING_ADVANCE()                                                                                     \
  {                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); /* this stage's reads returned before its slot is refilled */ \
    slot = (slot + 1 == S) ? 0 : slot + 1;                                                                 \
    fill = (fill + 1 == S) ? 0 : fill + 1;                                                                 \
  }
  bf16x8 af[8], bfr[4];
  // The two waves that share a SIMD (w and w+4) would otherwise move in lockstep between barriers: both read LDS
  // (matrix pipe idle), then both issue MFMAs.  Waves 4-7 therefore run ONE STAGE BEHIND in their MFMAs: inside
  // barrier interval t they first issue the MFMAs of stage t-1 (fragments kept in registers across the barrier),
  // then read stage t -- while waves 0-3 read stage t first and then issue its MFMAs.  Reads of one group overlap
  // MFMAs of the other; results are unchanged (same products, same order per accumulator).
  if (__builtin_amdgcn_readfirstlane(wave) < 4) {
    for (int kt = 0; kt < nk; ++kt) {
      RING_SYNC(kt)
      RING_REFILL(kt)              // DMA issue (~60-185 cycles per piece of in-order issue time) overlaps the partners' MFMAs
      RING_READ()
      RING_MFMA()
      RING_ADVANCE()
    }
  } else {
    for (int kt = 0; kt < nk; ++kt) {
      RING_SYNC(kt)
      if (kt > 0) RING_MFMA()      // matrix work first: starts right at the barrier, while waves 0-3 issue DMA and read
      RING_REFILL(kt)              // ... and this group's DMA issue lands in waves 0-3's MFMA phase
      RING_READ()                  // (measured: NT prefers DMA-then-read, TN read-then-DMA; +8-20 % over issuing DMA first in both groups)
      RING_ADVANCE()
    }
    if (nk > 0) RING_MFMA()
  }
#undef RING_REFILL
#undef RING_SYNC
#undef RING_READ
#undef RING_MFMA
#undef RING_ADVANCE
#undef RING_ISSUE
#pragma unroll
  for (int h = 0; h < 2; ++h)
    nt_epilogue<EPI, CT>(*reinterpret_cast<f32x4(*)[4][4]>(&acc[4 * h]), C, ldc, M, N, m0 + 128 * wm, n0 + 128 * (wn >> 1), h,
                         wn & 1, lane, ep);
}

// ------------------------------------------------------------------------------------------------
// NT kernel, 256x256 tile, BK = 64, EIGHT PHASES per two K-tiles (K % 128 == 0)
// ------------------------------------------------------------------------------------------------
// Structure after the "256^2 8-phase template" of cdna_hip_programming.md section 5 (its example source is not part of
// this image; the schedule below is derived from the description and checked hazard by hazard):
//  * 8 waves = 2 (M) x 4 (N), 128x64 outputs per wave, as the ring kernel -> same epilogue.
//  * a phase = 16 MFMAs = one 64x32 quadrant of the wave's outputs over one K-tile of 64.  Quadrant order per K-tile:
//    (qm0,qn0) (qm0,qn1) (qm1,qn1) (qm1,qn0): phase 1 reads B_q0 + A_q0 fragments, phase 2 B_q1, phase 3 A_q1, phase 4
//    nothing (B_q0 is still in registers).
//  * LDS = 2 buffers x 4 slots of 16 KiB.  A slot is organised BY USE, not by wave: A_qm holds, for both wave rows, the
//    64 rows of quadrant-row qm (128 rows x 64 k); B_qn holds the 32 columns of quadrant-column qn of all four wave
//    columns.  Every slot is therefore read in exactly ONE phase and can be restaged 2 phases later (1 phase later for
//    B_q0, whose reads are retired by lgkmcnt(8) before that phase's first barrier).
//  * every phase all 512 threads stage ONE slot (2 global_load_lds_dwordx4 each; rows are whole 128-byte lines):
//      phase: 1        2        3        4        5        6        7        8
//      slot : A_q1[t+1] B_q0[t+2] A_q0[t+2] B_q1[t+2] A_q1[t+2] B_q0[t+3] A_q0[t+3] B_q1[t+3]
//    Round 1 waited only twice: vmcnt(6) in phase 4 (everything up to phase 1 landed = K-tile t+1 complete, read in phases
//    5-7) and in phase 8 (K-tile t+2 complete): three slots (48 KiB) in flight across those waits.  Round 3 waits for each
//    slot in the phase before the one that reads it (vmcnt(10) in phases 1, 2, 4, 5, 6, 8: five slots in flight; see P8_KTILE).
//  * a phase is  [ds_reads | stage | (waits) | barrier | lgkmcnt(0) | 16 MFMA | barrier];  waves 4-7 run ONE BARRIER
//    behind waves 0-3, so the reads + DMA issue of one group overlap the MFMAs of the other on every SIMD.
//    RAW: a vmcnt wait sits before a phase's first barrier and the data is read in the NEXT phase, so both groups have
//    waited and met a barrier before either reads.  WAR: restaging 2 phases after the last read leaves >= 3 barrier
//    intervals between the lagging group's lgkmcnt(0) and the leading group's DMA issue.
//  * the last iteration is peeled: it stages only K-tile nk-1's last slot and waits vmcnt(0), so nothing is fetched
//    beyond K and nothing is outstanding at the epilogue.
constexpr int P8_SLOT = 128 * 64 * 2;   // 16 KiB
constexpr int P8_SMEM = 8 * P8_SLOT;    // 128 KiB
constexpr int P8_BQ0 = 0, P8_AQ0 = 1, P8_BQ1 = 2, P8_AQ1 = 3;   // slot order in a buffer = staging order

typedef __attribute__((ext_vector_type(4))) int i32x4;

// I8 = true: the operands are int8 (v_mfma_i32_16x16x64_i8, int32 accumulation: exact).  An int8 row of K bytes has the
// geometry of a bf16 row of K/2 elements and a lane's 16-byte fragment holds 16 consecutive k instead of 8, for BOTH
// operands alike, so staging, LDS image, swizzle and fragment reads are unchanged (the caller passes K/2, lda/2, ldb/2);
// only the MFMA differs, and twice the MACs ride on every staged byte.  The int32 sums leave as floats through the same
// epilogues, after the zero-point correction icorr[n] has been added in integer arithmetic.
// OPK = NT_F16: the operands are IEEE half (v_mfma_f32_16x16x32_f16): same 2-byte geometry as bf16, only the MFMA differs.
// Used for the forward products of the fake-quantised FP16 formats, whose operands are exactly representable in fp16.
// OPK = NT_F8C (round 4, not yet used by the model: profiles/r04_fp8_correction_study.txt): a split-operand product with its two
// CORRECTION segments on the 8-bit matrix path.  A row is [p0 as bf16 (K x 2 bytes) | Q(.) (K bytes) | Q(.) (K bytes)]: K / 64 bf16
// K-tiles, then K / 64 e4m3 K-tiles of 128 elements -- every tile the same 128-byte rows, LDS image and fragment reads; an 8-bit
// tile issues ONE v_mfma_f32_16x16x128_f8f6f4 on the two 16-byte chunks a bf16 tile gives to two 16x16x32 instructions (any k
// order serves: both operands use the same one), with the segments' common power of two in the instruction's e8m0 scale operand.
constexpr int NT_BF16 = 0, NT_I8 = 1, NT_F16 = 2, NT_F8C = 3;
typedef __attribute__((ext_vector_type(8))) int i32x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;

template <int EPI, typename CT, int OPK = NT_BF16, bool KSPLIT = false>
__global__ __launch_bounds__(512, 2) void gemm_nt_8phase_kernel(const bf16_t* __restrict__ A, int lda,
                                                                const bf16_t* __restrict__ B, int ldb, CT* __restrict__ C,
                                                                int ldc, int M, int N, int K, int tiles_n, EpiArgs ep,
                                                                int full_tiles, int item0) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr bool I8 = OPK == NT_I8;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int bid = blockIdx.x + item0;            // item0 > 0: only the items from item0 on (the half items)
  // Work items [0, full_tiles) are whole 256x256 output tiles (full_tiles = a multiple of the CU count, or all tiles).
  // The remaining "tail" tiles would occupy only part of the chip for one more full round; each is split into two
  // 128x256 HALF items (rows [0,128) and [128,256) of the tile) that run the same pipeline with quadrant-row 0 only:
  // wave (wm, wn) owns 64x64 outputs, slot A_q0 holds the item's 128 rows, slot A_q1 / phases 3-4 are idle.
  // 591 tiles on 256 CUs: 3 rounds -> 2 + ~0.6, with no extra memory traffic.
  const int nk = K >> 6;                         // even, >= 2 (dispatch)
  int t, half = -1;
  if (bid < full_tiles) {
    t = xcd_remap(bid, full_tiles);
    if (!KSPLIT && ep.band > 0) {
      // Column bands (tiles_n % band == 0, dispatch): the whole tile rows of the full region are enumerated band by band, row-major
      // inside a band, and an XCD owns a contiguous run of THAT order -- one or two bands per XCD for the whole launch, so its
      // slice of B (band x K x 512 B; 1.2 MB for three tile columns at K = 768) stays in its 4 MiB L2 while it walks down A.
      const int rows = full_tiles / tiles_n, per_band = rows * ep.band;
      if (t < rows * tiles_n) {
        const int b = t / per_band, r = t - b * per_band, rm = r / ep.band;
        t = rm * tiles_n + b * ep.band + (r - rm * ep.band);
      }
    }
  } else {
    const int idx = bid - full_tiles;
    t = full_tiles + (idx >> 1);
    half = idx & 1;
  }
  const bool is_half_rt = half >= 0;             // wave-uniform
  if constexpr (KSPLIT) {                        // K = the slice length; the bias rides on slice 0 only
    const int ks = t / ep.ks_tiles;
    t -= ks * ep.ks_tiles;
    A += (long)ks * ep.ks_len;
    B += (long)ks * ep.ks_len;
    C += (long)ks * ep.ks_slab;
    if (ks) ep.bias = nullptr;
  }
  // Whole tiles and half items run the same text with ``is_half`` a COMPILE-TIME constant (a generic lambda instantiated twice,
  // selected once per workgroup): as a run-time flag it put twelve scalar branches into every two K-tiles of the main loop --
  // around the A_q1 reads, the phase-3/4 MFMA clusters, the staging of A_q1 and the per-slot waits -- each a basic-block boundary
  // the scheduler could not move reads or waits across.
  auto body = [&](auto half_tag) {
  constexpr bool is_half = decltype(half_tag)::value;
  const int m0 = (t / tiles_n) * BM2 + (is_half ? 128 * half : 0), n0 = (t % tiles_n) * BN2;

  // staging: wave w moves pieces 2w and 2w+1 of a slot (1 KiB = 8 slot rows of 128 B each); lane l -> row l>>3, 16-byte
  // chunk l&7 of the LDS image, i.e. logical chunk (l&7) ^ swz(row) of the source row (swizzle on the SOURCE side)
  unsigned ps[4][2];                          // element offsets from A / B (dispatch guarantees they fit 32 bits)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int rloc = 16 * wave + 8 * i + (lane >> 3);
    const int ch = (lane & 7) ^ (((rloc >> 1) & 3) << 1);
    // full tile: slot row 64 wm' + rr <-> tile row 128 wm' + 64 qm + rr; half item: <-> item row 64 wm' + rr
    const int ra = m0 + (is_half ? 64 : 128) * (wave >> 2) + 16 * (wave & 3) + 8 * i + (lane >> 3);     // + 64 qm
    const int rb = n0 + 64 * (wave >> 1) + 16 * (wave & 1) + 8 * i + (lane >> 3);      // + 32 qn
    const int ra0 = ra < M ? ra : M - 1, ra1 = ra + 64 < M ? ra + 64 : M - 1;
    const int rb0 = rb < N ? rb : N - 1, rb1 = rb + 32 < N ? rb + 32 : N - 1;
    ps[P8_AQ0][i] = (unsigned)ra0 * (unsigned)lda + ch * 8;
    ps[P8_AQ1][i] = (unsigned)ra1 * (unsigned)lda + ch * 8;
    ps[P8_BQ0][i] = (unsigned)rb0 * (unsigned)ldb + ch * 8;
    ps[P8_BQ1][i] = (unsigned)rb1 * (unsigned)ldb + ch * 8;
  }
  // the wave's staging destination as a provably wave-uniform value: the DMA's LDS base goes to M0 by scalar arithmetic
  char* const wave_lds = smem + __builtin_amdgcn_readfirstlane(wave) * 2048;
  // The pieces are buffer_load_dwordx4 ... offen lds -- a wave-uniform descriptor (4 SGPRs) + the lane's
  // 32-bit byte offset (constant over the K loop) + the K-tile's byte offset in an SGPR -- instead of global_load_lds_dwordx4 with
  // a 64-bit per-lane address: no address arithmetic per piece, 16 fewer address VGPRs (the kernel's 10 spilled VGPRs are gone),
  // and a cheaper request: +3...9 % on every ViT-B shape, 38.40 -> 37.67 ms in the step (round 3, in-process A/B against the
  // global_load_lds form, since removed).  The descriptor's extent is the whole operand (rows are clamped, never beyond it).
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), (short)0, (int)((unsigned)M * (unsigned)lda * 2u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), (short)0, (int)((unsigned)N * (unsigned)ldb * 2u), 0x00020000);
#define P8_STAGE(buf_, slot_, kt_)                                                      \
  {                                                                                     \
    char* l_ = wave_lds + ((buf_) * 4 + (slot_)) * P8_SLOT;                             \
    const int so_ = (kt_) * 128;                                                        \
    if ((slot_) == P8_AQ0 || (slot_) == P8_AQ1) {                                       \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(void, l_), 16, (int)(ps[slot_][0] * 2u), so_, 0, 0);        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(void, l_ + 1024), 16, (int)(ps[slot_][1] * 2u), so_, 0, 0); \
    } else {                                                                            \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(void, l_), 16, (int)(ps[slot_][0] * 2u), so_, 0, 0);        \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(void, l_ + 1024), 16, (int)(ps[slot_][1] * 2u), so_, 0, 0); \
    }                                                                                   \
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  // fragment reads: slot rows 64 wm + 16 mt + (lane&15) (A) / 32 wn + 16 nt + (lane&15) (B); all bases are multiples
  // of 16 rows, so the sw128 term depends on the lane only
  const int fr = (((lane & 15) >> 1) & 3) << 1;
  const int rf0 = (lane & 15) * 128 + (((lane >> 4) ^ fr) << 4);
  const int rf1 = (lane & 15) * 128 + (((4 + (lane >> 4)) ^ fr) << 4);
  const char* const a_rd = smem + 64 * wm * 128;
  const char* const b_rd = smem + 32 * wn * 128;
  bf16x8 af[4][2], bf0[2][2], bf1[2][2];
  // NT_F8C keeps each fragment PAIR as one 8-dword value (the 8-bit instruction's operand: eight consecutive registers); its bf16
  // tiles take the halves as sub-registers.  (Built at the MFMA from two 4-dword values instead, the pairs cost 50-100 spilled dwords.)
  [[maybe_unused]] i32x8 af_8[4], bf0_8[2], bf1_8[2];
#define P8_PAIR(p0_, p1_)                                                                                    \
  __builtin_shufflevector(*reinterpret_cast<const i32x4*>(p0_), *reinterpret_cast<const i32x4*>(p1_), 0, 1, 2, 3, 4, 5, 6, 7)
#define P8_READ_A(buf_, slot_)                                                                               \
  {                                                                                                          \
    const char* s_ = a_rd + ((buf_) * 4 + (slot_)) * P8_SLOT;                                                \
    _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                                          \
      if constexpr (OPK == NT_F8C) {                                                                         \
        af_8[i] = P8_PAIR(s_ + i * 2048 + rf0, s_ + i * 2048 + rf1);                                         \
      } else {                                                                                               \
        af[i][0] = *reinterpret_cast<const bf16x8*>(s_ + i * 2048 + rf0);                                    \
        af[i][1] = *reinterpret_cast<const bf16x8*>(s_ + i * 2048 + rf1);                                    \
      }                                                                                                      \
    }                                                                                                        \
  }
#define P8_READ_B(dst_, buf_, slot_)                                                                         \
  {                                                                                                          \
    const char* s_ = b_rd + ((buf_) * 4 + (slot_)) * P8_SLOT;                                                \
    _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                          \
      if constexpr (OPK == NT_F8C) {                                                                         \
        dst_##_8[j] = P8_PAIR(s_ + j * 2048 + rf0, s_ + j * 2048 + rf1);                                     \
      } else {                                                                                               \
        dst_[j][0] = *reinterpret_cast<const bf16x8*>(s_ + j * 2048 + rf0);                                  \
        dst_[j][1] = *reinterpret_cast<const bf16x8*>(s_ + j * 2048 + rf1);                                  \
      }                                                                                                      \
    }                                                                                                        \
  }
#define P8_MFMA(mb_, nb_, bfx_, M8_)                                                                         \
  {                                                                                                          \
    __builtin_amdgcn_s_setprio(1);                                                                           \
    if constexpr (OPK == NT_F8C && (M8_)) {                                                                  \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j)                                                        \
          acc[(mb_) + i][(nb_) + j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(                      \
              bfx_##_8[j], af_8[i], acc[(mb_) + i][(nb_) + j], 0, 0, 0, 0x7F7F7F7F, 0, f8_sc);               \
    } else if constexpr (OPK == NT_F8C) {                                                                    \
      _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                       \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
          _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                    \
            const i32x4 a4 = ks ? __builtin_shufflevector(af_8[i], af_8[i], 4, 5, 6, 7) : __builtin_shufflevector(af_8[i], af_8[i], 0, 1, 2, 3);                 \
            const i32x4 b4 = ks ? __builtin_shufflevector(bfx_##_8[j], bfx_##_8[j], 4, 5, 6, 7) : __builtin_shufflevector(bfx_##_8[j], bfx_##_8[j], 0, 1, 2, 3); \
            acc[(mb_) + i][(nb_) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                             \
                __builtin_bit_cast(bf16x8, b4), __builtin_bit_cast(bf16x8, a4), acc[(mb_) + i][(nb_) + j], 0, 0, 0);   \
          }                                                                                                  \
    } else                                                                                                   \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                         \
      _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                          \
        _Pragma("unroll") for (int j = 0; j < 2; ++j) {                                                      \
          if constexpr (I8)                                                                                  \
            acc[(mb_) + i][(nb_) + j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_mfma_i32_16x16x64_i8(     \
                __builtin_bit_cast(i32x4, bfx_[j][ks]), __builtin_bit_cast(i32x4, af[i][ks]),                \
                __builtin_bit_cast(i32x4, acc[(mb_) + i][(nb_) + j]), 0, 0, 0));                             \
          else if constexpr (OPK == NT_F16)                                                                  \
            acc[(mb_) + i][(nb_) + j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(                              \
                __builtin_bit_cast(f16x8, bfx_[j][ks]), __builtin_bit_cast(f16x8, af[i][ks]),                \
                acc[(mb_) + i][(nb_) + j], 0, 0, 0);                                                         \
          else                                                                                               \
            acc[(mb_) + i][(nb_) + j] =                                                                      \
                __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfx_[j][ks], af[i][ks], acc[(mb_) + i][(nb_) + j], 0, 0, 0); \
        }                                                                                                    \
    __builtin_amdgcn_s_setprio(0);                                                                           \
  }
#define P8_BAR()                                  \
  {                                               \
    __builtin_amdgcn_sched_barrier(0);            \
    __builtin_amdgcn_s_barrier();                 \
    asm volatile("" ::: "memory");                \
    __builtin_amdgcn_sched_barrier(0);            \
  }
#define P8_LGKM0()                                          \
  {                                                         \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      \
    __builtin_amdgcn_sched_barrier(0);                      \
  }
// one K-tile (4 phases) out of buffer d_; ST_ = 1: stage per the schedule (k-tile indices t1_, t2_ for the two buffers'
// next contents), ST_ = 0: the drained tail.  WAIT_: asm string of the phase-4/8 wait.
// Waits.  The round-1 schedule waited twice per eight phases ("vmcnt(6)" in phases 4 and 8: the whole NEXT K-tile landed), which
// gives the slot staged last before the wait (A_q1, staged in phase 1/5) only THREE phases -- about 1 us at the kernel's pace -- to
// arrive, although it is first read three phases after the wait.  The throughput of a latency-bound L2 -> LDS stream is bytes in
// flight / latency (the kernel moves ~18 B/clk/CU with 48 KiB in flight and ~1.5 us of loaded latency), so the waits are now placed
// where the data is needed: each slot is waited for in the phase BEFORE the one that reads it (the RAW rule of the header: wait
// before a phase's first barrier, read in the next phase) and not earlier --
//      phase 4 / 8: B_q0, A_q0 of the next K-tile (read in phase 5 / 1)      phase 1 / 5: B_q1 of this K-tile (read in phase 2 / 6)
//      phase 2 / 6: A_q1 of this K-tile (read in phase 3 / 7)                phase 3 / 7: nothing
// In the steady state exactly five slots (10 DMA instructions) have been issued after the slot(s) being waited for at every one of
// those points, so every wait is vmcnt(10): FIVE slots (80 KiB) stay in flight across the waits instead of three, with the same
// staging times (the WAR argument is unchanged) and the same 128 KiB of LDS.  W1_/W2_/W4_ are the counts of a full tile; half items
// (no A_q1 slot, three slots per K-tile) keep the round-1 waits (H4_).  (Measured against the round-1 placement with a run-time
// switch, since removed: +0...3.8 % per shape, 37.16 vs 37.27 ms in the step.)
#define P8_WAIT(N_) asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory");
#define P8_KTILE(d_, ST_, first_slot_buf_, first_slot_kt_, next_buf_, next_kt_, W1_, W2_, W4_, H4_, STAGE_FIRST_, M8_)   \
  {                                                                                                          \
    /* phase 1/5 */                                                                                          \
    P8_READ_B(bf0, d_, P8_BQ0)                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    P8_READ_A(d_, P8_AQ0)                                                                                    \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if (STAGE_FIRST_ && !is_half) P8_STAGE(first_slot_buf_, P8_AQ1, first_slot_kt_)                          \
    asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory"); /* the 4 B_q0 reads (issued first) have returned */   \
    if (lazy) { W1_ }                                                                                        \
    P8_BAR()                                                                                                 \
    P8_LGKM0()                                                                                               \
    P8_MFMA(0, 0, bf0, M8_)                                                                                       \
    P8_BAR()                                                                                                 \
    /* phase 2/6 */                                                                                          \
    P8_READ_B(bf1, d_, P8_BQ1)                                                                               \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if (ST_) P8_STAGE(next_buf_, P8_BQ0, next_kt_)                                                           \
    if (lazy) { W2_ }                                                                                        \
    P8_BAR()                                                                                                 \
    P8_LGKM0()                                                                                               \
    P8_MFMA(0, 2, bf1, M8_)                                                                                       \
    P8_BAR()                                                                                                 \
    /* phase 3/7 */                                                                                          \
    if (!is_half) P8_READ_A(d_, P8_AQ1)                                                                      \
    __builtin_amdgcn_sched_barrier(0);                                                                       \
    if (ST_) P8_STAGE(next_buf_, P8_AQ0, next_kt_)                                                           \
    P8_BAR()                                                                                                 \
    P8_LGKM0()                                                                                               \
    if (!is_half) P8_MFMA(4, 2, bf1, M8_)                                                                         \
    P8_BAR()                                                                                                 \
    /* phase 4/8 */                                                                                          \
    if (ST_) P8_STAGE(next_buf_, P8_BQ1, next_kt_)                                                           \
    if (lazy) { W4_ } else { H4_ }                                                                           \
    P8_BAR()                                                                                                 \
    if (!is_half) P8_MFMA(4, 0, bf0, M8_)                                                                         \
    P8_BAR()                                                                                                 \
  }

  constexpr bool lazy = !is_half;
  P8_STAGE(0, P8_BQ0, 0) P8_STAGE(0, P8_AQ0, 0) P8_STAGE(0, P8_BQ1, 0) if (!is_half) P8_STAGE(0, P8_AQ1, 0)
  P8_STAGE(1, P8_BQ0, 1) P8_STAGE(1, P8_AQ0, 1) P8_STAGE(1, P8_BQ1, 1)
  if (lazy) { P8_WAIT(10) } else { P8_WAIT(6) }           // B_q0, A_q0 of K-tile 0 landed | the whole K-tile 0
  P8_BAR()
  if (__builtin_amdgcn_readfirstlane(wave) >= 4) P8_BAR()      // waves 4-7 run one barrier behind
  int kt = 0;
  [[maybe_unused]] const int f8_sc = (ep.f8_scale & 255) * 0x01010101;          // NT_F8C: the e8m0 scale byte, in every position
  if constexpr (OPK == NT_F8C) {
    // the same loop twice, once per MFMA type (compile-time, no branch inside): the bf16 tiles [0, f8_tiles16), then the 8-bit tiles;
    // staging is type-blind (whole 128-byte rows), so the pipeline runs across the boundary.  f8_tiles16 even, 2 <= f8_tiles16 <= nk - 2.
    for (; kt < ep.f8_tiles16; kt += 2) {
      P8_KTILE(0, 1, 1, kt + 1, 0, kt + 2, P8_WAIT(10), P8_WAIT(10), P8_WAIT(10), P8_WAIT(6), 1, 0)
      P8_KTILE(1, 1, 0, kt + 2, 1, kt + 3, P8_WAIT(10), P8_WAIT(10), P8_WAIT(10), P8_WAIT(6), 1, 0)
    }
    for (; kt < nk - 2; kt += 2) {
      P8_KTILE(0, 1, 1, kt + 1, 0, kt + 2, P8_WAIT(10), P8_WAIT(10), P8_WAIT(10), P8_WAIT(6), 1, 1)
      P8_KTILE(1, 1, 0, kt + 2, 1, kt + 3, P8_WAIT(10), P8_WAIT(10), P8_WAIT(10), P8_WAIT(6), 1, 1)
    }
    P8_KTILE(0, 0, 1, kt + 1, 0, 0, P8_WAIT(10), P8_WAIT(8), P8_WAIT(4), P8_WAIT(0), 1, 1)
    P8_KTILE(1, 0, 0, 0, 0, 0, P8_WAIT(2), P8_WAIT(0), , , 0, 1)
  } else {
  for (; kt < nk - 2; kt += 2) {
    // phases 1-4: compute buffer 0 (K-tile kt); stage A_q1[kt+1] -> buffer 1, then B_q0/A_q0/B_q1[kt+2] -> buffer 0
    P8_KTILE(0, 1, 1, kt + 1, 0, kt + 2, P8_WAIT(10), P8_WAIT(10), P8_WAIT(10), P8_WAIT(6), 1, 0)
    // phases 5-8: compute buffer 1 (K-tile kt+1); stage A_q1[kt+2] -> buffer 0, then B_q0/A_q0/B_q1[kt+3] -> buffer 1
    P8_KTILE(1, 1, 0, kt + 2, 1, kt + 3, P8_WAIT(10), P8_WAIT(10), P8_WAIT(10), P8_WAIT(6), 1, 0)
  }
  // peeled last iteration: only K-tile nk-1's A_q1 is still to be staged; the queue drains 10 -> 8 -> 4 -> 2 -> 0
  P8_KTILE(0, 0, 1, kt + 1, 0, 0, P8_WAIT(10), P8_WAIT(8), P8_WAIT(4), P8_WAIT(0), 1, 0)
  P8_KTILE(1, 0, 0, 0, 0, 0, P8_WAIT(2), P8_WAIT(0), , , 0, 0)
  }
  if (__builtin_amdgcn_readfirstlane(wave) < 4) P8_BAR()
#undef P8_WAIT
#undef P8_KTILE
#undef P8_LGKM0
#undef P8_BAR
#undef P8_MFMA
#undef P8_READ_A
#undef P8_PAIR
#undef P8_READ_B
#undef P8_STAGE
  // full tile: wave (wm, wn) finishes rows 128 wm + 64 h (h = 0, 1); half item: its only quadrant-row, rows 64 wm.
  // The epilogues that hoist a whole aux tile into registers (DGELU / MUL: 32, RESIDUAL: 64 VGPRs) cannot also hold both
  // 64-register accumulator halves: the second half waits in LDS (free now: every wave has passed the final barrier
  // after its last fragment read, and no DMA is outstanding) -- a 16 x ds_write_b128 / ds_read_b128 round trip per lane
  // instead of 60-170 registers spilled to scratch memory.
  if constexpr (I8) {
    // exact int32 dot products (+ the per-column zero-point correction, still in integer arithmetic) -> float
    const int nc = n0 + 128 * (wn >> 1) + 64 * (wn & 1) + 4 * (lane >> 4);          // + 16 j + r: this lane's columns
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      int c[4] = {0, 0, 0, 0};
      if (ep.icorr) {
#pragma unroll
        for (int r = 0; r < 4; ++r) c[r] = (nc + 16 * j + r < N) ? ep.icorr[nc + 16 * j + r] : 0;
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        // (bit-cast the WHOLE vector: __builtin_bit_cast of a single vector element reads element 0's storage)
        const i32x4 iv = __builtin_bit_cast(i32x4, acc[i][j]);
        acc[i][j] = (f32x4){(float)(iv[0] + c[0]), (float)(iv[1] + c[1]), (float)(iv[2] + c[2]), (float)(iv[3] + c[3])};
      }
    }
  }
  constexpr bool park = epi_is_dgelu(EPI) || EPI == MV_EPI_RESIDUAL || EPI == MV_EPI_SPLIT_DGELU;
  f32x4* const parked = reinterpret_cast<f32x4*>(smem) + wave * 16 * 64 + lane;
  if (park && !is_half) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) parked[(i * 4 + j) * 64] = acc[4 + i][j];
  }
  // both quadrant-rows' aux tiles are requested now; the second one lands under the first one's arithmetic and stores
  constexpr bool can_pre = (EPI == MV_EPI_RESIDUAL || EPI == MV_EPI_MUL8) && !KSPLIT;
  if constexpr (can_pre) {
    if (!is_half && nt_epi_prefetch_ok<EPI>(M, N, m0 + 128 * wm, n0 + 128 * (wn >> 1), ldc, ep)) {
      EpiPre<EPI> pre0, pre1;
      nt_epi_prefetch<EPI>(pre0, m0 + 128 * wm, n0 + 128 * (wn >> 1), 0, wn & 1, lane, ep);
      nt_epi_prefetch<EPI>(pre1, m0 + 128 * wm, n0 + 128 * (wn >> 1), 1, wn & 1, lane, ep);
      nt_epilogue<EPI, CT, true>(*reinterpret_cast<f32x4(*)[4][4]>(&acc[0]), C, ldc, M, N, m0 + 128 * wm, n0 + 128 * (wn >> 1), 0,
                                 wn & 1, lane, ep, nullptr, pre0);
      f32x4 accb[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) accb[i][j] = parked[(i * 4 + j) * 64];        // can_pre implies park
      nt_epilogue<EPI, CT, true>(accb, C, ldc, M, N, m0 + 128 * wm, n0 + 128 * (wn >> 1), 1, wn & 1, lane, ep, nullptr, pre1);
      return;
    }
  }
  nt_epilogue<EPI, CT>(*reinterpret_cast<f32x4(*)[4][4]>(&acc[0]), C, ldc, M, N, is_half ? m0 : m0 + 128 * wm,
                       n0 + 128 * (wn >> 1), is_half ? wm : 0, wn & 1, lane, ep);
  if (is_half) return;
  f32x4 acc2[4][4];
  if (park) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc2[i][j] = parked[(i * 4 + j) * 64];
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc2[i][j] = acc[4 + i][j];
  }
  nt_epilogue<EPI, CT>(acc2, C, ldc, M, N, m0 + 128 * wm, n0 + 128 * (wn >> 1), 1, wn & 1, lane, ep);
  };
  if (is_half_rt) body(std::true_type{});
  else body(std::false_type{});
}

__device__ __forceinline__ void tn_store(f32x4 (&acc)[4][4], float* __restrict__ Cs, long ldc, int M, int N, int m0, int n0,
                                         int wm, int wn, int lane) {
  const bool ldc_vec = (ldc & 3) == 0;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + wm * 64 + i * 16 + (lane & 15);
    if (m >= M) continue;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + wn * 64 + j * 16 + 4 * (lane >> 4);
      if (n >= N) continue;
      const int nvalid = (N - n) >= 4 ? 4 : (N - n);
      const float v[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
      store4(Cs + (long)m * ldc + n, v, ldc_vec && nvalid == 4, nvalid);
    }
  }
}


// ------------------------------------------------------------------------------------------------
// TN kernel: slab[s][M][N] = sum over this split's kc of A[kc][m] * B[kc][n]
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(const bf16_t* __restrict__ A, int lda,
                                                         const bf16_t* __restrict__ B, int ldb, float* __restrict__ C,
                                                         long ldc, long slab_stride, int M, int N, int Kc, int tiles_n,
                                                         int tiles_mn, int steps_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int t_all = xcd_remap(blockIdx.x, gridDim.x);
  const int split = t_all / tiles_mn;
  const int t = t_all - split * tiles_mn;
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  const int nk_total = (Kc + BK - 1) / BK;
  const int kt0 = split * steps_per_split;
  int nk = nk_total - kt0;
  if (nk > steps_per_split) nk = steps_per_split;
  float* Cs = C + (long)split * slab_stride;

  // staging map: 16 lanes cover one 256-byte row (16 chunks), 16 rows per pass, 4 passes per tile; loads are
  // unconditional from clamped addresses, zeroed by select (see the NT kernel)
  const int sc = tid & 15, sr = tid >> 4;
  u32x4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
  const u32x4 zero4 = {0u, 0u, 0u, 0u};
  const bool a_in = (m0 + sc * 8) < M, b_in = (n0 + sc * 8) < N;
  const bf16_t* pa = A + (a_in ? m0 + sc * 8 : 0);
  const bf16_t* pb = B + (b_in ? n0 + sc * 8 : 0);
  const int kc_last = Kc - 1;
#define TN_LOAD_TILE(kt_)                                                                 \
  {                                                                                       \
    const int kc_ = (kt0 + (kt_)) * BK + sr;                                              \
    const bool k0_ = kc_ < Kc, k1_ = kc_ + 16 < Kc, k2_ = kc_ + 32 < Kc, k3_ = kc_ + 48 < Kc; \
    const long r0_ = k0_ ? kc_ : kc_last, r1_ = k1_ ? kc_ + 16 : kc_last;                 \
    const long r2_ = k2_ ? kc_ + 32 : kc_last, r3_ = k3_ ? kc_ + 48 : kc_last;            \
    ra0 = *reinterpret_cast<const u32x4*>(pa + r0_ * lda);                                \
    ra1 = *reinterpret_cast<const u32x4*>(pa + r1_ * lda);                                \
    ra2 = *reinterpret_cast<const u32x4*>(pa + r2_ * lda);                                \
    ra3 = *reinterpret_cast<const u32x4*>(pa + r3_ * lda);                                \
    rb0 = *reinterpret_cast<const u32x4*>(pb + r0_ * ldb);                                \
    rb1 = *reinterpret_cast<const u32x4*>(pb + r1_ * ldb);                                \
    rb2 = *reinterpret_cast<const u32x4*>(pb + r2_ * ldb);                                \
    rb3 = *reinterpret_cast<const u32x4*>(pb + r3_ * ldb);                                \
    ra0 = (k0_ && a_in) ? ra0 : zero4; ra1 = (k1_ && a_in) ? ra1 : zero4;                 \
    ra2 = (k2_ && a_in) ? ra2 : zero4; ra3 = (k3_ && a_in) ? ra3 : zero4;                 \
    rb0 = (k0_ && b_in) ? rb0 : zero4; rb1 = (k1_ && b_in) ? rb1 : zero4;                 \
    rb2 = (k2_ && b_in) ? rb2 : zero4; rb3 = (k3_ && b_in) ? rb3 : zero4;                 \
  }
#define TN_STORE_TILE(stage_)                                                             \
  {                                                                                       \
    char* sa_ = smem + (stage_) * STAGE_BYTES;                                            \
    char* sb_ = sa_ + BK * BM * 2;                                                        \
    *reinterpret_cast<u32x4*>(sa_ + sw256(sr, sc)) = ra0;                                 \
    *reinterpret_cast<u32x4*>(sa_ + sw256(sr + 16, sc)) = ra1;                            \
    *reinterpret_cast<u32x4*>(sa_ + sw256(sr + 32, sc)) = ra2;                            \
    *reinterpret_cast<u32x4*>(sa_ + sw256(sr + 48, sc)) = ra3;                            \
    *reinterpret_cast<u32x4*>(sb_ + sw256(sr, sc)) = rb0;                                 \
    *reinterpret_cast<u32x4*>(sb_ + sw256(sr + 16, sc)) = rb1;                            \
    *reinterpret_cast<u32x4*>(sb_ + sw256(sr + 32, sc)) = rb2;                            \
    *reinterpret_cast<u32x4*>(sb_ + sw256(sr + 48, sc)) = rb3;                            \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if (nk > 0) {
    TN_LOAD_TILE(0)
    TN_STORE_TILE(0)
    if (nk > 1) TN_LOAD_TILE(1)
  }
  __syncthreads();

  // transposed-read lane map (ds_read_b64_tr_b16): within a 16-lane group, lane 4q+p addresses row q, columns
  // 4p..4p+3 of a 4x16 block and receives column (lane&15) of its 4 rows.
  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    const char* sa = smem + cur * STAGE_BYTES;
    const char* sb = sa + BK * BM * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
      const int r0 = 32 * ks + 8 * g + q;  // fragment element e <-> kc row 32ks + 8g + e
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ca = (wm * 64 + i * 16) >> 3, cb = (wn * 64 + i * 16) >> 3;
        const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sa + sw256(r0, ca + (p >> 1)) + 8 * (p & 1)));
        const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sa + sw256(r0 + 4, ca + (p >> 1)) + 8 * (p & 1)));
        const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sb + sw256(r0, cb + (p >> 1)) + 8 * (p & 1)));
        const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sb + sw256(r0 + 4, cb + (p >> 1)) + 8 * (p & 1)));
        af[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        bfr[i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    if (kt + 1 < nk) TN_STORE_TILE(cur ^ 1)
    if (kt + 2 < nk) TN_LOAD_TILE(kt + 2)
    __syncthreads();
  }
#undef TN_LOAD_TILE
#undef TN_STORE_TILE

  tn_store(acc, Cs, ldc, M, N, m0, n0, wm, wn, lane);
}

// TN kernel, direct-to-LDS staging (Kc % 64 == 0): same product, tiles [64 kc][128 cols] = 256-byte rows; one
// wave-instruction covers 4 rows.  Columns beyond the operand's width are clamped to chunk 0 (outputs never stored).
__global__ __launch_bounds__(256, 2) void gemm_tn_glds_kernel(const bf16_t* __restrict__ A, int lda,
                                                              const bf16_t* __restrict__ B, int ldb, float* __restrict__ C,
                                                              long ldc, long slab_stride, int M, int N, int Kc, int tiles_n,
                                                              int tiles_mn, int steps_per_split) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int t_all = xcd_remap(blockIdx.x, gridDim.x);
  const int split = t_all / tiles_mn;
  const int t = t_all - split * tiles_mn;
  const int m0 = (t / tiles_n) * BM, n0 = (t % tiles_n) * BN;
  const int nk_total = Kc / BK;
  const int kt0 = split * steps_per_split;
  int nk = nk_total - kt0;
  if (nk > steps_per_split) nk = steps_per_split;
  float* Cs = C + (long)split * slab_stride;

  // wave w stages kc rows [16w, 16w+16) of both tiles: 4 wave-instructions of 4 rows x 256 B
  unsigned pa[4], pb[4];                       // BYTE offsets from A / B (buffer-descriptor DMA; extents checked by the host)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 16 * wave + 4 * i + (lane >> 4);
    const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    const int ca = (m0 + ch * 8 + 8 <= lda) ? m0 + ch * 8 : 0, cb = (n0 + ch * 8 + 8 <= ldb) ? n0 + ch * 8 : 0;
    pa[i] = 2u * ((unsigned)(kt0 * BK + row) * (unsigned)lda + (unsigned)ca);
    pb[i] = 2u * ((unsigned)(kt0 * BK + row) * (unsigned)ldb + (unsigned)cb);
  }
  char* const wave_lds = smem + 16 * __builtin_amdgcn_readfirstlane(wave) * 256;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), (short)0, (int)0xFFFFFFFCu, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), (short)0, (int)0xFFFFFFFCu, 0x00020000);
  const unsigned astep = 2u * BK * (unsigned)lda, bstep = 2u * BK * (unsigned)ldb;      // bytes per K-step
#define TN_ISSUE(stage_, kt_)                                                          \
  {                                                                                    \
    char* la_ = wave_lds + (stage_) * STAGE_BYTES;                                     \
    char* lb_ = la_ + BK * BM * 2;                                                     \
    const int ao_ = (int)((unsigned)(kt_) * astep), bo_ = (int)((unsigned)(kt_) * bstep); \
    _Pragma("unroll") for (int i_ = 0; i_ < 4; ++i_) {                                 \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(void, la_ + 1024 * i_), 16, (int)pa[i_], ao_, 0, 0); \
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(void, lb_ + 1024 * i_), 16, (int)pb[i_], bo_, 0, 0); \
    }                                                                                  \
  }

  f32x4 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  if (nk > 0) TN_ISSUE(0, 0)
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) {
      TN_ISSUE(cur ^ 1, kt + 1)
      asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    } else {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    const char* sa = smem + cur * STAGE_BYTES;
    const char* sb = sa + BK * BM * 2;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 af[4], bfr[4];
      const int r0 = 32 * ks + 8 * g + q;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ca = (wm * 64 + i * 16) >> 3, cb = (wn * 64 + i * 16) >> 3;
        const bf16x4 a0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sa + sw256(r0, ca + (p >> 1)) + 8 * (p & 1)));
        const bf16x4 a1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sa + sw256(r0 + 4, ca + (p >> 1)) + 8 * (p & 1)));
        const bf16x4 b0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sb + sw256(r0, cb + (p >> 1)) + 8 * (p & 1)));
        const bf16x4 b1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, sb + sw256(r0 + 4, cb + (p >> 1)) + 8 * (p & 1)));
        af[i] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
        bfr[i] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
#undef TN_ISSUE
  tn_store(acc, Cs, ldc, M, N, m0, n0, wm, wn, lane);
}

// ------------------------------------------------------------------------------------------------
// TN kernel, 256x256 tile, LDS ring of S stages of 32 contraction rows (Kc % 32 == 0), staggered wave groups
// ------------------------------------------------------------------------------------------------
// Same structure as gemm_nt_ring_kernel.  Each operand stage is two PANELS of [32 kc][128 cols] (256-byte rows, the
// sw256 image, transposed fragment reads conflict-free); wave (wm, wn) reads A panel wm and half of B panel wn>>1.
// SEG (the bf16x6 fp32 product, mv_gemm_tn_bf16_x6): the contraction runs over six SEGMENTS of `seg.tiles` stages each; segment
// s of A starts seg.a[s] elements into A (a column block of the side-by-side split3 layout), likewise B -- the stage
// addresses follow the table instead of one linear walk, nothing else changes.
struct TnSeg {
  int tiles;
  int a[6], b[6];
};
template <int S, bool SEG = false>
__global__ __launch_bounds__(512, 2) void gemm_tn_ring_kernel(const bf16_t* __restrict__ A, int lda,
                                                              const bf16_t* __restrict__ B, int ldb, float* __restrict__ C,
                                                              long ldc, long slab_stride, int M, int N, int Kc, int tiles_n,
                                                              int tiles_mn, int steps_per_split, TnSeg seg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3;
  const int t_all = xcd_remap(blockIdx.x, gridDim.x);
  const int split = t_all / tiles_mn;
  const int t = t_all - split * tiles_mn;
  const int m0 = (t / tiles_n) * BM2, n0 = (t % tiles_n) * BN2;
  const int nk_total = Kc / BKR;
  const int kt0 = split * steps_per_split;
  int nk = nk_total - kt0;
  if (nk > steps_per_split) nk = steps_per_split;
  float* Cs = C + (long)split * slab_stride;

  // wave w stages rows [8 (w&3), +8) of panel (w>>2) of both operands: 2 + 2 wave-instructions of 4 rows x 256 B
  // staging sources as 32-bit BYTE offsets from A / B (buffer-descriptor DMA, bufdma16_hidden): lane part here, the stage's
  // row offset in an SGPR
  unsigned pa[2], pb[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int row = 8 * (wave & 3) + 4 * i + (lane >> 4);
    const int ch = (lane & 15) ^ (((row & 3) << 2) | ((row >> 2) & 3));
    const int col = 128 * (wave >> 2) + ch * 8;
    // SEG: a segment is M (N) columns wide, the row stride covers all six
    const int wa = SEG ? ((M + 7) & ~7) : lda, wb = SEG ? ((N + 7) & ~7) : ldb;
    const int ca = (m0 + col + 8 <= wa) ? m0 + col : 0, cb = (n0 + col + 8 <= wb) ? n0 + col : 0;
    pa[i] = 2u * ((unsigned)row * (unsigned)lda + (unsigned)ca);
    pb[i] = 2u * ((unsigned)row * (unsigned)ldb + (unsigned)cb);
  }
  char* const wave_lds = smem + (wave >> 2) * 8192 + 8 * (wave & 3) * 256;
  const mv_srd_t rsA = mv_make_srd(A, 0xFFFFFFFCu), rsB = mv_make_srd(B, 0xFFFFFFFCu);   // whole-operand extents (< 4 GiB: checked by the host)
  const long astep = (long)BKR * lda, bstep = (long)BKR * ldb;
  // SEG: running (segment, stage-in-segment) of the NEXT stage to issue; stages are issued strictly in order
  int seg_s = SEG ? kt0 / seg.tiles : 0, seg_r = SEG ? kt0 - seg_s * seg.tiles : 0;
  long seg_ao = SEG ? seg.a[seg_s] + seg_r * astep : 0, seg_bo = SEG ? seg.b[seg_s] + seg_r * bstep : 0;
#define TNR_ISSUE(slot_, kt_)                                                          \
  {                                                                                    \
    char* la_ = wave_lds + (slot_) * RSTAGE_BYTES;                                     \
    char* lb_ = la_ + 16384;                                                           \
    const unsigned ao_ = 2u * (unsigned)(SEG ? seg_ao : (kt0 + (kt_)) * astep);         \
    const unsigned bo_ = 2u * (unsigned)(SEG ? seg_bo : (kt0 + (kt_)) * bstep);         \
    bufdma16_hidden(rsA, pa[0], ao_, la_);   bufdma16_hidden(rsA, pa[1], ao_, la_ + 1024);     \
    bufdma16_hidden(rsB, pb[0], bo_, lb_);   bufdma16_hidden(rsB, pb[1], bo_, lb_ + 1024);     \
    if constexpr (SEG) {                                                               \
      seg_ao += astep;                                                                 \
      seg_bo += bstep;                                                                 \
      if (++seg_r == seg.tiles) {                                                      \
        seg_r = 0;                                                                     \
        seg_s = seg_s < 5 ? seg_s + 1 : 5;                                             \
        seg_ao = seg.a[seg_s];                                                         \
        seg_bo = seg.b[seg_s];                                                         \
      }                                                                                \
    }                                                                                  \
  }

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, q = (lane >> 2) & 3, p = lane & 3;
  const int r0 = 8 * g + q;                    // fragment element e <-> kc row 8g + e of the stage
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) TNR_ISSUE(s, s)
  int slot = 0, fill = S - 1;
#define TNR_SYNC(kt_)                                                                                      \
  {                                                                                                        \
    if ((kt_) + S - 2 < nk) {                                                                              \
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(4 * (S - 2)) : "memory");                                  \
    } else {                                                                                               \
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                     \
    }                                                                                                      \
    __builtin_amdgcn_s_barrier();                                                                          \
    asm volatile("" ::: "memory");                                                                         \
  }
#define TNR_REFILL(kt_)                                                                                    \
  {                                                                                                        \
    if ((kt_) + S - 1 < nk) TNR_ISSUE(fill, (kt_) + S - 1)                                                 \
  }
#define TNR_FRAG(base_, ch_)                                                                               \
  __builtin_shufflevector(                                                                                 \
      __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, (base_) + sw256(r0, (ch_) + (p >> 1)) + 8 * (p & 1))),     \
      __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDS_PTR(bf16x4, (base_) + sw256(r0 + 4, (ch_) + (p >> 1)) + 8 * (p & 1))), \
      0, 1, 2, 3, 4, 5, 6, 7)
#define TNR_READ()                                                                                         \
  {                                                                                                        \
    const char* sa_ = smem + slot * RSTAGE_BYTES + wm * 8192;                                              \
    const char* sb_ = smem + slot * RSTAGE_BYTES + 16384 + (wn >> 1) * 8192;                               \
    _Pragma("unroll") for (int j = 0; j < 4; ++j) bfr[j] = TNR_FRAG(sb_, ((wn & 1) * 64 + j * 16) >> 3);   \
    _Pragma("unroll") for (int i = 0; i < 8; ++i) af[i] = TNR_FRAG(sa_, i * 2);                          \
  }
#define TNR_MFMA()                                                                                         \
  {                                                                                                        \
    _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                          \
      _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[j], af[i], acc[i][j], 0, 0, 0);            \
  }
#define TNR_ADVANCE()                                                                                      \
  {                                                                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                     \
    slot = (slot + 1 == S) ? 0 : slot + 1;                                                                 \
    fill = (fill + 1 == S) ? 0 : fill + 1;                                                                 \
  }
  bf16x8 af[8], bfr[4];
  if (__builtin_amdgcn_readfirstlane(wave) < 4) {       // waves 0-3: read stage t, then its MFMAs
    for (int kt = 0; kt < nk; ++kt) {
      TNR_SYNC(kt)
      TNR_READ()                   // 24 transposed LDS reads first: their latency hides under this wave's DMA issue
      TNR_REFILL(kt)               // (measured +20 % over DMA-first; the NT kernel prefers the opposite order)
      TNR_MFMA()
      TNR_ADVANCE()
    }
  } else {                                              // waves 4-7 (SIMD partners): MFMAs one stage behind
    for (int kt = 0; kt < nk; ++kt) {
      TNR_SYNC(kt)
      if (kt > 0) TNR_MFMA()
      TNR_READ()
      TNR_REFILL(kt)
      TNR_ADVANCE()
    }
    if (nk > 0) TNR_MFMA()
  }
#undef TNR_ISSUE
#undef TNR_SYNC
#undef TNR_REFILL
#undef TNR_FRAG
#undef TNR_READ
#undef TNR_MFMA
#undef TNR_ADVANCE
#pragma unroll
  for (int h = 0; h < 2; ++h)
    tn_store(*reinterpret_cast<f32x4(*)[4][4]>(&acc[4 * h]), Cs, ldc, M, N, m0 + 128 * wm, n0 + 128 * (wn >> 1), h, wn & 1, lane);
}

// C[m][n] = (accumulate ? C : 0) + sum_s slab[s][m][n]   (fixed order: deterministic)
__global__ void splitk_reduce_kernel(const float* __restrict__ slabs, long slab_stride, int S, float* C, int ldc, int M,
                                     int N, int accumulate) {
  const long total = (long)M * N;
  if (ldc == N && (total & 3) == 0 && (slab_stride & 3) == 0) {
    // dense destination (every dW slot of the gradient arena): 16-byte accesses, no index division -- the scalar form below
    // ran at 2.8 TB/s (15 us per launch, 49 launches per ViT-B step); same summation order per element, same bits
    const long total4 = total >> 2;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
      float4 s = reinterpret_cast<const float4*>(slabs)[i];
      s.x += 0.f; s.y += 0.f; s.z += 0.f; s.w += 0.f;     // the scalar form starts from 0.f + slab 0: -0.0 becomes +0.0 there
      for (int k = 1; k < S; ++k) {
        const float4 t = reinterpret_cast<const float4*>(slabs + (long)k * slab_stride)[i];
        s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
      }
      float4* c = reinterpret_cast<float4*>(C) + i;
      if (accumulate) {
        const float4 o = *c;
        s.x = o.x + s.x; s.y = o.y + s.y; s.z = o.z + s.z; s.w = o.w + s.w;
      }
      *c = s;
    }
    return;
  }
  for (long idx = (long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long)gridDim.x * blockDim.x) {
    const int m = (int)(idx / N), n = (int)(idx - (long)m * N);
    float s = 0.f;
    for (int k = 0; k < S; ++k) s += slabs[(long)k * slab_stride + idx];
    float* c = C + (long)m * ldc + n;
    *c = accumulate ? (*c + s) : s;
  }
}

// ---- column sums (bias gradient): partial[y][c] = sum over this block's rows of x[r][c] -------------
template <typename T>
__global__ __launch_bounds__(256) void colsum_partial_kernel(const T* __restrict__ x, long ld, float* __restrict__ partial,
                                                             long rows, int cols, int rows_per_block) {
  __shared__ float red[4][256];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * 256 + lane * 4;
  const long r_begin = (long)blockIdx.y * rows_per_block;
  long r_end = r_begin + rows_per_block;
  if (r_end > rows) r_end = rows;
  float s[4] = {0.f, 0.f, 0.f, 0.f};
  if (c < cols) {
    const bool vec = (c + 4 <= cols) && ((ld & 3) == 0);
    for (long r = r_begin + wave; r < r_end; r += 4) {
      const T* p = x + r * ld + c;
      if (vec) {
        if constexpr (sizeof(T) == 4) {
          const float4 v = *reinterpret_cast<const float4*>(p);
          s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
        } else {
          const bf16x4 v = *reinterpret_cast<const bf16x4*>(p);
          s[0] += (float)v[0]; s[1] += (float)v[1]; s[2] += (float)v[2]; s[3] += (float)v[3];
        }
      } else {
        for (int e = 0; e < 4 && c + e < cols; ++e) s[e] += (float)p[e];
      }
    }
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) red[wave][lane * 4 + e] = s[e];
  __syncthreads();
  const int cc = blockIdx.x * 256 + threadIdx.x;
  if (cc < cols) {
    const int i = threadIdx.x;
    partial[(long)blockIdx.y * cols + cc] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
  }
}

int colsum_parts(long rows) {
  long p = (rows + 255) / 256;
  if (p < 1) p = 1;
  if (p > 256) p = 256;
  return (int)p;
}

// kernel-variant override (0 = automatic): initialised from MV_GEMM_TILE / MV_GEMM_TN, changed by mv_gemm_force_variant
std::atomic<int> g_force_nt{getenv("MV_GEMM_TILE") ? atoi(getenv("MV_GEMM_TILE")) : 0};
std::atomic<int> g_force_tn{getenv("MV_GEMM_TN") ? atoi(getenv("MV_GEMM_TN")) : 0};

struct TnPlan {
  int tiles_m, tiles_n, splits, steps_per_split;
};

TnPlan tn_plan(int M, int N, int Kc) {
  TnPlan pl;
  pl.tiles_m = mv_cdiv(M, BM);
  pl.tiles_n = mv_cdiv(N, BN);
  const int tiles = pl.tiles_m * pl.tiles_n;
  const int nk = mv_cdiv(Kc, BK) < 1 ? 1 : mv_cdiv(Kc, BK);
  int s = mv_cdiv(512, tiles);          // aim for >= 2 workgroups per CU
  const int max_s = mv_cdiv(nk, 8);     // at least 8 K-steps (512 tokens) per split
  if (s > max_s) s = max_s;
  if (s > 64) s = 64;
  if (s < 1) s = 1;
  pl.steps_per_split = mv_cdiv(nk, s);
  pl.splits = mv_cdiv(nk, pl.steps_per_split);
  return pl;
}

// 256x256 ring variant: one workgroup per CU, so the split count is chosen to land just under ONE round of the 256 CUs
TnPlan tn_plan256(int M, int N, int Kc) {
  TnPlan pl;
  pl.tiles_m = mv_cdiv(M, BM2);
  pl.tiles_n = mv_cdiv(N, BN2);
  const int tiles = pl.tiles_m * pl.tiles_n;
  const int nk = Kc / BKR;
  int s = 256 / tiles;
  const int max_s = mv_cdiv(nk, 16);    // at least 16 stages (512 contraction rows) per split
  if (s > max_s) s = max_s;
  if (s < 1) s = 1;
  pl.steps_per_split = mv_cdiv(nk, s);
  pl.splits = mv_cdiv(nk, pl.steps_per_split);
  return pl;
}
bool tn_use_ring(int M, int N, int Kc) {
  const int force = g_force_tn.load(std::memory_order_relaxed);                        // 128 | 256: tuning and tests
  if (Kc <= 0 || Kc % BKR != 0 || force == 128) return false;
  if (force == 256) return true;
  if (!((long)M * N >= 256L * 256 * 4 && Kc >= 4096)) return false;   // >= 4 tiles and a contraction long enough to split over the chip
  // ... and enough (tile, split) items for most of the 256 CUs: a split keeps >= 16 stages, so a SHORT contraction over a
  // small output (dW of proj at batch 32: 9 tiles x 13 splits = 117 workgroups) leaves half the chip idle, where the
  // 128-tile kernel's 36 x 13 two-per-CU workgroups fill it (191 -> 250 TFLOP/s on that shape; tools/bench_gemm.py M=6304)
  const TnPlan pl = tn_plan256(M, N, Kc);
  return pl.tiles_m * pl.tiles_n * pl.splits >= 160;
}

template <typename K>
int set_smem(K kernel) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             SMEM_BYTES) == hipSuccess
             ? 0
             : -1;
}

// Tail split of the 8-phase kernel: tiles beyond the last full round of the 256 CUs become two HALF items each when
// that round would be less than half full (otherwise a whole-tile round is already the better use of the chip).
constexpr int NT_CUS = 256;
inline int nt_full_tiles(int tiles) {
  const int tail = tiles % NT_CUS;
  return (tiles >= NT_CUS && tail > 0 && 2 * tail <= NT_CUS) ? tiles - tail : tiles;
}

// Optional features of the 8-phase kernel that the automatic dispatch switches on (bit 1: column bands);
// mv_gemm_force_variant(3000 | 3002) overrides it for in-process A/B runs (tools/ab_step.py, tools/bench_gemm.py).
// Round 4 (profiles/r04_step_ab.txt): bands on = 35.54 vs 35.63 ms in the step (pairwise alternation; per shape the qkv product
// gains 0...+4 % depending on which arm runs first).  A second feature measured and REMOVED: touching the operand lines two K-tiles
// ahead of their DMA (one lane per line, +1 entry per K-tile in the vmcnt queue): bit-identical, -11...-17 % per shape, 36.95 vs
// 34.94 ms in the step -- the touches take the same request path the DMA is limited by.
constexpr int NT_FEATURES_DEFAULT = 2;
// Column-band width (in 256-column tiles) for a product with tiles_n tile columns and contraction K, or 0.  MV_NT_BAND=W forces W
// wherever it divides tiles_n (shape scans).  Measured per shape (round 3, finding 37; round 4): only the qkv projection
// (9 tile columns, K = 768) gains; N = 3072 loses with every width.
inline int nt_band_width(int tiles_n, int K, bool whole_tiles_only) {
  static const int forced = getenv("MV_NT_BAND") ? atoi(getenv("MV_NT_BAND")) : 0;
  if (forced > 0) return tiles_n % forced == 0 && tiles_n > forced ? forced : 0;
  (void)whole_tiles_only;
  return (tiles_n == 9 && K <= 1024) ? 3 : 0;
}

template <int EPI, typename CT>
int launch_nt(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, EpiArgs ep,
              hipStream_t s) {
  const int attr = MV_ONCE_PER_DEVICE(set_smem(gemm_nt_kernel<EPI, CT>) | set_smem(gemm_nt_glds_kernel<EPI, CT>) |
                          set_smem(gemm_nt_glds_kernel<EPI, CT, false>));
  if (attr != 0) return MV_ERR_LAUNCH;
  const int tiles_m = mv_cdiv(M, BM), tiles_n = mv_cdiv(N, BN);
  const int t2m = mv_cdiv(M, BM2), t2n = mv_cdiv(N, BN2);
  // Kernel choice (measured on MI355X, tools/bench_gemm.py, M = 50432): the 256x256 ring kernel wins whenever its
  // grid fills the chip for >= 4 rounds (N >= 1536) or K is long enough to amortise its fill/drain (K >= 2048);
  // otherwise two 128x128 workgroups per CU overlap each other's epilogues better.  Where the ring kernel would be
  // picked and K % 128 == 0, the 8-phase kernel replaces it (+6-14 % on every ViT-B shape: whole 128-byte lines per
  // DMA row, 16-MFMA phases).  MV_GEMM_TILE = 128 | 2564 (ring) | 2568 (8-phase) forces a variant (tuning, tests).
  int force = g_force_nt.load(std::memory_order_relaxed);
  // 3000 / 3002: automatic dispatch with the column bands of the 8-phase kernel off / on (A/B in one process, tools/ab_step.py);
  // 3100 / 3102: the same with the 8-phase kernel forced wherever it is legal, as 2568 (unit tests on small shapes)
  int feat = NT_FEATURES_DEFAULT;
  if (force == 3000 || force == 3002) {
    feat = force - 3000;
    force = 0;
  } else if (force == 3100 || force == 3102) {
    feat = force - 3100;
    force = 2568;
  }
  const bool ring_ok = K > 0 && K % BKR == 0;
  const bool ring_pick = ring_ok && ((long)t2m * t2n >= 1024 || (K >= 2048 && (long)t2m * t2n >= 256));
  const bool p8_ok = K >= 128 && K % 128 == 0 && (long)M * lda < (1L << 31) && (long)N * ldb < (1L << 31);
  // 8-phase kernel: wherever the ring would be picked, and from three quarters of one round of the chip on (half-item tail from
  // one full round; 225 tiles -- the qkv projection at batch 32 -- run 780 TFLOP/s here against 657 as 900 128-tiles)
  // Round 4 (tools/bench_gemm.py M=12608 / M=6304, profiles/r04_gemm_small_batch.txt): from half a round on (150 tiles: the
  // 768-wide outputs at batch 64) whole tiles beat the 128-tile kernel by 7 % (K = 768) to 19-26 % (K >= 2304); from a quarter of
  // a round on (75 tiles: the same outputs at batch 32) the tiles run as HALF items only -- 150 workgroups of 128 x 256 instead
  // of 75 of 256 x 256 or 300 of 128 x 128 -- for +4...15 %.
  const long t2 = (long)t2m * t2n;
  const bool p8_halves = p8_ok && !ring_pick && t2 >= NT_CUS / 4 && t2 < NT_CUS / 2;
  const bool p8_pick = p8_ok && (ring_pick || t2 >= NT_CUS / 4);
  if (((force == 2568 || force == 25680 || force == 25681) && p8_ok) || (force == 0 && p8_pick)) {
    const int a8 = MV_ONCE_PER_DEVICE(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_8phase_kernel<EPI, CT>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, P8_SMEM) == hipSuccess ? 0 : -1);
    if (a8) return MV_ERR_LAUNCH;
    // 25680: whole tiles only, 25681: half items only (A/B)
    const int tiles = t2m * t2n,
              full = force == 25680 ? tiles : (force == 25681 || (force == 0 && p8_halves)) ? 0 : nt_full_tiles(tiles);
    if (feat & 2) ep.band = nt_band_width(t2n, K, full == tiles);
    gemm_nt_8phase_kernel<EPI, CT><<<full + 2 * (tiles - full), 512, P8_SMEM, s>>>(
        (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (CT*)C, ldc, M, N, K, t2n, ep, full, 0);
    MV_CHECK_LAUNCH();
    return MV_OK;
  }
  if ((force == 2564 && ring_ok) || (force == 0 && ring_pick)) {
    const int a4 = MV_ONCE_PER_DEVICE(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_ring_kernel<EPI, CT, 4>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 4 * RSTAGE_BYTES) == hipSuccess ? 0 : -1);
    if (a4) return MV_ERR_LAUNCH;
    gemm_nt_ring_kernel<EPI, CT, 4><<<t2m * t2n, 512, 4 * RSTAGE_BYTES, s>>>((const bf16_t*)A, lda, (const bf16_t*)B, ldb,
                                                                              (CT*)C, ldc, M, N, K, t2n, ep);
  } else if (K > 0 && K % BK == 0) {
    if ((long)M * lda < (1L << 31) && (long)N * ldb < (1L << 31))       // 32-bit byte offsets of the buffer-descriptor DMA
      gemm_nt_glds_kernel<EPI, CT><<<tiles_m * tiles_n, 256, SMEM_BYTES, s>>>((const bf16_t*)A, lda, (const bf16_t*)B, ldb,
                                                                               (CT*)C, ldc, M, N, K, tiles_n, ep);
    else
      gemm_nt_glds_kernel<EPI, CT, false><<<tiles_m * tiles_n, 256, SMEM_BYTES, s>>>((const bf16_t*)A, lda, (const bf16_t*)B, ldb,
                                                                                      (CT*)C, ldc, M, N, K, tiles_n, ep);
  }
  else
    gemm_nt_kernel<EPI, CT><<<tiles_m * tiles_n, 256, SMEM_BYTES, s>>>((const bf16_t*)A, lda, (const bf16_t*)B, ldb,
                                                                        (CT*)C, ldc, M, N, K, tiles_n, ep);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

}  // namespace

extern "C" int mv_gemm_nt_bf16_scaled(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_dtype, int M,
                                      int N, int K, float alpha, const float* bias, int epilogue, const void* aux, int ld_aux,
                                      int aux_i, void* out2, int ld_out2, mv_stream_t stream);

extern "C" int mv_gemm_nt_bf16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_dtype, int M,
                               int N, int K, const float* bias, int epilogue, const void* aux, int ld_aux, int aux_i,
                               void* out2, int ld_out2, mv_stream_t stream) {
  return mv_gemm_nt_bf16_scaled(A, lda, B, ldb, C, ldc, c_dtype, M, N, K, 1.0f, bias, epilogue, aux, ld_aux, aux_i, out2,
                                ld_out2, stream);
}

extern "C" int mv_gemm_nt_bf16_scaled(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_dtype, int M,
                                      int N, int K, float alpha, const float* bias, int epilogue, const void* aux, int ld_aux,
                                      int aux_i, void* out2, int ld_out2, mv_stream_t stream) {
  MV_REQUIRE(M >= 0 && N >= 0 && K >= 0, MV_ERR_SHAPE);
  if (M == 0 || N == 0) return MV_OK;
  MV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= ((K + 7) & ~7) && ldb >= ((K + 7) & ~7), MV_ERR_ALIGN);
  MV_REQUIRE(mv_aligned16(A) && mv_aligned16(B) && mv_aligned16(C), MV_ERR_ALIGN);
  hipStream_t s = (hipStream_t)stream;
  EpiArgs ep{alpha, bias, aux, ld_aux, aux_i, out2, ld_out2, nullptr, 0.f, 0.f};
  switch (epilogue) {
    case MV_EPI_NONE:
      if (c_dtype == MV_F16)       // IEEE-half output (round 4: q / k / v of precision "bf16x3h", straight from the to_qkv product)
        return launch_nt<MV_EPI_NONE, _Float16>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
      return c_dtype == MV_F32 ? launch_nt<MV_EPI_NONE, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s)
                               : launch_nt<MV_EPI_NONE, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    case MV_EPI_GELU:
      MV_REQUIRE(c_dtype == MV_BF16, MV_ERR_UNSUPPORTED);
      return launch_nt<MV_EPI_GELU, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    case MV_EPI_RESIDUAL:
      MV_REQUIRE(c_dtype == MV_F32 && aux, MV_ERR_UNSUPPORTED);
      return launch_nt<MV_EPI_RESIDUAL, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    case MV_EPI_DGELU:
      MV_REQUIRE(c_dtype == MV_BF16 && aux, MV_ERR_UNSUPPORTED);
      return launch_nt<MV_EPI_DGELU, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    case MV_EPI_GELU_GRAD:
      MV_REQUIRE(c_dtype == MV_BF16, MV_ERR_UNSUPPORTED);
      return launch_nt<MV_EPI_GELU_GRAD, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    case MV_EPI_MUL:
      MV_REQUIRE(c_dtype == MV_BF16 && aux, MV_ERR_UNSUPPORTED);
      return launch_nt<MV_EPI_MUL, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    case MV_EPI_GELU_GRAD8:
      MV_REQUIRE(c_dtype == MV_BF16, MV_ERR_UNSUPPORTED);
      return launch_nt<MV_EPI_GELU_GRAD8, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    case MV_EPI_MUL8:
      MV_REQUIRE(c_dtype == MV_BF16 && aux, MV_ERR_UNSUPPORTED);
      return launch_nt<MV_EPI_MUL8, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    case MV_EPI_EMBED:
      MV_REQUIRE(c_dtype == MV_F32 && aux && aux_i > 0, MV_ERR_UNSUPPORTED);
      return launch_nt<MV_EPI_EMBED, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    case MV_EPI_SPLIT_DGELU:
    case MV_EPI_SPLIT_GELU: {
      // pieces out: interior tiles only (the edge path of nt_epilogue does not know these forms), fp32 aux / out2, 16-byte rows
      MV_REQUIRE(c_dtype == MV_BF16 && (aux_i == 3 || aux_i == 6) && M % 256 == 0 && N % 256 == 0 && K % 128 == 0 && K >= 128,
                 MV_ERR_UNSUPPORTED);
      MV_REQUIRE(ldc >= aux_i * N && ldc % 4 == 0, MV_ERR_ALIGN);
      if (epilogue == MV_EPI_SPLIT_DGELU) {
        MV_REQUIRE(aux && ld_aux % 4 == 0 && mv_aligned16(aux) && !bias, MV_ERR_ALIGN);
        return launch_nt<MV_EPI_SPLIT_DGELU, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
      }
      MV_REQUIRE(out2 && ld_out2 % 4 == 0 && mv_aligned16(out2), MV_ERR_ALIGN);
      return launch_nt<MV_EPI_SPLIT_GELU, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    }
    default:
      return MV_ERR_UNSUPPORTED;
  }
}

// K-split NT product for outputs with too few 256x256 tiles to fill the chip (768-wide outputs at batch 64: 150 tiles on
// 256 CUs) and a long contraction (the bf16x6 products: 6 K): slabs[s][M][N] (fp32, dense) = A[:, s-th slice] B[:, s-th
// slice]^T, bias added to slab 0; the caller sums the slabs (mv_sum_slabs_add).  K % (128 * splits) == 0.
extern "C" int mv_gemm_nt_bf16_ksplit(const void* A, int lda, const void* B, int ldb, float* slabs, int M, int N, int K,
                                      int splits, const float* bias, mv_stream_t stream) {
  MV_REQUIRE(M > 0 && N > 0 && K > 0 && splits >= 1, MV_ERR_SHAPE);
  MV_REQUIRE(K % (128 * splits) == 0 && N % 4 == 0, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K, MV_ERR_ALIGN);
  MV_REQUIRE(mv_aligned16(A) && mv_aligned16(B) && mv_aligned16(slabs), MV_ERR_ALIGN);
  // buffer-descriptor DMA: BYTE offsets and the descriptor's extent are 32-bit (as in p8_ok / launch_nt)
  MV_REQUIRE((long)M * lda < (1L << 31) && (long)N * ldb < (1L << 31), MV_ERR_UNSUPPORTED);
  const int a8 = MV_ONCE_PER_DEVICE(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_8phase_kernel<MV_EPI_NONE, float, NT_BF16, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, P8_SMEM) == hipSuccess ? 0 : -1);
  if (a8) return MV_ERR_LAUNCH;
  const int t2m = mv_cdiv(M, BM2), t2n = mv_cdiv(N, BN2);
  const int tiles = t2m * t2n, items = tiles * splits, full = nt_full_tiles(items);
  EpiArgs ep{1.0f, bias, nullptr, 0, 0, nullptr, 0, nullptr, 0.f, 0.f, tiles, K / splits, (long)M * N};
  gemm_nt_8phase_kernel<MV_EPI_NONE, float, NT_BF16, true><<<full + 2 * (items - full), 512, P8_SMEM, (hipStream_t)stream>>>(
      (const bf16_t*)A, lda, (const bf16_t*)B, ldb, slabs, N, M, N, K / splits, t2n, ep, full, 0);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// dW of an fp32 nn.Linear as a bf16x6 product (see mv_split3_bf16): A6 = split3(dY) [rows, 6 M], B6 = split3(X) [rows, 6 N],
// both in the role-0 side-by-side layout (pieces p0 p0 p1 p0 p1 p2 at column blocks 0..5, so p0 / p1 / p2 sit at blocks
// 0 / 2 / 5); C[M, N] = sum over the six pairings (0,0) (0,1) (1,0) (0,2) (1,1) (2,0) of piece_i(dY)^T piece_j(X).  The ring
// kernel walks 6 * rows contraction rows through the segment table -- the SAME splits the dX and forward products use, no
// stacked copies.
namespace {
// NSEG = 6: the bf16x6 pairings over three pieces per operand; NSEG = 3: the bf16x3 pairings (0,0) (0,1) (1,0) over TWO pieces per
// operand in the [p0 p0 p1] layout of mv_split2_bf16 (p0 at column block 0, p1 at block 2).
template <int NSEG>
int launch_tn_segments(const void* A, const void* B, float* C, int ldc, int M, int N, int rows, float* workspace,
                       size_t workspace_bytes, hipStream_t s) {
  MV_REQUIRE(M > 0 && N > 0 && rows > 0, MV_ERR_SHAPE);
  MV_REQUIRE(M % 8 == 0 && N % 8 == 0 && rows % BKR == 0, MV_ERR_UNSUPPORTED);
  MV_REQUIRE((long)rows * NSEG * M * 2 < (1L << 32) && (long)rows * NSEG * N * 2 < (1L << 32), MV_ERR_UNSUPPORTED);   // 32-bit byte offsets of the DMA
  MV_REQUIRE(mv_aligned16(A) && mv_aligned16(B) && mv_aligned16(C) && mv_aligned16(workspace), MV_ERR_ALIGN);
  const int Kc = NSEG * rows;
  MV_REQUIRE(workspace_bytes >= mv_gemm_tn_workspace_bytes(M, N, Kc), MV_ERR_WORKSPACE);
  const TnPlan pl = tn_plan256(M, N, Kc);
  const int tiles_mn = pl.tiles_m * pl.tiles_n;
  const bool direct = pl.splits == 1;
  const long slab_stride = (long)M * N;
  const int a4 = MV_ONCE_PER_DEVICE(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_ring_kernel<4, true>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, 4 * RSTAGE_BYTES) == hipSuccess ? 0 : -1);
  if (a4) return MV_ERR_LAUNCH;
  TnSeg seg;
  seg.tiles = rows / BKR;
  const int pa[6] = {0, 0, 1, 0, 1, 2}, pb[6] = {0, 1, 0, 2, 1, 0}, block[3] = {0, 2, 5};
  for (int i = 0; i < 6; ++i) {
    const int j = i < NSEG ? i : NSEG - 1;                 // entries past the last segment are never issued; keep them valid
    seg.a[i] = block[pa[j]] * M;
    seg.b[i] = block[pb[j]] * N;
  }
  gemm_tn_ring_kernel<4, true><<<tiles_mn * pl.splits, 512, 4 * RSTAGE_BYTES, s>>>(
      (const bf16_t*)A, NSEG * M, (const bf16_t*)B, NSEG * N, direct ? C : workspace, direct ? (long)ldc : (long)N,
      direct ? 0 : slab_stride, M, N, Kc, pl.tiles_n, tiles_mn, pl.steps_per_split, seg);
  MV_CHECK_LAUNCH();
  if (!direct) {
    int grid = mv_cdiv(slab_stride, 256);
    if (grid > 2048) grid = 2048;
    splitk_reduce_kernel<<<grid, 256, 0, s>>>(workspace, slab_stride, pl.splits, C, ldc, M, N, 0);
    MV_CHECK_LAUNCH();
  }
  return MV_OK;
}
}  // namespace

extern "C" int mv_gemm_tn_bf16_x6(const void* A6, const void* B6, float* C, int ldc, int M, int N, int rows, float* workspace,
                                  size_t workspace_bytes, mv_stream_t stream) {
  return launch_tn_segments<6>(A6, B6, C, ldc, M, N, rows, workspace, workspace_bytes, (hipStream_t)stream);
}

// dW of an fp32 nn.Linear as a bf16x3 product: A3 = split2(dY) [rows, 3 M], B3 = split2(X) [rows, 3 N] (mv_split2_bf16, role 0);
// C[M, N] = p0(dY)^T p0(X) + p0(dY)^T p1(X) + p1(dY)^T p0(X): 2^-16-relative products at half the work of bf16x6.
extern "C" int mv_gemm_tn_bf16_x3(const void* A3, const void* B3, float* C, int ldc, int M, int N, int rows, float* workspace,
                                  size_t workspace_bytes, mv_stream_t stream) {
  return launch_tn_segments<3>(A3, B3, C, ldc, M, N, rows, workspace, workspace_bytes, (hipStream_t)stream);
}

namespace {
template <int EPI, typename CT>
int launch_nt_i8(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, EpiArgs ep,
                 hipStream_t s) {
  const int a8 = MV_ONCE_PER_DEVICE(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_8phase_kernel<EPI, CT, NT_I8>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, P8_SMEM) == hipSuccess ? 0 : -1);
  if (a8) return MV_ERR_LAUNCH;
  const int t2m = mv_cdiv(M, BM2), t2n = mv_cdiv(N, BN2);
  const int tiles = t2m * t2n, full = nt_full_tiles(tiles);
  // bytes -> the bf16-element geometry the kernel is written in
  gemm_nt_8phase_kernel<EPI, CT, NT_I8><<<full + 2 * (tiles - full), 512, P8_SMEM, s>>>(
      (const bf16_t*)A, lda / 2, (const bf16_t*)B, ldb / 2, (CT*)C, ldc, M, N, K / 2, t2n, ep, full, 0);
  MV_CHECK_LAUNCH();
  return MV_OK;
}
}  // namespace

namespace {
template <int EPI, typename CT>
int launch_nt_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, EpiArgs ep,
                  hipStream_t s) {
  const int a8 = MV_ONCE_PER_DEVICE(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_8phase_kernel<EPI, CT, NT_F16>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, P8_SMEM) == hipSuccess ? 0 : -1);
  if (a8) return MV_ERR_LAUNCH;
  const int t2m = mv_cdiv(M, BM2), t2n = mv_cdiv(N, BN2);
  const int tiles = t2m * t2n, full = nt_full_tiles(tiles);
  gemm_nt_8phase_kernel<EPI, CT, NT_F16><<<full + 2 * (tiles - full), 512, P8_SMEM, s>>>(
      (const bf16_t*)A, lda, (const bf16_t*)B, ldb, (CT*)C, ldc, M, N, K, t2n, ep, full, 0);
  MV_CHECK_LAUNCH();
  return MV_OK;
}
}  // namespace

extern "C" int mv_gemm_nt_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                              const float* bias, int epilogue, const void* aux, int ld_aux, void* out2, int ld_out2,
                              mv_stream_t stream) {
  MV_REQUIRE(M >= 0 && N >= 0 && K > 0, MV_ERR_SHAPE);
  if (M == 0 || N == 0) return MV_OK;
  MV_REQUIRE(K % 128 == 0, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= K && ldb >= K, MV_ERR_ALIGN);
  MV_REQUIRE(mv_aligned16(A) && mv_aligned16(B) && mv_aligned16(C), MV_ERR_ALIGN);
  MV_REQUIRE((long)M * lda < (1L << 31) && (long)N * ldb < (1L << 31), MV_ERR_SHAPE);
  hipStream_t s = (hipStream_t)stream;
  EpiArgs ep{1.0f, bias, aux, ld_aux, 0, out2, ld_out2, nullptr, 0.f, 0.f};
  switch (epilogue) {
    case MV_EPI_NONE: return launch_nt_f16<MV_EPI_NONE, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    case MV_EPI_RESIDUAL:
      MV_REQUIRE(aux, MV_ERR_UNSUPPORTED);
      return launch_nt_f16<MV_EPI_RESIDUAL, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    default: return MV_ERR_UNSUPPORTED;
  }
}

namespace {
template <int EPI, typename CT>
int launch_nt_f8c(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, EpiArgs ep, hipStream_t s) {
  const int a8 = MV_ONCE_PER_DEVICE(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_nt_8phase_kernel<EPI, CT, NT_F8C>),
                                            hipFuncAttributeMaxDynamicSharedMemorySize, P8_SMEM) == hipSuccess ? 0 : -1);
  if (a8) return MV_ERR_LAUNCH;
  const int t2m = mv_cdiv(M, BM2), t2n = mv_cdiv(N, BN2);
  const int tiles = t2m * t2n, full = nt_full_tiles(tiles);
  ep.f8_tiles16 = K / 64;
  // a row of 4 K bytes in the 2-byte geometry the kernel is written in: 2 K "elements", K / 32 K-tiles
  gemm_nt_8phase_kernel<EPI, CT, NT_F8C><<<full + 2 * (tiles - full), 512, P8_SMEM, s>>>(
      (const bf16_t*)A, lda / 2, (const bf16_t*)B, ldb / 2, (CT*)C, ldc, M, N, 2 * K, t2n, ep, full, 0);
  MV_CHECK_LAUNCH();
  return MV_OK;
}
}  // namespace

extern "C" int mv_gemm_nt_f8c(const void* A, long lda_bytes, const void* B, long ldb_bytes, void* C, int ldc, int c_dtype, int M, int N,
                              int K, int scale_exp, const float* bias, int epilogue, const void* aux, int ld_aux, mv_stream_t stream) {
  MV_REQUIRE(M >= 0 && N >= 0 && K > 0, MV_ERR_SHAPE);
  if (M == 0 || N == 0) return MV_OK;
  MV_REQUIRE(K % 128 == 0 && scale_exp > -127 && scale_exp <= 127, MV_ERR_UNSUPPORTED);        // whole K-tile pairs per segment
  MV_REQUIRE(lda_bytes % 16 == 0 && ldb_bytes % 16 == 0 && lda_bytes >= 4L * K && ldb_bytes >= 4L * K, MV_ERR_ALIGN);
  MV_REQUIRE(mv_aligned16(A) && mv_aligned16(B) && mv_aligned16(C), MV_ERR_ALIGN);
  MV_REQUIRE((long)M * lda_bytes < (1L << 32) && (long)N * ldb_bytes < (1L << 32), MV_ERR_SHAPE);  // 32-bit byte offsets of the DMA
  hipStream_t s = (hipStream_t)stream;
  EpiArgs ep{1.0f, bias, aux, ld_aux, 0, nullptr, 0, nullptr, 0.f, 0.f};
  ep.f8_scale = scale_exp + 127;
  switch (epilogue) {
    case MV_EPI_NONE:
      MV_REQUIRE(c_dtype == MV_F32 || c_dtype == MV_BF16, MV_ERR_UNSUPPORTED);
      return c_dtype == MV_F32 ? launch_nt_f8c<MV_EPI_NONE, float>(A, (int)lda_bytes, B, (int)ldb_bytes, C, ldc, M, N, K, ep, s)
                               : launch_nt_f8c<MV_EPI_NONE, bf16_t>(A, (int)lda_bytes, B, (int)ldb_bytes, C, ldc, M, N, K, ep, s);
    case MV_EPI_RESIDUAL:
      MV_REQUIRE(c_dtype == MV_F32 && aux, MV_ERR_UNSUPPORTED);
      return launch_nt_f8c<MV_EPI_RESIDUAL, float>(A, (int)lda_bytes, B, (int)ldb_bytes, C, ldc, M, N, K, ep, s);
    default:
      return MV_ERR_UNSUPPORTED;
  }
}

extern "C" int mv_gemm_nt_i8(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int c_dtype, int M, int N, int K,
                             float alpha, const float* bias, const int* icorr, int epilogue, const void* aux, int ld_aux,
                             float q_scale, int q_zero_point, mv_stream_t stream) {
  MV_REQUIRE(M >= 0 && N >= 0 && K > 0, MV_ERR_SHAPE);
  if (M == 0 || N == 0) return MV_OK;
  MV_REQUIRE(K % 256 == 0, MV_ERR_UNSUPPORTED);                       // two whole 128-byte K-tiles per 8-phase iteration
  MV_REQUIRE(lda % 16 == 0 && ldb % 16 == 0 && lda >= K && ldb >= K, MV_ERR_ALIGN);
  MV_REQUIRE(mv_aligned16(A) && mv_aligned16(B) && mv_aligned16(C), MV_ERR_ALIGN);
  MV_REQUIRE((long)M * lda < (1L << 32) && (long)N * ldb < (1L << 32), MV_ERR_SHAPE);
  hipStream_t s = (hipStream_t)stream;
  EpiArgs ep{alpha, bias, aux, ld_aux, 0, nullptr, 0, icorr, q_scale > 0.f ? 1.0f / q_scale : 0.f, (float)q_zero_point};
  switch (epilogue) {
    case MV_EPI_GELU_Q8:
      MV_REQUIRE(c_dtype == MV_I8 && q_scale > 0.f && q_zero_point >= 0 && q_zero_point <= 255, MV_ERR_UNSUPPORTED);
      MV_REQUIRE(M % 256 == 0 && N % 256 == 0 && ldc % 4 == 0 && (reinterpret_cast<uintptr_t>(bias) & 15) == 0, MV_ERR_UNSUPPORTED);
      return launch_nt_i8<MV_EPI_GELU_Q8, int8_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    case MV_EPI_NONE:
      MV_REQUIRE(c_dtype == MV_F32 || c_dtype == MV_BF16, MV_ERR_UNSUPPORTED);
      return c_dtype == MV_F32 ? launch_nt_i8<MV_EPI_NONE, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s)
                               : launch_nt_i8<MV_EPI_NONE, bf16_t>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    case MV_EPI_RESIDUAL:
      MV_REQUIRE(c_dtype == MV_F32 && aux, MV_ERR_UNSUPPORTED);
      return launch_nt_i8<MV_EPI_RESIDUAL, float>(A, lda, B, ldb, C, ldc, M, N, K, ep, s);
    default:
      return MV_ERR_UNSUPPORTED;
  }
}

extern "C" int mv_gemm_force_variant(int nt_variant, int tn_variant) {
  const bool nt_ok = nt_variant == 0 || nt_variant == 128 || nt_variant == 2564 || nt_variant == 2568 || nt_variant == 25680 || nt_variant == 25681 ||
                     nt_variant == 3000 || nt_variant == 3002 || nt_variant == 3100 || nt_variant == 3102;
  const bool tn_ok = tn_variant == 0 || tn_variant == 128 || tn_variant == 256;
  MV_REQUIRE(nt_ok && tn_ok, MV_ERR_UNSUPPORTED);
  g_force_nt.store(nt_variant, std::memory_order_relaxed);
  g_force_tn.store(tn_variant, std::memory_order_relaxed);
  return MV_OK;
}

extern "C" size_t mv_gemm_tn_workspace_bytes(int M, int N, int Kc) {
  // the larger of the two plans: independent of which variant is (or is forced to be) launched later
  int splits = tn_plan(M, N, Kc).splits;
  const int kr = Kc & ~(BKR - 1);               // the ring kernel also takes the whole stages of a ragged contraction
  if (kr > 0 && tn_plan256(M, N, kr).splits > splits) splits = tn_plan256(M, N, kr).splits;
  if (kr > 0 && kr != Kc && tn_plan(M, N, kr).splits > splits) splits = tn_plan(M, N, kr).splits;
  const size_t slabs = (size_t)splits * (size_t)M * (size_t)N * sizeof(float);
  const size_t cs = (size_t)colsum_parts(Kc) * (size_t)M * sizeof(float);
  return slabs + cs + 256;
}

extern "C" int mv_colsum(const void* x, int x_dtype, long ld, float* out, int accumulate, long rows, int cols,
                         float* workspace, size_t workspace_bytes, mv_stream_t stream) {
  MV_REQUIRE(rows >= 0 && cols > 0, MV_ERR_SHAPE);
  MV_REQUIRE(x_dtype == MV_F32 || x_dtype == MV_BF16, MV_ERR_UNSUPPORTED);
  const int parts = colsum_parts(rows);
  MV_REQUIRE(workspace_bytes >= (size_t)parts * cols * sizeof(float), MV_ERR_WORKSPACE);
  hipStream_t s = (hipStream_t)stream;
  if (x_dtype == MV_F32 && rows <= 2048) {      // already a partial-sum table (e.g. the DGELU epilogue's): one stage is enough
    mv_reduce_rows_kernel<<<mv_reduce_rows_grid(cols), 1024, 0, s>>>((const float*)x, (int)rows, cols, ld, out, out, out, cols, cols,
                                                                     accumulate);
    MV_CHECK_LAUNCH();
    return MV_OK;
  }
  const int rpb = (int)((rows + parts - 1) / parts);
  dim3 grid(mv_cdiv(cols, 256), parts);
  if (x_dtype == MV_F32)
    colsum_partial_kernel<float><<<grid, 256, 0, s>>>((const float*)x, ld, workspace, rows, cols, rpb < 1 ? 1 : rpb);
  else
    colsum_partial_kernel<bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)x, ld, workspace, rows, cols, rpb < 1 ? 1 : rpb);
  MV_CHECK_LAUNCH();
  mv_reduce_rows_kernel<<<mv_reduce_rows_grid(cols), 1024, 0, s>>>(workspace, parts, cols, (long)cols, out, out, out, cols, cols, accumulate);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_gemm_tn_bf16(const void* A, int lda, const void* B, int ldb, float* C, int ldc, int M, int N, int Kc,
                               int accumulate, float* colsum, float* workspace, size_t workspace_bytes,
                               mv_stream_t stream) {
  MV_REQUIRE(M > 0 && N > 0 && Kc >= 0, MV_ERR_SHAPE);
  MV_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && lda >= ((M + 7) & ~7) && ldb >= ((N + 7) & ~7), MV_ERR_ALIGN);
  MV_REQUIRE(mv_aligned16(A) && mv_aligned16(B) && mv_aligned16(C) && mv_aligned16(workspace), MV_ERR_ALIGN);
  MV_REQUIRE(workspace_bytes >= mv_gemm_tn_workspace_bytes(M, N, Kc), MV_ERR_WORKSPACE);
  const int attr = MV_ONCE_PER_DEVICE(set_smem(gemm_tn_kernel) | set_smem(gemm_tn_glds_kernel));
  if (attr != 0) return MV_ERR_LAUNCH;
  hipStream_t s = (hipStream_t)stream;
  // A contraction that is not a whole number of the ring kernel's 32-row stages (197 tokens x a batch that is not a multiple
  // of 32) used to fall to the 128-tile register-staged kernel for ALL of it: the step took as long at batch 48 as at 64.
  // Now the ring kernel takes the whole stages and the general kernel adds the last < 32 rows (accumulate).
  if (Kc % BKR != 0 && tn_use_ring(M, N, Kc & ~(BKR - 1))) {
    const int main_rows = Kc & ~(BKR - 1);
    int rc = mv_gemm_tn_bf16(A, lda, B, ldb, C, ldc, M, N, main_rows, accumulate, nullptr, workspace, workspace_bytes, stream);
    if (rc != MV_OK) return rc;
    rc = mv_gemm_tn_bf16(reinterpret_cast<const bf16_t*>(A) + (long)main_rows * lda, lda,
                         reinterpret_cast<const bf16_t*>(B) + (long)main_rows * ldb, ldb, C, ldc, M, N, Kc - main_rows, 1, nullptr,
                         workspace, workspace_bytes, stream);
    if (rc != MV_OK) return rc;
    if (colsum) {
      const TnPlan plc = tn_plan256(M, N, main_rows);
      float* cs_ws = workspace + (size_t)plc.splits * (size_t)M * (size_t)N;
      return mv_colsum(A, MV_BF16, lda, colsum, accumulate, Kc, M, cs_ws, (size_t)colsum_parts(Kc) * (size_t)M * sizeof(float),
                       stream);
    }
    return MV_OK;
  }
  const bool ring = tn_use_ring(M, N, Kc) && (long)Kc * lda * 2 < (1L << 32) && (long)Kc * ldb * 2 < (1L << 32);   // 32-bit byte offsets
  // (An 8-phase port of this kernel -- the gemm_nt_8phase_kernel schedule with [64 kc][128 col] slots -- measured 10-15 %
  // SLOWER than the ring in the same process, 851 vs 968 and 909 vs 1031 TFLOP/s: the ring's DMA rows are whole
  // 256-byte lines already, so the port only added barriers and halved the bytes in flight.  Not kept.)
  const TnPlan pl = ring ? tn_plan256(M, N, Kc) : tn_plan(M, N, Kc);
  const int tiles_mn = pl.tiles_m * pl.tiles_n;
  const bool direct = pl.splits == 1 && !accumulate;
  const long slab_stride = (long)M * N;
  if (ring) {
    const int a4 = MV_ONCE_PER_DEVICE(hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_tn_ring_kernel<4>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, 4 * RSTAGE_BYTES) == hipSuccess ? 0 : -1);
    if (a4) return MV_ERR_LAUNCH;
    gemm_tn_ring_kernel<4><<<tiles_mn * pl.splits, 512, 4 * RSTAGE_BYTES, s>>>(
        (const bf16_t*)A, lda, (const bf16_t*)B, ldb, direct ? C : workspace, direct ? (long)ldc : (long)N,
        direct ? 0 : slab_stride, M, N, Kc, pl.tiles_n, tiles_mn, pl.steps_per_split, TnSeg{});
  } else if (Kc > 0 && Kc % BK == 0 && (long)Kc * lda * 2 < (1L << 32) && (long)Kc * ldb * 2 < (1L << 32))
    gemm_tn_glds_kernel<<<tiles_mn * pl.splits, 256, SMEM_BYTES, s>>>(
        (const bf16_t*)A, lda, (const bf16_t*)B, ldb, direct ? C : workspace, direct ? (long)ldc : (long)N,
        direct ? 0 : slab_stride, M, N, Kc, pl.tiles_n, tiles_mn, pl.steps_per_split);
  else
    gemm_tn_kernel<<<tiles_mn * pl.splits, 256, SMEM_BYTES, s>>>(
        (const bf16_t*)A, lda, (const bf16_t*)B, ldb, direct ? C : workspace, direct ? (long)ldc : (long)N,
        direct ? 0 : slab_stride, M, N, Kc, pl.tiles_n, tiles_mn, pl.steps_per_split);
  MV_CHECK_LAUNCH();
  if (!direct) {
    int grid = mv_cdiv(slab_stride, 256);
    if (grid > 2048) grid = 2048;
    splitk_reduce_kernel<<<grid, 256, 0, s>>>(workspace, slab_stride, pl.splits, C, ldc, M, N, accumulate);
    MV_CHECK_LAUNCH();
  }
  if (colsum) {   // separate pass over dY (next step: produce these sums where dY is written; fusing them here as
                  // ones-operand MFMAs pushed this 235-VGPR kernel into spills)
    float* cs_ws = workspace + (size_t)pl.splits * (size_t)slab_stride;
    const int rc = mv_colsum(A, MV_BF16, lda, colsum, accumulate, Kc, M, cs_ws,
                             (size_t)colsum_parts(Kc) * (size_t)M * sizeof(float), stream);
    if (rc != MV_OK) return rc;
  }
  return MV_OK;
}
