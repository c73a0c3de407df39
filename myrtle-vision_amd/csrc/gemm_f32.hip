// Generic strided, batched fp32 contraction + row softmax: the exact-parity (precision="fp32") path.
// Two kernels with bit-identical results: gemm_f32_mfma_kernel (matrix cores, the default) and gemm_f32_kernel (FMA).
//
// Every product of the hot path can be expressed through element strides, so this one kernel covers
// nn.Linear forward / dX / dW and the materialised attention products q k^T, p v and their gradients
// (vit.py:92-96) when bit-level agreement with the fp32 reference matters more than speed, or when a
// forward hook on Attention.attn_output (vit.py:80-82,94) needs the probabilities in memory.
// Arithmetic: one fp32 FMA chain per output in k order (same as an ATen/BLAS fp32 dot up to ordering).
// 64x64 tile, K-step 16, 256 threads x (4x4) outputs, LDS-staged, staging map chosen per operand so the
// unit-stride axis is the one consecutive lanes walk.
#include "mv_common.h"

#include <string.h>

#include <atomic>

namespace {

constexpr int TM = 64, TN = 64, TK = 16;

struct F32Args {
  const float* A; long sa_m, sa_k, sa_b1, sa_b2;
  const float* B; long sb_k, sb_n, sb_b1, sb_b2;
  float* C; long sc_m, sc_n, sc_b1, sc_b2;
  int M, N, K, nb2;
  float alpha; int accumulate;
  const float* bias; const float* aux; long ld_aux; int aux_i; float* out2; long ld_out2;
};

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_kernel(F32Args a) {
  __shared__ float As[TK][TM + 4];
  __shared__ float Bs[TK][TN + 4];
  const int tid = threadIdx.x;
  const int b1 = blockIdx.z / a.nb2, b2 = blockIdx.z % a.nb2;
  const float* A = a.A + b1 * a.sa_b1 + b2 * a.sa_b2;
  const float* B = a.B + b1 * a.sb_b1 + b2 * a.sb_b2;
  float* C = a.C + b1 * a.sc_b1 + b2 * a.sc_b2;
  const int m0 = blockIdx.y * TM, n0 = blockIdx.x * TN;
  const int ty = tid >> 4, tx = tid & 15;
  const bool a_kfast = (a.sa_k == 1), b_nfast = (a.sb_n == 1);

  float acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = 0.f;

  for (int k0 = 0; k0 < a.K; k0 += TK) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      int m, k;
      if (a_kfast) { k = tid & 15; m = (tid >> 4) + 16 * i; } else { m = tid & 63; k = (tid >> 6) + 4 * i; }
      const int gm = m0 + m, gk = k0 + k;
      As[k][m] = (gm < a.M && gk < a.K) ? A[gm * a.sa_m + gk * a.sa_k] : 0.f;
      int n, kb;
      if (b_nfast) { n = tid & 63; kb = (tid >> 6) + 4 * i; } else { kb = tid & 15; n = (tid >> 4) + 16 * i; }
      const int gn = n0 + n, gkb = k0 + kb;
      Bs[kb][n] = (gn < a.N && gkb < a.K) ? B[gkb * a.sb_k + gn * a.sb_n] : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TK; ++k) {
      const float4 av = *reinterpret_cast<const float4*>(&As[k][ty * 4]);
      const float4 bv = *reinterpret_cast<const float4*>(&Bs[k][tx * 4]);
      const float ar[4] = {av.x, av.y, av.z, av.w}, br[4] = {bv.x, bv.y, bv.z, bv.w};
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = fmaf(ar[i], br[j], acc[i][j]);
    }
    __syncthreads();
  }

#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int m = m0 + ty * 4 + i;
    if (m >= a.M) continue;
    long crow = m;
    int patch = 0;
    if constexpr (EPI == MV_EPI_EMBED) {
      const int img = m / a.aux_i;
      patch = m - img * a.aux_i;
      crow = (long)img * (a.aux_i + 1) + 1 + patch;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int n = n0 + tx * 4 + j;
      if (n >= a.N) continue;
      float v = acc[i][j] * a.alpha;
      if (a.bias) v += a.bias[n];
      if constexpr (EPI == MV_EPI_GELU) {
        if (a.out2) a.out2[(long)m * a.ld_out2 + n] = v;
        v = gelu_f(v);
      } else if constexpr (EPI == MV_EPI_RESIDUAL) {
        v += a.aux[(long)m * a.ld_aux + n];
      } else if constexpr (EPI == MV_EPI_DGELU) {
        v *= dgelu_f(a.aux[(long)m * a.ld_aux + n]);
      } else if constexpr (EPI == MV_EPI_EMBED) {
        v += a.aux[(long)(1 + patch) * a.ld_aux + n];
      }
      float* c = C + crow * a.sc_m + n * a.sc_n;
      *c = a.accumulate ? (*c + v) : v;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same contraction on the matrix cores: v_mfma_f32_16x16x4_f32 (f32 in, f32 accumulate).  The instruction IS a
// k-ordered fmaf chain (one rounding per product, no wider accumulator: MI355X_MICROARCH.md, Matrix cores), so every output
// is bit-for-bit what gemm_f32_kernel computes -- at the f32 MFMA rate (64 FLOP/clk/SIMD = the f32 VALU peak, but with one
// operand VGPR per 16x16x4 block instead of a 4x4 register tile per thread, and no VALU issue competing with address
// arithmetic): 4-6x the FMA kernel on the ViT shapes.
//   block = 4 waves as 2 x 2, wave tile (16 WT)^2 with WT = 4 (128 x 128 block) or 2 (64 x 64 block: the 197-token
//   attention products); K-step 16 = four MFMA k-blocks.
//   LDS image, both operands: X[idx][16 k] stored as [idx][g][kk] with k = 4 kk + g, rows of 20 floats (80 B): lane
//   (idx = l & 15, g = l >> 4) -- exactly the MFMA operand map A[l&15][k = l>>4] -- reads its four k-blocks as ONE
//   ds_read_b128; element kk of that vector feeds MFMA kk.
//   The MFMA is issued as D' = Bfrag x Afrag (i.e. it produces C^T), so a lane holds four CONSECUTIVE COLUMNS of one output
//   row: float4 epilogue accesses when C is row-major.
//   Staging is register double-buffered (global loads of step t+1 in flight under the MFMAs of step t, one barrier per
//   step) with the same stride-adaptive thread map as the FMA kernel: consecutive lanes walk the operand's unit-stride axis.
// shared epilogue of the two MFMA kernels
template <int EPI, int WT>
__device__ __forceinline__ void f32_mfma_epilogue(const F32Args& a, float* C, f32x4 (&acc)[WT][WT], int m0, int n0, int wr, int wc,
                                                  int lane) {
  constexpr int TB = 32 * WT;
  // epilogue: lane holds, per 16x16 block (i, j), row m = .. + (lane & 15) and columns n = .. + 4 (lane >> 4) + 0..3
  const bool vec = a.sc_n == 1 && (a.sc_m & 3) == 0 && (reinterpret_cast<uintptr_t>(C) & 15) == 0 && !a.accumulate;
  // interior blocks of row-major outputs: branch-free, every bias / aux vector loaded before the first store (a
  // per-element "n < N ? load : 0" makes the compiler branch around each load and wait vmcnt(0) per element)
  const bool interior = vec && m0 + TB <= a.M && n0 + TB <= a.N && (a.ld_aux & 3) == 0 && (a.ld_out2 & 3) == 0 &&
                        (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0 && (reinterpret_cast<uintptr_t>(a.aux) & 15) == 0 &&
                        (reinterpret_cast<uintptr_t>(a.out2) & 15) == 0;
  if (interior) {
    const int mb = m0 + wr * WT * 16 + (lane & 15), nb0 = n0 + wc * WT * 16 + (lane >> 4) * 4;
    float4 bv[WT];
#pragma unroll
    for (int j = 0; j < WT; ++j)
      bv[j] = a.bias ? *reinterpret_cast<const float4*>(a.bias + nb0 + 16 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int i = 0; i < WT; ++i) {
      const int m = mb + 16 * i;
      long crow = m, arow = m;
      if constexpr (EPI == MV_EPI_EMBED) {
        const int img = m / a.aux_i;
        arow = 1 + (m - img * a.aux_i);
        crow = (long)img * (a.aux_i + 1) + arow;
      }
      float4 ax[WT];
      if constexpr (EPI == MV_EPI_RESIDUAL || EPI == MV_EPI_DGELU || EPI == MV_EPI_EMBED) {
#pragma unroll
        for (int j = 0; j < WT; ++j) ax[j] = *reinterpret_cast<const float4*>(a.aux + arow * a.ld_aux + nb0 + 16 * j);
      }
#pragma unroll
      for (int j = 0; j < WT; ++j) {
        const float bb[4] = {bv[j].x, bv[j].y, bv[j].z, bv[j].w};
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float t = acc[i][j][r] * a.alpha;
          if (a.bias) t += bb[r];
          v[r] = t;
        }
        if constexpr (EPI == MV_EPI_GELU) {
          if (a.out2) *reinterpret_cast<float4*>(a.out2 + (long)m * a.ld_out2 + nb0 + 16 * j) = make_float4(v[0], v[1], v[2], v[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = gelu_f(v[r]);
        } else if constexpr (EPI == MV_EPI_RESIDUAL || EPI == MV_EPI_EMBED) {
          v[0] += ax[j].x; v[1] += ax[j].y; v[2] += ax[j].z; v[3] += ax[j].w;
        } else if constexpr (EPI == MV_EPI_DGELU) {
          v[0] *= dgelu_f(ax[j].x); v[1] *= dgelu_f(ax[j].y); v[2] *= dgelu_f(ax[j].z); v[3] *= dgelu_f(ax[j].w);
        }
        *reinterpret_cast<float4*>(C + crow * a.sc_m + nb0 + 16 * j) = make_float4(v[0], v[1], v[2], v[3]);
      }
    }
    return;
  }
#pragma unroll
  for (int i = 0; i < WT; ++i) {
    const int m = m0 + (wr * WT + i) * 16 + (lane & 15);
    if (m >= a.M) continue;
    long crow = m;
    int patch = 0;
    if constexpr (EPI == MV_EPI_EMBED) {
      const int img = m / a.aux_i;
      patch = m - img * a.aux_i;
      crow = (long)img * (a.aux_i + 1) + 1 + patch;
    }
#pragma unroll
    for (int j = 0; j < WT; ++j) {
      const int nb = n0 + (wc * WT + j) * 16 + (lane >> 4) * 4;
      float v[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int n = nb + r;
        float t = acc[i][j][r] * a.alpha;
        if (n < a.N) {
          if (a.bias) t += a.bias[n];
          if constexpr (EPI == MV_EPI_GELU) {
            if (a.out2) a.out2[(long)m * a.ld_out2 + n] = t;
            t = gelu_f(t);
          } else if constexpr (EPI == MV_EPI_RESIDUAL) {
            t += a.aux[(long)m * a.ld_aux + n];
          } else if constexpr (EPI == MV_EPI_DGELU) {
            t *= dgelu_f(a.aux[(long)m * a.ld_aux + n]);
          } else if constexpr (EPI == MV_EPI_EMBED) {
            t += a.aux[(long)(1 + patch) * a.ld_aux + n];
          }
        }
        v[r] = t;
      }
      if (vec && nb + 3 < a.N) {
        *reinterpret_cast<float4*>(C + crow * a.sc_m + nb) = make_float4(v[0], v[1], v[2], v[3]);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if (nb + r >= a.N) continue;
          float* c = C + crow * a.sc_m + (long)(nb + r) * a.sc_n;
          *c = a.accumulate ? (*c + v[r]) : v[r];
        }
      }
    }
  }
}

template <int EPI, int WT>
__global__ __launch_bounds__(256) void gemm_f32_mfma_kernel(F32Args a) {
  constexpr int TB = 32 * WT;              // block tile edge
  constexpr int LDK = 20;                  // floats per LDS row (16 + 4 pad; keeps 16-byte alignment of the b128 reads)
  constexpr int PER = TB * 16 / 256;       // staged elements per thread and operand
  __shared__ __attribute__((aligned(16))) float As[2][TB * LDK];
  __shared__ __attribute__((aligned(16))) float Bs[2][TB * LDK];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int b1 = blockIdx.z / a.nb2, b2 = blockIdx.z % a.nb2;
  const float* A = a.A + b1 * a.sa_b1 + b2 * a.sa_b2;
  const float* B = a.B + b1 * a.sb_b1 + b2 * a.sb_b2;
  float* C = a.C + b1 * a.sc_b1 + b2 * a.sc_b2;
  const int m0 = blockIdx.y * TB, n0 = blockIdx.x * TB;
  const bool a_kfast = (a.sa_k == 1), b_kfast = (a.sb_k == 1);

  // staging map: element e (0 .. PER-1) of this thread is tile position (idx0 + e di, k0 + e dk) -- consecutive lanes walk
  // the operand's unit-stride axis; the per-element step is wave-uniform, so one base pointer per operand is enough
  const int ai0 = a_kfast ? (tid >> 4) : (tid % TB), ak0 = a_kfast ? (tid & 15) : (tid / TB);
  const int bi0 = b_kfast ? (tid >> 4) : (tid % TB), bk0 = b_kfast ? (tid & 15) : (tid / TB);
  const int adi = a_kfast ? 16 : 0, adk = a_kfast ? 0 : 256 / TB, bdi = b_kfast ? 16 : 0, bdk = b_kfast ? 0 : 256 / TB;
  const float* const pa = A + (long)(m0 + ai0) * a.sa_m + (long)ak0 * a.sa_k;
  const float* const pb = B + (long)(n0 + bi0) * a.sb_n + (long)bk0 * a.sb_k;
  const long astep = (long)adi * a.sa_m + (long)adk * a.sa_k, bstep = (long)bdi * a.sb_n + (long)bdk * a.sb_k;
  // Loads are unconditional from clamped addresses; the zero-fill select happens at STORE time, after the MFMAs of the
  // current step -- selecting right after the load would make the compiler wait for the data before the MFMA block and
  // expose a full memory latency per step (seen in the first version's ISA: vmcnt(0) in front of the MFMAs).
  float ra[PER], rb[PER];
  unsigned oka = 0, okb = 0;                 // bit e: element e is inside the operand
  auto load_tile = [&](int k0) {
    oka = okb = 0;
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const bool ok = (m0 + ai0 + e * adi) < a.M && (k0 + ak0 + e * adk) < a.K;
      ra[e] = *(ok ? pa + e * astep + (long)k0 * a.sa_k : A);
      oka |= (ok ? 1u : 0u) << e;
      const bool ob = (n0 + bi0 + e * bdi) < a.N && (k0 + bk0 + e * bdk) < a.K;
      rb[e] = *(ob ? pb + e * bstep + (long)k0 * a.sb_k : B);
      okb |= (ob ? 1u : 0u) << e;
    }
  };
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int e = 0; e < PER; ++e) {
      const int ka = ak0 + e * adk, kb = bk0 + e * bdk;
      As[buf][(ai0 + e * adi) * LDK + (ka & 3) * 4 + (ka >> 2)] = ((oka >> e) & 1u) ? ra[e] : 0.f;
      Bs[buf][(bi0 + e * bdi) * LDK + (kb & 3) * 4 + (kb >> 2)] = ((okb >> e) & 1u) ? rb[e] : 0.f;
    }
  };

  f32x4 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = (a.K + 15) >> 4;
  const int frag = (lane & 15) * LDK + (lane >> 4) * 4;
  if (nk > 0) {
    load_tile(0);
    store_tile(0);
  }
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile((kt + 1) << 4);
    f32x4 af[WT], bf[WT];
#pragma unroll
    for (int i = 0; i < WT; ++i) {
      af[i] = *reinterpret_cast<const f32x4*>(&As[cur][(wr * WT + i) * 16 * LDK + frag]);
      bf[i] = *reinterpret_cast<const f32x4*>(&Bs[cur][(wc * WT + i) * 16 * LDK + frag]);
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][kk], af[i][kk], acc[i][j], 0, 0, 0);
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }

  f32_mfma_epilogue<EPI, WT>(a, C, acc, m0, n0, wr, wc, lane);
}

// ------------------------------------------------------------------------------------------------------------------
// Fast form of the MFMA kernel for the three nn.Linear products at their usual alignment (what the generic kernel's ISA
// showed: ~700 address / select / branch instructions per wave and K-step around 64 MFMAs, 32 scalar loads per thread):
//   * the unit-stride axis of each operand is a template parameter (AK / BK: true = K is contiguous, false = the M / N index
//     is), so the staging map and all strides are compile-time shapes;
//   * global loads are 16 bytes: 2 + 2 per thread and K-step instead of 16 + 16; rows past the end are CLAMPED to the last
//     valid row (their products only reach outputs that are never stored), so there are no bounds selects in the loop;
//   * requirements checked by the dispatcher: K % 16 == 0, 16-byte aligned operands and leading dimensions % 4 == 0, and
//     M % 4 == 0 / N % 4 == 0 for an operand read along its row index.
// Same LDS image, fragments, MFMA order (k ascending) and epilogue as gemm_f32_mfma_kernel<EPI, 4>: identical bits.
template <int EPI, bool AK, bool BKF>
__global__ __launch_bounds__(256, 2) void gemm_f32_mfma_fast_kernel(F32Args a) {
  // LDS image per operand.  K-contiguous: [idx][g][kk] with k = 4 kk + g, rows of LDK = 20 floats (ds_read_b128 fragments,
  // as gemm_f32_mfma_kernel).  Index-contiguous: k-major [16 k][LDI = 144] -- the thread's four consecutive rows of one k go
  // out as ONE ds_write_b128 (written row by row into the [idx][..] image they were a 16-way bank conflict: all lanes of a
  // wave 80 floats apart), fragments are four conflict-free ds_read_b32 (lane groups g = 0..3 read k rows 144 floats = 16
  // banks apart).
  constexpr int WT = 4, TB = 128, LDK = 20, LDI = 144;
  constexpr int A_SZ = AK ? TB * LDK : 16 * LDI, B_SZ = BKF ? TB * LDK : 16 * LDI;
  __shared__ __attribute__((aligned(16))) float As[2][A_SZ];
  __shared__ __attribute__((aligned(16))) float Bs[2][B_SZ];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wr = wave >> 1, wc = wave & 1;
  const int b1 = blockIdx.z / a.nb2, b2 = blockIdx.z % a.nb2;
  const float* A = a.A + b1 * a.sa_b1 + b2 * a.sa_b2;
  const float* B = a.B + b1 * a.sb_b1 + b2 * a.sb_b2;
  float* C = a.C + b1 * a.sc_b1 + b2 * a.sc_b2;
  const int m0 = blockIdx.y * TB, n0 = blockIdx.x * TB;

  // K-contiguous operand: thread -> (row (tid >> 2) + 64 e, 16-byte chunk tid & 3 of the 16 k); index-contiguous operand:
  // thread -> (k (tid >> 5) + 8 e, rows 4 (tid & 31) .. + 3).  One base pointer per element e, advanced by 16 k per step.
  const float* pa[2];
  const float* pb[2];
  long a_adv, b_adv;
  if constexpr (AK) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      int r = m0 + (tid >> 2) + 64 * e;
      r = r < a.M ? r : a.M - 1;
      pa[e] = A + (long)r * a.sa_m + 4 * (tid & 3);
    }
    a_adv = 16;
  } else {
    int r = m0 + 4 * (tid & 31);
    r = r + 3 < a.M ? r : a.M - 4;
#pragma unroll
    for (int e = 0; e < 2; ++e) pa[e] = A + (long)((tid >> 5) + 8 * e) * a.sa_k + r;
    a_adv = 16 * a.sa_k;
  }
  if constexpr (BKF) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      int r = n0 + (tid >> 2) + 64 * e;
      r = r < a.N ? r : a.N - 1;
      pb[e] = B + (long)r * a.sb_n + 4 * (tid & 3);
    }
    b_adv = 16;
  } else {
    int r = n0 + 4 * (tid & 31);
    r = r + 3 < a.N ? r : a.N - 4;
#pragma unroll
    for (int e = 0; e < 2; ++e) pb[e] = B + (long)((tid >> 5) + 8 * e) * a.sb_k + r;
    b_adv = 16 * a.sb_k;
  }
  f32x4 ra[2], rb[2];
  auto load_tile = [&]() {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      ra[e] = *reinterpret_cast<const f32x4*>(pa[e]);
      rb[e] = *reinterpret_cast<const f32x4*>(pb[e]);
      pa[e] += a_adv;
      pb[e] += b_adv;
    }
  };
  // LDS image [idx][g][kk] with k = 4 kk + g (see gemm_f32_mfma_kernel)
  auto store_tile = [&](int buf) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      if constexpr (AK) {
        float* d = &As[buf][((tid >> 2) + 64 * e) * LDK + (tid & 3)];          // k = 4 c + j  ->  g = j, kk = c
#pragma unroll
        for (int j = 0; j < 4; ++j) d[4 * j] = ra[e][j];
      } else {
        *reinterpret_cast<f32x4*>(&As[buf][((tid >> 5) + 8 * e) * LDI + 4 * (tid & 31)]) = ra[e];
      }
      if constexpr (BKF) {
        float* d = &Bs[buf][((tid >> 2) + 64 * e) * LDK + (tid & 3)];
#pragma unroll
        for (int j = 0; j < 4; ++j) d[4 * j] = rb[e][j];
      } else {
        *reinterpret_cast<f32x4*>(&Bs[buf][((tid >> 5) + 8 * e) * LDI + 4 * (tid & 31)]) = rb[e];
      }
    }
  };

  f32x4 acc[WT][WT];
#pragma unroll
  for (int i = 0; i < WT; ++i)
#pragma unroll
    for (int j = 0; j < WT; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

  const int nk = a.K >> 4;
  const int frag = (lane & 15) * LDK + (lane >> 4) * 4;
  load_tile();
  store_tile(0);
  __syncthreads();
  for (int kt = 0; kt < nk; ++kt) {
    const int cur = kt & 1;
    if (kt + 1 < nk) load_tile();
    f32x4 af[WT], bf[WT];
#pragma unroll
    for (int i = 0; i < WT; ++i) {
      if constexpr (AK) {
        af[i] = *reinterpret_cast<const f32x4*>(&As[cur][(wr * WT + i) * 16 * LDK + frag]);
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) af[i][kk] = As[cur][(4 * kk + (lane >> 4)) * LDI + (wr * WT + i) * 16 + (lane & 15)];
      }
      if constexpr (BKF) {
        bf[i] = *reinterpret_cast<const f32x4*>(&Bs[cur][(wc * WT + i) * 16 * LDK + frag]);
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) bf[i][kk] = Bs[cur][(4 * kk + (lane >> 4)) * LDI + (wc * WT + i) * 16 + (lane & 15)];
      }
    }
#pragma unroll
    for (int kk = 0; kk < 4; ++kk)
#pragma unroll
      for (int i = 0; i < WT; ++i)
#pragma unroll
        for (int j = 0; j < WT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bf[j][kk], af[i][kk], acc[i][j], 0, 0, 0);
    if (kt + 1 < nk) store_tile(cur ^ 1);
    __syncthreads();
  }
  f32_mfma_epilogue<EPI, WT>(a, C, acc, m0, n0, wr, wc, lane);
}

// one wave per row; row kept in registers when cols <= 64*VPL, else re-read (cols here: 197..257)
__global__ __launch_bounds__(256) void softmax_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long rows,
                                                          int cols, float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float* xr = x + row * cols;
    float* yr = y + row * cols;
    float mx = -INFINITY;
    for (int c = lane; c < cols; c += 64) mx = fmaxf(mx, xr[c] * scale);
    mx = wave_max(mx);
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += expf(xr[c] * scale - mx);
    s = wave_sum(s);
    const float inv = 1.0f / s;
    for (int c = lane; c < cols; c += 64) yr[c] = expf(xr[c] * scale - mx) * inv;
  }
}

__global__ __launch_bounds__(256) void softmax_bwd_kernel(const float* __restrict__ y, const float* __restrict__ dy,
                                                          float* __restrict__ dx, long rows, int cols, float scale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float* yr = y + row * cols;
    const float* dr = dy + row * cols;
    float* o = dx + row * cols;
    float s = 0.f;
    for (int c = lane; c < cols; c += 64) s += yr[c] * dr[c];
    s = wave_sum(s);
    for (int c = lane; c < cols; c += 64) o[c] = scale * yr[c] * (dr[c] - s);
  }
}

// test / tuning hook: 1 = FMA kernel, 0 = matrix cores (initialised from MV_GEMM_F32=fma)
std::atomic<int> g_f32_fma{getenv("MV_GEMM_F32") && !strcmp(getenv("MV_GEMM_F32"), "fma") ? 1 : 0};
// 2 (mv_gemm_f32_force_fma(2)): matrix cores, but only the generic kernel (A/B of the fast form)
std::atomic<int> g_f32_generic{0};

}  // namespace

extern "C" int mv_gemm_f32_force_fma(int on) {
  g_f32_fma.store(on == 1 ? 1 : 0, std::memory_order_relaxed);
  g_f32_generic.store(on == 2 ? 1 : 0, std::memory_order_relaxed);
  return MV_OK;
}

extern "C" int mv_gemm_f32(const float* A, long sa_m, long sa_k, long sa_b1, long sa_b2, const float* B, long sb_k,
                           long sb_n, long sb_b1, long sb_b2, float* C, long sc_m, long sc_n, long sc_b1, long sc_b2,
                           int M, int N, int K, int nb1, int nb2, float alpha, int accumulate, const float* bias,
                           int epilogue, const float* aux, long ld_aux, int aux_i, float* out2, long ld_out2,
                           mv_stream_t stream) {
  MV_REQUIRE(M >= 0 && N >= 0 && K >= 0 && nb1 >= 1 && nb2 >= 1, MV_ERR_SHAPE);
  if (M == 0 || N == 0) return MV_OK;
  MV_REQUIRE((long)nb1 * nb2 <= 65535, MV_ERR_SHAPE);
  F32Args a{A, sa_m, sa_k, sa_b1, sa_b2, B, sb_k, sb_n, sb_b1, sb_b2, C, sc_m, sc_n, sc_b1, sc_b2,
            M, N, K, nb2, alpha, accumulate, bias, aux, ld_aux, aux_i, out2, ld_out2};
  hipStream_t s = (hipStream_t)stream;
  // Matrix-core form (bit-identical results).  128 x 128 blocks when that wastes little of the last row / column of blocks,
  // 64 x 64 otherwise (M = N = 197: 77 % useful instead of 59 %).  mv_gemm_f32_force_fma(1) / MV_GEMM_F32=fma keeps the FMA kernel.
  if (!g_f32_fma.load(std::memory_order_relaxed)) {
    const long pad128 = (long)mv_cdiv(M, 128) * 128 * mv_cdiv(N, 128) * 128, pad64 = (long)mv_cdiv(M, 64) * 64 * mv_cdiv(N, 64) * 64;
    const bool big = pad128 * 10 <= pad64 * 11;             // at most 10 % more padded work than the 64-tiles need
    // fast form: K % 16 == 0, one unit stride per operand, everything 16-byte addressable
    const bool ak = sa_k == 1, am = sa_m == 1, bk = sb_k == 1, bn = sb_n == 1;
    auto al4 = [](long v) { return (v & 3) == 0; };
    const bool fast = big && K >= 16 && K % 16 == 0 && M >= 4 && N >= 4 && (ak || am) && (bk || bn) && mv_aligned16(A) &&
                      mv_aligned16(B) && al4(sa_b1) && al4(sa_b2) && al4(sb_b1) && al4(sb_b2) &&
                      (ak ? al4(sa_m) : (al4(sa_k) && M % 4 == 0)) && (bk ? al4(sb_n) : (al4(sb_k) && N % 4 == 0)) &&
                      !g_f32_generic.load(std::memory_order_relaxed);
    if (fast) {
      const dim3 fgrid(mv_cdiv(N, 128), mv_cdiv(M, 128), nb1 * nb2);
#define MV_F32_FAST(E)                                                                     \
      if (ak && bk) gemm_f32_mfma_fast_kernel<E, true, true><<<fgrid, 256, 0, s>>>(a);     \
      else if (ak) gemm_f32_mfma_fast_kernel<E, true, false><<<fgrid, 256, 0, s>>>(a);     \
      else if (bk) gemm_f32_mfma_fast_kernel<E, false, true><<<fgrid, 256, 0, s>>>(a);     \
      else gemm_f32_mfma_fast_kernel<E, false, false><<<fgrid, 256, 0, s>>>(a);
      switch (epilogue) {
        case MV_EPI_NONE: MV_F32_FAST(MV_EPI_NONE) break;
        case MV_EPI_GELU: MV_F32_FAST(MV_EPI_GELU) break;
        case MV_EPI_RESIDUAL:
          MV_REQUIRE(aux, MV_ERR_UNSUPPORTED);
          MV_F32_FAST(MV_EPI_RESIDUAL) break;
        case MV_EPI_DGELU:
          MV_REQUIRE(aux, MV_ERR_UNSUPPORTED);
          MV_F32_FAST(MV_EPI_DGELU) break;
        case MV_EPI_EMBED:
          MV_REQUIRE(aux && aux_i > 0, MV_ERR_UNSUPPORTED);
          MV_F32_FAST(MV_EPI_EMBED) break;
        default: return MV_ERR_UNSUPPORTED;
      }
#undef MV_F32_FAST
      MV_CHECK_LAUNCH();
      return MV_OK;
    }
#define MV_F32_LAUNCH(E)                                                                                    \
    if (big) gemm_f32_mfma_kernel<E, 4><<<dim3(mv_cdiv(N, 128), mv_cdiv(M, 128), nb1 * nb2), 256, 0, s>>>(a); \
    else gemm_f32_mfma_kernel<E, 2><<<dim3(mv_cdiv(N, 64), mv_cdiv(M, 64), nb1 * nb2), 256, 0, s>>>(a);
    switch (epilogue) {
      case MV_EPI_NONE: MV_F32_LAUNCH(MV_EPI_NONE) break;
      case MV_EPI_GELU: MV_F32_LAUNCH(MV_EPI_GELU) break;
      case MV_EPI_RESIDUAL:
        MV_REQUIRE(aux, MV_ERR_UNSUPPORTED);
        MV_F32_LAUNCH(MV_EPI_RESIDUAL) break;
      case MV_EPI_DGELU:
        MV_REQUIRE(aux, MV_ERR_UNSUPPORTED);
        MV_F32_LAUNCH(MV_EPI_DGELU) break;
      case MV_EPI_EMBED:
        MV_REQUIRE(aux && aux_i > 0, MV_ERR_UNSUPPORTED);
        MV_F32_LAUNCH(MV_EPI_EMBED) break;
      default: return MV_ERR_UNSUPPORTED;
    }
#undef MV_F32_LAUNCH
    MV_CHECK_LAUNCH();
    return MV_OK;
  }
  dim3 grid(mv_cdiv(N, TN), mv_cdiv(M, TM), nb1 * nb2);
  switch (epilogue) {
    case MV_EPI_NONE: gemm_f32_kernel<MV_EPI_NONE><<<grid, 256, 0, s>>>(a); break;
    case MV_EPI_GELU: gemm_f32_kernel<MV_EPI_GELU><<<grid, 256, 0, s>>>(a); break;
    case MV_EPI_RESIDUAL:
      MV_REQUIRE(aux, MV_ERR_UNSUPPORTED);
      gemm_f32_kernel<MV_EPI_RESIDUAL><<<grid, 256, 0, s>>>(a);
      break;
    case MV_EPI_DGELU:
      MV_REQUIRE(aux, MV_ERR_UNSUPPORTED);
      gemm_f32_kernel<MV_EPI_DGELU><<<grid, 256, 0, s>>>(a);
      break;
    case MV_EPI_EMBED:
      MV_REQUIRE(aux && aux_i > 0, MV_ERR_UNSUPPORTED);
      gemm_f32_kernel<MV_EPI_EMBED><<<grid, 256, 0, s>>>(a);
      break;
    default: return MV_ERR_UNSUPPORTED;
  }
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_softmax_fwd(const float* x, float* y, long rows, int cols, float scale, mv_stream_t stream) {
  MV_REQUIRE(rows >= 0 && cols > 0, MV_ERR_SHAPE);
  if (rows == 0) return MV_OK;
  long g = (rows + 3) / 4;
  if (g > 4096) g = 4096;
  softmax_fwd_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>(x, y, rows, cols, scale);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_softmax_bwd(const float* y, const float* dy, float* dx, long rows, int cols, float scale,
                              mv_stream_t stream) {
  MV_REQUIRE(rows >= 0 && cols > 0, MV_ERR_SHAPE);
  if (rows == 0) return MV_OK;
  long g = (rows + 3) / 4;
  if (g > 4096) g = 4096;
  softmax_bwd_kernel<<<(int)g, 256, 0, (hipStream_t)stream>>>(y, dy, dx, rows, cols, scale);
  MV_CHECK_LAUNCH();
  return MV_OK;
}
