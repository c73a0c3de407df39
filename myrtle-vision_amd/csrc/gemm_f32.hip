// Generic strided, batched fp32 contraction + row softmax: the exact-parity (precision="fp32") path.
//
// Every product of the hot path can be expressed through element strides, so this one kernel covers
// nn.Linear forward / dX / dW and the materialised attention products q k^T, p v and their gradients
// (vit.py:92-96) when bit-level agreement with the fp32 reference matters more than speed, or when a
// forward hook on Attention.attn_output (vit.py:80-82,94) needs the probabilities in memory.
// Arithmetic: one fp32 FMA chain per output in k order (same as an ATen/BLAS fp32 dot up to ordering).
// 64x64 tile, K-step 16, 256 threads x (4x4) outputs, LDS-staged, staging map chosen per operand so the
// unit-stride axis is the one consecutive lanes walk.
#include "mv_common.h"

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

}  // namespace

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
  dim3 grid(mv_cdiv(N, TN), mv_cdiv(M, TM), nb1 * nb2);
  hipStream_t s = (hipStream_t)stream;
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
