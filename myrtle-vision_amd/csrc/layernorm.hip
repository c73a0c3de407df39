// LayerNorm forward/backward for the ViT residual stream (reference: nn.LayerNorm(dim), eps 1e-5,
// vit.py:37,41 PreNorm; :332,:353 decoder norms).
//
// HBM-bound.  One wave64 per row: the row lives in registers (VPL float4 per lane), statistics are
// two-pass in registers (mean, then centred sum of squares -- matches ATen's accuracy) reduced with
// wave shuffles; no LDS, no re-read.  Algorithmic bytes per row: forward 4*dim (x) + sizeof(y)*dim;
// backward 4*dim (x) + sizeof(dy)*dim + 4*dim (dx) [+ 4*dim dx_add].
#include "mv_common.h"

namespace {

// YT = int8_t: the output goes straight into the NEXT layer's quint8 quantiser (q_inv = 1 / scale, q_zp) and leaves as int8
// MFMA operands q - 128: the fp32 LayerNorm output of the converted int8 model (620 MB at batch 1024) is neither written
// nor read back.  Same expressions as the float path followed by affine_code_one: bit-identical to LayerNorm + quantiser.
// NSEG = 3 / 6 (round 4; YT = bf16_t): the output leaves as the bf16 PIECES of the split-operand Linear products (mv_split2_bf16 /
// mv_split3_bf16, role 0: p0 p0 p1 [p0 p1 p2], `dim` columns apart in rows of NSEG * dim) -- the fp32 LayerNorm output is neither
// written nor read back by a split pass (8 B per element of traffic and a launch per LayerNorm in precision fp32 / bf16x3 / bf16x3h).
// Same expressions, then the same piece arithmetic as split3_kernel: bit-identical to LayerNorm + split.
template <int VPL, typename YT, int NSEG = 0>
__global__ __launch_bounds__(256) void ln_fwd_kernel(const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, YT* __restrict__ y,
                                                     float* __restrict__ mean, float* __restrict__ rstd, int rows,
                                                     int dim, float eps, float q_inv = 0.f, float q_zp = 0.f) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nvec = dim >> 2;
  const float inv_dim = 1.0f / (float)dim;
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
    float4 v[VPL];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      v[i] = (c < nvec) ? xr[c] : make_float4(0.f, 0.f, 0.f, 0.f);
      s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
    }
    const float mu = wave_sum(s) * inv_dim;
    float q = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        const float a = v[i].x - mu, b = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
        q += (a * a + b * b) + (cc * cc + d * d);
      }
    }
    const float rs = rsqrtf(wave_sum(q) * inv_dim + eps);
    if (lane == 0 && mean) {
      mean[row] = mu;
      rstd[row] = rs;
    }
    YT* yr = y + row * (long)dim * (NSEG ? NSEG : 1);
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        const float4 g = reinterpret_cast<const float4*>(gamma)[c];
        const float4 b = reinterpret_cast<const float4*>(beta)[c];
        const float o0 = (v[i].x - mu) * rs * g.x + b.x, o1 = (v[i].y - mu) * rs * g.y + b.y;
        const float o2 = (v[i].z - mu) * rs * g.z + b.z, o3 = (v[i].w - mu) * rs * g.w + b.w;
        if constexpr (NSEG != 0) {
          const float o[4] = {o0, o1, o2, o3};
          bf16x4 p[3];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bf16_t p0 = (bf16_t)o[e];
            const float r1 = o[e] - (float)p0;
            const bf16_t p1 = (bf16_t)r1;
            p[0][e] = p0;
            p[1][e] = p1;
            p[2][e] = (bf16_t)(r1 - (float)p1);
          }
          constexpr int order[6] = {0, 0, 1, 0, 1, 2};
#pragma unroll
          for (int sgm = 0; sgm < NSEG; ++sgm) reinterpret_cast<bf16x4*>(yr + (long)sgm * dim)[c] = p[order[sgm]];
        } else if constexpr (sizeof(YT) == 1) {
          reinterpret_cast<unsigned*>(yr)[c] = affine_i8_pack4<0>(o0, o1, o2, o3, q_inv, q_zp);
        } else if constexpr (sizeof(YT) == 4) {
          reinterpret_cast<float4*>(yr)[c] = make_float4(o0, o1, o2, o3);
        } else {
          bf16x4 o = {(bf16_t)o0, (bf16_t)o1, (bf16_t)o2, (bf16_t)o3};
          reinterpret_cast<bf16x4*>(yr)[c] = o;
        }
      }
    }
  }
}

// Backward.  dx = rstd * (g - mean(g) - xhat * mean(g * xhat)),  g = dy * gamma  (+ dx_add).
// Per-lane partial dgamma/dbeta are kept in registers across the rows a wave visits (a lane owns the
// same columns for every row), reduced across the block's 4 waves through LDS, and written as one
// partial row per block; ln_bwd_finish sums the partial rows (deterministic, no atomics).
// Optional extras for the fused transformer blocks (backward order: this dx IS the next consumer's incoming gradient):
//   dx16  : bf16 copy of dx (the MFMA operand of that consumer's dX / dW products) -- replaces a separate cast pass;
//   colsum: third group of per-column partial sums, sum_rows dx = the consumer's output-projection bias gradient.
template <int VPL, typename DT>
__global__ __launch_bounds__(256) void ln_bwd_kernel(const DT* __restrict__ dy, const float* __restrict__ x, long ldx,
                                                     const float* __restrict__ gamma, const float* __restrict__ mean,
                                                     const float* __restrict__ rstd, const float* dx_add, float* dx,
                                                     long lddx, float* __restrict__ partial, int rows, int dim,
                                                     bf16_t* __restrict__ dx16, int want_colsum, int dx16_nseg = 0) {
  // dx16_nseg = 3 / 6 (round 4): dx16 receives the bf16 PIECES of dx (mv_split2_bf16 / mv_split3_bf16 role 0, rows of nseg * dim)
  // instead of one bf16 copy -- the dY operand of the consumer's split-operand products, without its split pass over dx
  __shared__ float red[4][VPL * 64 * 4 * 2];
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nvec = dim >> 2;
  const float inv_dim = 1.0f / (float)dim;
  float4 g4[VPL], dg[VPL], db[VPL], dc[VPL];
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    const int c = lane + 64 * i;
    g4[i] = (c < nvec) ? reinterpret_cast<const float4*>(gamma)[c] : make_float4(0.f, 0.f, 0.f, 0.f);
    dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    dc[i] = make_float4(0.f, 0.f, 0.f, 0.f);
  }
  for (long row = (long)blockIdx.x * 4 + wave; row < rows; row += (long)gridDim.x * 4) {
    const float4* xr = reinterpret_cast<const float4*>(x + row * ldx);
    const DT* dyr = dy + row * (long)dim;
    const float mu = mean[row], rs = rstd[row];
    float4 xh[VPL], gg[VPL];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      float4 d = make_float4(0.f, 0.f, 0.f, 0.f), xv = make_float4(mu, mu, mu, mu);
      if (c < nvec) {
        xv = xr[c];
        if constexpr (sizeof(DT) == 4) {
          d = reinterpret_cast<const float4*>(dyr)[c];
        } else {
          const bf16x4 t = reinterpret_cast<const bf16x4*>(dyr)[c];
          d = make_float4((float)t[0], (float)t[1], (float)t[2], (float)t[3]);
        }
      }
      xh[i] = make_float4((xv.x - mu) * rs, (xv.y - mu) * rs, (xv.z - mu) * rs, (xv.w - mu) * rs);
      gg[i] = make_float4(d.x * g4[i].x, d.y * g4[i].y, d.z * g4[i].z, d.w * g4[i].w);
      s1 += (gg[i].x + gg[i].y) + (gg[i].z + gg[i].w);
      s2 += (gg[i].x * xh[i].x + gg[i].y * xh[i].y) + (gg[i].z * xh[i].z + gg[i].w * xh[i].w);
      dg[i].x += d.x * xh[i].x; dg[i].y += d.y * xh[i].y; dg[i].z += d.z * xh[i].z; dg[i].w += d.w * xh[i].w;
      db[i].x += d.x; db[i].y += d.y; db[i].z += d.z; db[i].w += d.w;
    }
    const float c1 = wave_sum(s1) * inv_dim;
    const float c2 = wave_sum(s2) * inv_dim;
    float4* dxr = reinterpret_cast<float4*>(dx + row * lddx);
    const float4* addr = dx_add ? reinterpret_cast<const float4*>(dx_add + row * lddx) : nullptr;
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      const int c = lane + 64 * i;
      if (c < nvec) {
        float4 o = make_float4(rs * (gg[i].x - c1 - xh[i].x * c2), rs * (gg[i].y - c1 - xh[i].y * c2),
                               rs * (gg[i].z - c1 - xh[i].z * c2), rs * (gg[i].w - c1 - xh[i].w * c2));
        if (addr) {
          const float4 a = addr[c];
          o.x += a.x; o.y += a.y; o.z += a.z; o.w += a.w;
        }
        dxr[c] = o;
        if (dx16 && dx16_nseg == 0) {
          bf16x4 o16 = {(bf16_t)o.x, (bf16_t)o.y, (bf16_t)o.z, (bf16_t)o.w};
          reinterpret_cast<bf16x4*>(dx16 + row * (long)dim)[c] = o16;
        } else if (dx16) {
          const float ov[4] = {o.x, o.y, o.z, o.w};
          bf16x4 p[3];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bf16_t p0 = (bf16_t)ov[e];
            const float r1 = ov[e] - (float)p0;
            const bf16_t p1 = (bf16_t)r1;
            p[0][e] = p0;
            p[1][e] = p1;
            p[2][e] = (bf16_t)(r1 - (float)p1);
          }
          bf16x4* pr = reinterpret_cast<bf16x4*>(dx16 + row * (long)dim * dx16_nseg) + c;
          const int nv = dim >> 2;                          // segments are dim elements = nv vectors apart
          pr[0] = p[0]; pr[nv] = p[0]; pr[2 * nv] = p[1];
          if (dx16_nseg == 6) { pr[3 * nv] = p[0]; pr[4 * nv] = p[1]; pr[5 * nv] = p[2]; }
        }
        dc[i].x += o.x; dc[i].y += o.y; dc[i].z += o.z; dc[i].w += o.w;
      }
    }
  }
  // block reduction of the per-wave column partials
#pragma unroll
  for (int i = 0; i < VPL; ++i) {
    float* r = &red[wave][(i * 64 + lane) * 8];
    r[0] = dg[i].x; r[1] = dg[i].y; r[2] = dg[i].z; r[3] = dg[i].w;
    r[4] = db[i].x; r[5] = db[i].y; r[6] = db[i].z; r[7] = db[i].w;
  }
  __syncthreads();
  const int groups = want_colsum ? 3 : 2;
  float* prow = partial + (long)blockIdx.x * groups * dim;
  for (int idx = threadIdx.x; idx < VPL * 64 * 8; idx += 256) {
    const int vec = idx >> 3, e = idx & 7;       // vec = i*64 + lane -> column block c = lane + 64*i
    const int i = vec >> 6, l = vec & 63;
    const int c = l + 64 * i;
    if (c < nvec) {
      const float s = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
      const int col = c * 4 + (e & 3);
      prow[(e >> 2) * dim + col] = s;
    }
  }
  if (want_colsum) {                             // third group through the same LDS buffer
    __syncthreads();
#pragma unroll
    for (int i = 0; i < VPL; ++i) {
      float* r = &red[wave][(i * 64 + lane) * 8];
      r[0] = dc[i].x; r[1] = dc[i].y; r[2] = dc[i].z; r[3] = dc[i].w;
    }
    __syncthreads();
    for (int idx = threadIdx.x; idx < VPL * 64 * 8; idx += 256) {
      const int vec = idx >> 3, e = idx & 7;
      const int i = vec >> 6, l = vec & 63;
      const int c = l + 64 * i;
      if (c < nvec && e < 4) prow[2 * dim + c * 4 + e] = (red[0][idx] + red[1][idx]) + (red[2][idx] + red[3][idx]);
    }
  }
}

int ln_grid(int rows) {
  int g = mv_cdiv(rows, 4);
  return g < 1 ? 1 : (g > 1024 ? 1024 : g);
}

}  // namespace

extern "C" size_t mv_layernorm_bwd_workspace_bytes(int rows, int dim) {
  return (size_t)ln_grid(rows) * 3 * (size_t)dim * sizeof(float);
}

// launch with the grid capped at the workgroups resident at once (MV_RESIDENT_BLOCKS: the wide and the split-output forms keep
// fewer than the eight groups per CU that the 2 048 cap assumes)
#define LN_GO(args_, ...)                                                        \
  do {                                                                           \
    const int g_ = min(grid, MV_RESIDENT_BLOCKS((__VA_ARGS__), 256, 0));         \
    __VA_ARGS__<<<g_, 256, 0, s>>> args_;                                        \
  } while (0)
#define LN_FWD_CASE(V)                                                                                      \
  case V:                                                                                                   \
    if (y_dtype == MV_F32)                                                                                  \
      LN_GO((x, ldx, gamma, beta, (float*)y, mean, rstd, rows, dim, eps), ln_fwd_kernel<V, float>); \
    else                                                                                                    \
      LN_GO((x, ldx, gamma, beta, (bf16_t*)y, mean, rstd, rows, dim, eps), ln_fwd_kernel<V, bf16_t>); \
    break;

extern "C" int mv_layernorm_fwd(const float* x, long ldx, const float* gamma, const float* beta, void* y, int y_dtype,
                                float* mean, float* rstd, int rows, int dim, float eps, mv_stream_t stream) {
  MV_REQUIRE(rows >= 0 && dim > 0 && dim % 4 == 0 && dim <= 4096 && ldx % 4 == 0, MV_ERR_SHAPE);
  MV_REQUIRE(y_dtype == MV_F32 || y_dtype == MV_BF16, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(mv_aligned16(x) && mv_aligned16(gamma) && mv_aligned16(beta) && mv_aligned16(y), MV_ERR_ALIGN);
  if (rows == 0) return MV_OK;
  hipStream_t s = (hipStream_t)stream;
  int grid = mv_cdiv(rows, 4);
  if (grid > 2048) grid = 2048;  // grid-stride beyond 8 blocks per CU
  const int vpl = mv_cdiv(dim / 4, 64);
  switch (vpl) {
    LN_FWD_CASE(1) LN_FWD_CASE(2) LN_FWD_CASE(3) LN_FWD_CASE(4)
    case 5: case 6: case 7: case 8:
      if (y_dtype == MV_F32)
        LN_GO((x, ldx, gamma, beta, (float*)y, mean, rstd, rows, dim, eps), ln_fwd_kernel<8, float>);
      else
        LN_GO((x, ldx, gamma, beta, (bf16_t*)y, mean, rstd, rows, dim, eps), ln_fwd_kernel<8, bf16_t>);
      break;
    default:
      if (y_dtype == MV_F32)
        LN_GO((x, ldx, gamma, beta, (float*)y, mean, rstd, rows, dim, eps), ln_fwd_kernel<16, float>);
      else
        LN_GO((x, ldx, gamma, beta, (bf16_t*)y, mean, rstd, rows, dim, eps), ln_fwd_kernel<16, bf16_t>);
  }
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_layernorm_fwd_split(const float* x, long ldx, const float* gamma, const float* beta, void* y_split, int nseg,
                                      float* mean, float* rstd, int rows, int dim, float eps, mv_stream_t stream) {
  MV_REQUIRE(rows >= 0 && dim > 0 && dim % 4 == 0 && dim <= 1024 && ldx % 4 == 0, MV_ERR_SHAPE);
  MV_REQUIRE(nseg == 3 || nseg == 6, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(mv_aligned16(x) && mv_aligned16(gamma) && mv_aligned16(beta) && mv_aligned16(y_split), MV_ERR_ALIGN);
  if (rows == 0) return MV_OK;
  hipStream_t s = (hipStream_t)stream;
  int grid = mv_cdiv(rows, 4);
  if (grid > 2048) grid = 2048;
  bf16_t* y = (bf16_t*)y_split;
#define LN_SPLIT_CASE(V)                                                                                               \
  case V:                                                                                                              \
    if (nseg == 3) LN_GO((x, ldx, gamma, beta, y, mean, rstd, rows, dim, eps), ln_fwd_kernel<V, bf16_t, 3>); \
    else LN_GO((x, ldx, gamma, beta, y, mean, rstd, rows, dim, eps), ln_fwd_kernel<V, bf16_t, 6>);           \
    break;
  switch (mv_cdiv(dim / 4, 64)) {
    LN_SPLIT_CASE(1) LN_SPLIT_CASE(2) LN_SPLIT_CASE(3)
    default:
      if (nseg == 3) LN_GO((x, ldx, gamma, beta, y, mean, rstd, rows, dim, eps), ln_fwd_kernel<4, bf16_t, 3>);
      else LN_GO((x, ldx, gamma, beta, y, mean, rstd, rows, dim, eps), ln_fwd_kernel<4, bf16_t, 6>);
  }
#undef LN_SPLIT_CASE
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_layernorm_fwd_q8(const float* x, long ldx, const float* gamma, const float* beta, void* codes, int rows,
                                   int dim, float eps, float scale, int zero_point, mv_stream_t stream) {
  MV_REQUIRE(rows >= 0 && dim > 0 && dim % 16 == 0 && dim <= 1024 && ldx % 4 == 0 && scale > 0.f, MV_ERR_SHAPE);
  MV_REQUIRE(zero_point >= 0 && zero_point <= 255, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(mv_aligned16(x) && mv_aligned16(gamma) && mv_aligned16(beta) && mv_aligned16(codes), MV_ERR_ALIGN);
  if (rows == 0) return MV_OK;
  hipStream_t s = (hipStream_t)stream;
  int grid = mv_cdiv(rows, 4);
  if (grid > 2048) grid = 2048;
  const float inv = 1.0f / scale, fz = (float)zero_point;
  switch (mv_cdiv(dim / 4, 64)) {
    case 1: ln_fwd_kernel<1, int8_t><<<grid, 256, 0, s>>>(x, ldx, gamma, beta, (int8_t*)codes, nullptr, nullptr, rows, dim, eps, inv, fz); break;
    case 2: ln_fwd_kernel<2, int8_t><<<grid, 256, 0, s>>>(x, ldx, gamma, beta, (int8_t*)codes, nullptr, nullptr, rows, dim, eps, inv, fz); break;
    case 3: ln_fwd_kernel<3, int8_t><<<grid, 256, 0, s>>>(x, ldx, gamma, beta, (int8_t*)codes, nullptr, nullptr, rows, dim, eps, inv, fz); break;
    default: ln_fwd_kernel<4, int8_t><<<grid, 256, 0, s>>>(x, ldx, gamma, beta, (int8_t*)codes, nullptr, nullptr, rows, dim, eps, inv, fz); break;
  }
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// grid: every workgroup resident at once (the kernel keeps 142 registers at dim 768: three groups per CU, not the four that
// ln_grid's 1 024 assumes)
#define LN_BWD_LAUNCH(V)                                                                                     \
  if (dy_dtype == MV_F32) {                                                                                  \
    grid = min(grid, MV_RESIDENT_BLOCKS((ln_bwd_kernel<V, float>), 256, 0));                                 \
    ln_bwd_kernel<V, float><<<grid, 256, 0, s>>>((const float*)dy, x, ldx, gamma, mean, rstd, dx_add, dx, lddx, \
                                                 workspace, rows, dim, (bf16_t*)dx_bf16, dx_colsum != nullptr, nseg); \
  } else {                                                                                                   \
    grid = min(grid, MV_RESIDENT_BLOCKS((ln_bwd_kernel<V, bf16_t>), 256, 0));                                \
    ln_bwd_kernel<V, bf16_t><<<grid, 256, 0, s>>>((const bf16_t*)dy, x, ldx, gamma, mean, rstd, dx_add, dx, lddx, \
                                                  workspace, rows, dim, (bf16_t*)dx_bf16, dx_colsum != nullptr, nseg); \
  }

namespace {
int ln_bwd_impl(const void* dy, int dy_dtype, const float* x, long ldx, const float* gamma, const float* mean, const float* rstd,
                const float* dx_add, float* dx, long lddx, float* dgamma, float* dbeta, int accumulate, float* workspace,
                size_t workspace_bytes, int rows, int dim, void* dx_bf16, float* dx_colsum, int nseg, mv_stream_t stream);
}
extern "C" int mv_layernorm_bwd(const void* dy, int dy_dtype, const float* x, long ldx, const float* gamma,
                                const float* mean, const float* rstd, const float* dx_add, float* dx, long lddx,
                                float* dgamma, float* dbeta, int accumulate, float* workspace, size_t workspace_bytes,
                                int rows, int dim, void* dx_bf16, float* dx_colsum, mv_stream_t stream) {
  return ln_bwd_impl(dy, dy_dtype, x, ldx, gamma, mean, rstd, dx_add, dx, lddx, dgamma, dbeta, accumulate, workspace, workspace_bytes,
                     rows, dim, dx_bf16, dx_colsum, 0, stream);
}
extern "C" int mv_layernorm_bwd_split(const void* dy, int dy_dtype, const float* x, long ldx, const float* gamma,
                                      const float* mean, const float* rstd, const float* dx_add, float* dx, long lddx,
                                      float* dgamma, float* dbeta, int accumulate, float* workspace, size_t workspace_bytes,
                                      int rows, int dim, void* dx_split, int nseg, float* dx_colsum, mv_stream_t stream) {
  MV_REQUIRE((nseg == 3 || nseg == 6) && dx_split && mv_aligned16(dx_split), MV_ERR_UNSUPPORTED);
  return ln_bwd_impl(dy, dy_dtype, x, ldx, gamma, mean, rstd, dx_add, dx, lddx, dgamma, dbeta, accumulate, workspace, workspace_bytes,
                     rows, dim, dx_split, dx_colsum, nseg, stream);
}
namespace {
int ln_bwd_impl(const void* dy, int dy_dtype, const float* x, long ldx, const float* gamma, const float* mean, const float* rstd,
                const float* dx_add, float* dx, long lddx, float* dgamma, float* dbeta, int accumulate, float* workspace,
                size_t workspace_bytes, int rows, int dim, void* dx_bf16, float* dx_colsum, int nseg, mv_stream_t stream) {
  MV_REQUIRE(rows >= 0 && dim > 0 && dim % 4 == 0 && dim <= 2048 && ldx % 4 == 0 && lddx % 4 == 0, MV_ERR_SHAPE);
  MV_REQUIRE(dy_dtype == MV_F32 || dy_dtype == MV_BF16, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(mv_aligned16(x) && mv_aligned16(gamma) && mv_aligned16(dy) && mv_aligned16(dx) &&
                 (!dx_add || mv_aligned16(dx_add)),
             MV_ERR_ALIGN);
  MV_REQUIRE(workspace_bytes >= mv_layernorm_bwd_workspace_bytes(rows, dim), MV_ERR_WORKSPACE);
  hipStream_t s = (hipStream_t)stream;
  int grid = ln_grid(rows);
  if (rows > 0) {
    const int vpl = mv_cdiv(dim / 4, 64);
    switch (vpl) {
      case 1: LN_BWD_LAUNCH(1) break;
      case 2: LN_BWD_LAUNCH(2) break;
      case 3: LN_BWD_LAUNCH(3) break;
      case 4: LN_BWD_LAUNCH(4) break;
      default: LN_BWD_LAUNCH(8) break;
    }
    MV_CHECK_LAUNCH();
  }
  const int groups = dx_colsum ? 3 : 2;
  // ONE finishing launch for all (2 or 3) column groups of the workspace: dgamma | dbeta | dx column sums.  (accumulate
  // applies to dgamma/dbeta; the column sums are always overwritten: their group is reduced into a zero-initialised
  // view only when accumulate == 0, otherwise it takes its own launch.)
  if (dx_colsum && accumulate) {
    mv_reduce_rows_kernel<<<mv_reduce_rows_grid(2 * dim), 1024, 0, s>>>(workspace, rows > 0 ? grid : 0, 2 * dim, (long)groups * dim,
                                                                        dgamma, dbeta, dbeta, dim, 2 * dim, 1);
    MV_CHECK_LAUNCH();
    mv_reduce_rows_kernel<<<mv_reduce_rows_grid(dim), 1024, 0, s>>>(workspace + 2 * dim, rows > 0 ? grid : 0, dim,
                                                                    (long)groups * dim, dx_colsum, dx_colsum, dx_colsum, dim, dim, 0);
  } else {
    mv_reduce_rows_kernel<<<mv_reduce_rows_grid(groups * dim), 1024, 0, s>>>(workspace, rows > 0 ? grid : 0, groups * dim,
                                                                             (long)groups * dim, dgamma, dbeta, dx_colsum, dim,
                                                                             2 * dim, accumulate);
  }
  MV_CHECK_LAUNCH();
  return MV_OK;
}
}  // namespace
