// HBM-bound elementwise / layout kernels of the ViT hot path (gfx950).  All are grid-stride, 16 bytes per
// lane where the layout allows it.
#include "mv_common.h"

namespace {

inline int ew_grid(long n_items, int per_block = 256) {
  long g = (n_items + per_block - 1) / per_block;
  if (g < 1) g = 1;
  if (g > 4096) g = 4096;  // 16 blocks per CU, grid-stride beyond
  return (int)g;
}

// ---- patchify (vit.py:271-275): out[(b*gh+gy)*gw+gx][(py*p+px)*C + c] = img[b][c][gy*p+py][gx*p+px] ----
// One thread produces the C values of 4 consecutive px (4*C consecutive outputs): its C reads are float4 loads
// from C planes (coalesced along x), its writes are 4*C consecutive elements.
template <typename T, int C>
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int H,
                                                       int W, int p) {
  const int gw = W / p, gh = H / p;
  const int p4 = p >> 2;
  const long total = (long)B * H * (W >> 2);  // one item = 4 pixels of one row
  for (long it = (long)blockIdx.x * 256 + threadIdx.x; it < total; it += (long)gridDim.x * 256) {
    const int x4 = (int)(it % (W >> 2));
    const long t = it / (W >> 2);
    const int y = (int)(t % H);
    const int b = (int)(t / H);
    const int gx = x4 / p4, px = (x4 - gx * p4) * 4;
    const int gy = y / p, py = y - gy * p;
    float v[C][4];
#pragma unroll
    for (int c = 0; c < C; ++c) {
      const float4 f = *reinterpret_cast<const float4*>(img + (((long)b * C + c) * H + y) * W + x4 * 4);
      v[c][0] = f.x; v[c][1] = f.y; v[c][2] = f.z; v[c][3] = f.w;
    }
    T* o = out + (((long)b * gh + gy) * gw + gx) * ((long)p * p * C) + ((long)py * p + px) * C;
#pragma unroll
    for (int e = 0; e < 4; ++e)
#pragma unroll
      for (int c = 0; c < C; ++c) o[e * C + c] = (T)v[c][e];
  }
}

template <typename T>
__global__ void patchify_generic_kernel(const float* __restrict__ img, T* __restrict__ out, int B, int C, int H, int W,
                                        int p) {
  const int gw = W / p, gh = H / p;
  const long total = (long)B * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // iterate in OUTPUT order so writes coalesce
    const long pd = (long)p * p * C;
    const long row = i / pd;
    const int k = (int)(i - row * pd);
    const int c = k % C, pp = k / C, px = pp % p, py = pp / p;
    const int gx = (int)(row % gw);
    const long t = row / gw;
    const int gy = (int)(t % gh), b = (int)(t / gh);
    out[i] = (T)img[(((long)b * C + c) * H + gy * p + py) * W + gx * p + px];
  }
}

__global__ void embed_cls_kernel(const float* __restrict__ cls, const float* __restrict__ pos, float* __restrict__ x,
                                 int B, int T, int D) {
  const long total = (long)B * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d = (int)(i % D);
    const long b = i / D;
    x[b * T * D + d] = cls[d] + pos[d];
  }
}

// dpos[t][d] = sum_b dx[b][t][d]; dcls[d] = dpos[0][d]
__global__ void embed_bwd_kernel(const float* __restrict__ dx, float* dpos, float* dcls, int accumulate, int B, int T,
                                 int D) {
  const long total = (long)T * D;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float s = 0.f;
    for (int b = 0; b < B; ++b) s += dx[(long)b * total + i];
    if (dpos) dpos[i] = accumulate ? dpos[i] + s : s;
    if (dcls && i < D) dcls[i] = accumulate ? dcls[i] + s : s;
  }
}

template <typename T>
__global__ void gather_patch_rows_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int Tk, int D) {
  const int D4 = D >> 2;
  const long total = (long)B * (Tk - 1) * D4;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int d4 = (int)(i % D4);
    const long r = i / D4;
    const long b = r / (Tk - 1), t = r - b * (Tk - 1);
    const float4 v = *reinterpret_cast<const float4*>(src + ((b * Tk + t + 1) * (long)D) + d4 * 4);
    T* o = dst + r * D + d4 * 4;
    if constexpr (sizeof(T) == 4) {
      *reinterpret_cast<float4*>(o) = v;
    } else {
      bf16x4 w = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
      *reinterpret_cast<bf16x4*>(o) = w;
    }
  }
}

template <typename S, typename T>
__global__ void cast_kernel(const S* __restrict__ src, T* __restrict__ dst, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float v[4];
    if constexpr (sizeof(S) == 4) {
      const float4 f = reinterpret_cast<const float4*>(src)[i];
      v[0] = f.x; v[1] = f.y; v[2] = f.z; v[3] = f.w;
    } else {
      const bf16x4 f = reinterpret_cast<const bf16x4*>(src)[i];
      v[0] = (float)f[0]; v[1] = (float)f[1]; v[2] = (float)f[2]; v[3] = (float)f[3];
    }
    if constexpr (sizeof(T) == 4) {
      reinterpret_cast<float4*>(dst)[i] = make_float4(v[0], v[1], v[2], v[3]);
    } else {
      bf16x4 w = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
      reinterpret_cast<bf16x4*>(dst)[i] = w;
    }
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = (T)(float)src[i];
}

// fp32 -> IEEE half (the operands of the "bf16x3" attention core); n % 4 handled like cast_kernel
__global__ void cast_f16_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, long n) {
  typedef __attribute__((ext_vector_type(4))) _Float16 h4;
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 f = reinterpret_cast<const float4*>(src)[i];
    const h4 w = {(_Float16)f.x, (_Float16)f.y, (_Float16)f.z, (_Float16)f.w};
    reinterpret_cast<h4*>(dst)[i] = w;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dst[i] = (_Float16)src[i];
}

// w fp32 [R][C] -> wb bf16 [R][ldw] (pad columns zero) and wt bf16 [C][ldt] (pad zero).  32x32 tile through LDS.
__device__ __forceinline__ void weight_prep_tile(const float* __restrict__ w, bf16_t* wb, int ldw, bf16_t* wt, int ldt, int R,
                                                 int C, int r0, int c0, float (*tile)[33]) {
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + tx;
    const float v = (r < R && c < C) ? w[(long)r * C + c] : 0.f;
    tile[ty + 8 * i][tx] = v;
    if (wb && r < R && c < ldw) wb[(long)r * ldw + c] = (bf16_t)v;
  }
  __syncthreads();
  if (wt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int c = c0 + ty + 8 * i, r = r0 + tx;  // wt[c][r]
      if (c < C && r < ldt) wt[(long)c * ldt + r] = (bf16_t)tile[tx][ty + 8 * i];
    }
  }
}
__global__ __launch_bounds__(256) void weight_prep_kernel(const float* __restrict__ w, bf16_t* wb, int ldw, bf16_t* wt,
                                                          int ldt, int R, int C) {
  __shared__ float tile[32][33];
  weight_prep_tile(w, wb, ldw, wt, ldt, R, C, blockIdx.y * 32, blockIdx.x * 32, tile);
}
// 64x64 tile, two adjacent elements per lane: 8-byte loads (when the rows of w are 8-byte aligned, i.e. C even) and 4-byte
// stores into both copies.  Needs ldw and ldt even (pad8 makes them multiples of 8).
__device__ __forceinline__ void weight_prep_tile64(const float* __restrict__ w, bf16_t* __restrict__ wb, int ldw,
                                                   bf16_t* __restrict__ wt, int ldt, int R, int C, int r0, int c0,
                                                   float (*tile)[65]) {
  typedef __attribute__((ext_vector_type(2))) bf16_t bf16x2;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const bool v2 = (C & 1) == 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + 2 * tx;
    float a = 0.f, b = 0.f;
    if (r < R) {
      const float* p = w + (long)r * C + c;
      if (v2 && c + 1 < C) {
        const float2 v = *reinterpret_cast<const float2*>(p);
        a = v.x;
        b = v.y;
      } else {
        if (c < C) a = p[0];
        if (c + 1 < C) b = p[1];
      }
    }
    tile[ty + 8 * i][2 * tx] = a;
    tile[ty + 8 * i][2 * tx + 1] = b;
    if (wb && r < R && c < ldw) *reinterpret_cast<bf16x2*>(wb + (long)r * ldw + c) = bf16x2{(bf16_t)a, (bf16_t)b};
  }
  __syncthreads();
  if (wt) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + ty + 8 * i, r = r0 + 2 * tx;      // wt[c][r], wt[c][r + 1]
      if (c < C && r < ldt)
        *reinterpret_cast<bf16x2*>(wt + (long)c * ldt + r) = bf16x2{(bf16_t)tile[2 * tx][ty + 8 * i], (bf16_t)tile[2 * tx + 1][ty + 8 * i]};
    }
  }
}
__global__ __launch_bounds__(256) void weight_prep64_kernel(const float* __restrict__ w, bf16_t* wb, int ldw, bf16_t* wt,
                                                            int ldt, int R, int C) {
  __shared__ float tile[64][65];
  weight_prep_tile64(w, wb, ldw, wt, ldt, R, C, blockIdx.y * 64, blockIdx.x * 64, tile);
}
// All the Linear weights of a model in ONE launch (50 launches of ~7 us each per training step otherwise): block b
// belongs to the item with the largest first_block <= b (binary search over the table, which is a few KB and L2-hot).
__global__ __launch_bounds__(256) void weight_prep_batch_kernel(const mv_weight_prep_item* __restrict__ items, int count) {
  __shared__ float tile[64][65];
  const int b = blockIdx.x;
  int lo = 0, hi = count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (items[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const mv_weight_prep_item it = items[lo];
  const int t = b - it.first_block, ty = t / it.tiles_x, tx = t - ty * it.tiles_x;
  weight_prep_tile64(it.w, (bf16_t*)it.w_bf16, it.ldw, (bf16_t*)it.wt_bf16, it.ldt, it.R, it.C, ty * 64, tx * 64, tile);
}

// The right-operand (role 1) bf16 pieces of an nn.Linear weight for the split-operand products, BOTH layouts from one read of w:
//   fwd [R, NSEG * C]: the pieces of w      (y  = x W^T),   segments C apart:  b0 b1 b0 (b2 b1 b0)
//   dx  [C, NSEG * R]: the pieces of w^T    (dx = dy W),    segments R apart
// (mv_split2_bf16 / mv_split3_bf16 role 1 of w and of a transposed copy of w: the same values, without the copy and the second pass).
template <int NSEG>
__global__ __launch_bounds__(256) void weight_split_kernel(const float* __restrict__ w, bf16_t* __restrict__ fwd,
                                                           bf16_t* __restrict__ dx, int R, int C) {
  typedef __attribute__((ext_vector_type(2))) bf16_t bf16x2;
  __shared__ float tile[64][65];
  constexpr int order[6] = {0, 1, 0, 2, 1, 0};
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  auto pieces = [](float v, bf16_t (&p)[3]) {
    p[0] = (bf16_t)v;
    const float r1 = v - (float)p[0];
    p[1] = (bf16_t)r1;
    p[2] = (bf16_t)(r1 - (float)p[1]);
  };
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int r = r0 + ty + 8 * i, c = c0 + 2 * tx;       // C, R even (host): a pair is inside or outside together
    float a = 0.f, b = 0.f;
    if (r < R && c < C) {
      const float2 v = *reinterpret_cast<const float2*>(w + (long)r * C + c);
      a = v.x;
      b = v.y;
    }
    tile[ty + 8 * i][2 * tx] = a;
    tile[ty + 8 * i][2 * tx + 1] = b;
    if (fwd && r < R && c < C) {
      bf16_t pa[3], pb[3];
      pieces(a, pa);
      pieces(b, pb);
#pragma unroll
      for (int sg = 0; sg < NSEG; ++sg)
        *reinterpret_cast<bf16x2*>(fwd + (long)r * NSEG * C + (long)sg * C + c) = bf16x2{pa[order[sg]], pb[order[sg]]};
    }
  }
  __syncthreads();
  if (dx) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const int c = c0 + ty + 8 * i, r = r0 + 2 * tx;      // dx[c][sg * R + r], dx[c][sg * R + r + 1]
      if (c < C && r < R) {
        bf16_t pa[3], pb[3];
        pieces(tile[2 * tx][ty + 8 * i], pa);
        pieces(tile[2 * tx + 1][ty + 8 * i], pb);
#pragma unroll
        for (int sg = 0; sg < NSEG; ++sg)
          *reinterpret_cast<bf16x2*>(dx + (long)c * NSEG * R + (long)sg * R + r) = bf16x2{pa[order[sg]], pb[order[sg]]};
      }
    }
  }
}

template <typename T>
__global__ void gelu_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = (T)gelu_f((float)x[i]);
}
template <typename T>
__global__ void gelu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dx[i] = (T)((float)dy[i] * dgelu_f((float)x[i]));
}
__global__ void add_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 x = reinterpret_cast<const float4*>(a)[i], y = reinterpret_cast<const float4*>(b)[i];
    reinterpret_cast<float4*>(o)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    o[i] = a[i] + b[i];
}

// ---- fake quantisation: fused quant+dequant, fp32 in/out (utils/quantize.py:46-72,84; algorithm: qtorch 0.3.0
// float_quantize_nearest / fixed_point_quantize_nearest, restated in oracle/quant_oracle.py) ----
__device__ __forceinline__ float quant_float_one(float a, int exp_bits, int man_bits) {
  const unsigned bits = __float_as_uint(a);
  const unsigned sign = bits & 0x80000000u;
  const unsigned mask = (1u << (23 - man_bits)) - 1u;
  const unsigned half = 1u << (23 - man_bits - 1);
  const int texp = (int)((bits & 0x7FFFFFFFu) >> 23) - 127;
  const int min_exp = -((1 << (exp_bits - 1)) - 2);
  if (texp < min_exp) {  // target-subnormal range: round on the grid of spacing 2^(min_exp - man)
    const float shift = __uint_as_float(((unsigned)(127 + min_exp) << 23) | sign);
    const float val = a + shift;
    const unsigned qb = (__float_as_uint(val) + half) & ~mask;
    return __uint_as_float(qb) - shift;
  }
  unsigned q = (bits + half) & ~mask;
  const int qexp = (int)((q & 0x7FFFFFFFu) >> 23);
  const int max_store = (1 << (exp_bits - 1)) - 1 + 127;
  const int min_store = min_exp + 127;
  if (qexp > max_store) {
    const unsigned max_man = (0x7FFFFFu >> (23 - man_bits)) << (23 - man_bits);
    q = sign | ((unsigned)max_store << 23) | max_man;
  } else if (qexp < min_store && q != 0u) {
    const unsigned mag = q & 0x7FFFFFFFu;
    q = mag > ((unsigned)(min_store - 1) << 23) ? (sign | ((unsigned)min_store << 23)) : 0u;
  }
  return __uint_as_float(q);
}

__global__ void quant_float_kernel(const float* __restrict__ x, float* __restrict__ y, long n, int e, int m) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<float4*>(y)[i] = make_float4(quant_float_one(v.x, e, m), quant_float_one(v.y, e, m),
                                                  quant_float_one(v.z, e, m), quant_float_one(v.w, e, m));
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    y[i] = quant_float_one(x[i], e, m);
}

__global__ void quant_fixed_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float scale, float inv,
                                   float tmin, float tmax, int clamp) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float r = floorf(x[i] * scale + 0.5f) * inv;
    if (clamp) r = fminf(fmaxf(r, tmin), tmax);
    y[i] = r;
  }
}

__global__ void quant_affine_kernel(const float* __restrict__ x, float* __restrict__ y, long n, float scale, float inv,
                                    int zp, int qmin, int qmax) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float q = rintf(x[i] * inv) + (float)zp;  // rintf = round-half-even = std::nearbyint default mode
    q = fminf(fmaxf(q, (float)qmin), (float)qmax);
    y[i] = (q - (float)zp) * scale;
  }
}

// Integer CODES of the same quantiser, re-centred: c = clamp(rint(x / scale) + zp, qmin, qmax) - zp, an integer in
// [qmin - zp, qmax - zp] (|c| <= 255 for 8-bit), written as bf16 -- exactly representable, so a bf16 MFMA GEMM of two
// code tensors with fp32 accumulation IS the integer dot product (mv_gemm_nt_bf16_scaled applies scale_x * scale_w).
// Rows are written with leading dimension ld (>= cols, padding zeroed) so the result is a valid MFMA operand.
// (affine_code_one / affine_i8_pack4: mv_common.h -- shared with the fused producers)
template <int PRE, typename XT>
__global__ void quant_affine_codes_kernel(const XT* __restrict__ x, bf16_t* __restrict__ y, long rows, int cols, int ld,
                                          float inv, int zp, int qmin, int qmax) {
  const long n = rows * ld;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / ld;
    const int c = (int)(i - r * ld);
    y[i] = (bf16_t)(c < cols ? affine_code_one<PRE>((float)x[r * cols + c], inv, (float)zp, (float)qmin, (float)qmax) : 0.f);
  }
}
// cols % 4 == 0 and ld % 4 == 0: one row per block iteration, 16-byte loads and 8-byte stores, no division
template <int PRE, typename XT>
__global__ __launch_bounds__(256) void quant_affine_codes_vec_kernel(const XT* __restrict__ x, bf16_t* __restrict__ y,
                                                                     long rows, int cols, int ld, float inv, int zp,
                                                                     int qmin, int qmax) {
  const float fz = (float)zp, lo = (float)qmin, hi = (float)qmax;
  const int c4 = cols >> 2, l4 = ld >> 2;
  for (long r = blockIdx.x; r < rows; r += gridDim.x) {
    const XT* xr = x + r * cols;
    bf16x4* yr = reinterpret_cast<bf16x4*>(y + r * ld);
    for (int j = threadIdx.x; j < l4; j += 256) {
      bf16x4 o = {(bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f, (bf16_t)0.f};
      if (j < c4) {
        float4 v;
        if constexpr (sizeof(XT) == 4) {
          v = reinterpret_cast<const float4*>(xr)[j];
        } else {
          const bf16x4 h = reinterpret_cast<const bf16x4*>(xr)[j];
          v = make_float4((float)h[0], (float)h[1], (float)h[2], (float)h[3]);
        }
        o = bf16x4{(bf16_t)affine_code_one<PRE>(v.x, inv, fz, lo, hi), (bf16_t)affine_code_one<PRE>(v.y, inv, fz, lo, hi),
                   (bf16_t)affine_code_one<PRE>(v.z, inv, fz, lo, hi), (bf16_t)affine_code_one<PRE>(v.w, inv, fz, lo, hi)};
      }
      yr[j] = o;
    }
  }
}

// int8 form for the v_mfma_i32_16x16x64_i8 GEMM: y = q - 128 (q the uint8 quantiser output in [0, 255]), 16 codes = one
// 16-byte store per thread iteration; columns [cols, ld) are zero-filled (they meet zero weight codes)
template <int PRE, typename XT>
__global__ __launch_bounds__(256) void quant_affine_i8_kernel(const XT* __restrict__ x, int8_t* __restrict__ y, long rows,
                                                              int cols, int ld, float inv, int zp) {
  const float fz = (float)zp;
  const int l16 = ld >> 4;
  for (long r = blockIdx.x; r < rows; r += gridDim.x) {
    const XT* xr = x + r * cols;
    u32x4* yr = reinterpret_cast<u32x4*>(y + r * ld);
    for (int j = threadIdx.x; j < l16; j += 256) {
      unsigned w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int c = 16 * j + e;
        int code = 0;
        if (c < cols) code = (int)(affine_code_one<PRE>((float)xr[c], inv, fz, 0.f, 255.f) + fz) - 128;   // q - 128
        w[e >> 2] |= ((unsigned)code & 0xFFu) << (8 * (e & 3));
      }
      yr[j] = (u32x4){w[0], w[1], w[2], w[3]};
    }
  }
}

// flat form (ld == cols: no padding columns): a wave takes 1024 consecutive elements per iteration; lane l reads the four
// 16-byte groups 4 l + 256 k (coalesced 1 KiB per load instruction) and stores one packed dword per group (256 B per
// store instruction).  The row-wise kernel above reads 64 B per lane at a 64-byte lane stride: 3.5x off the HBM rate.
template <int PRE, typename XT>
__global__ __launch_bounds__(256) void quant_affine_i8_flat_kernel(const XT* __restrict__ x, unsigned* __restrict__ y, long n,
                                                                   float inv, int zp) {
  const float fz = (float)zp;
  const long nchunk = n >> 10;                      // n % 1024 == 0 (dispatch)
  const int lane = threadIdx.x & 63;
  for (long ch = (long)blockIdx.x * 4 + (threadIdx.x >> 6); ch < nchunk; ch += (long)gridDim.x * 4) {
    const XT* xc = x + (ch << 10);
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if constexpr (sizeof(XT) == 4) {
        v[k] = reinterpret_cast<const float4*>(xc)[lane + 64 * k];
      } else {
        const bf16x4 hh = reinterpret_cast<const bf16x4*>(xc)[lane + 64 * k];
        v[k] = make_float4((float)hh[0], (float)hh[1], (float)hh[2], (float)hh[3]);
      }
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const float e[4] = {v[k].x, v[k].y, v[k].z, v[k].w};
      unsigned w = 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int code = (int)(affine_code_one<PRE>(e[j], inv, fz, 0.f, 255.f) + fz) - 128;
        w |= ((unsigned)code & 0xFFu) << (8 * j);
      }
      y[(ch << 8) + lane + 64 * k] = w;
    }
  }
}

// order-preserving float <-> uint map so min/max can use integer atomics
__device__ __forceinline__ unsigned f2ord(float f) {
  const unsigned u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float ord2f(unsigned o) {
  return __uint_as_float((o & 0x80000000u) ? (o & 0x7FFFFFFFu) : ~o);
}
__global__ __launch_bounds__(256) void minmax_kernel(const float* __restrict__ x, long n, unsigned* ord) {
  float mn = INFINITY, mxv = -INFINITY;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const float v = x[i];
    mn = fminf(mn, v);
    mxv = fmaxf(mxv, v);
  }
  mn = -wave_max(-mn);
  mxv = wave_max(mxv);
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&ord[0], f2ord(mn));
    atomicMax(&ord[1], f2ord(mxv));
  }
}
__global__ void minmax_begin_kernel(const float* minmax, unsigned* ord) {
  ord[0] = f2ord(minmax[0]);
  ord[1] = f2ord(minmax[1]);
}
__global__ void minmax_end_kernel(float* minmax, const unsigned* ord) {
  minmax[0] = ord2f(ord[0]);
  minmax[1] = ord2f(ord[1]);
}

// ---- cross entropy, mean reduction.  One wave per sample; classes strided by `inner` -------------------
// ---- labels: torch.nn.CrossEntropyLoss() semantics.  ignore_index = -100 contributes no loss and no gradient and is left
// out of the mean's denominator; any OTHER label outside [0, C) is an error (torch device-asserts): it is never used as an
// index, and it poisons the returned loss with NaN (no host synchronisation needed to notice).
// stat[4] = {mean loss, number of counted labels, number of bad labels, unused}; stat[1..2] are filled by ce_label_scan_kernel
// before the loss kernel runs on the same stream.
constexpr long MV_IGNORE_INDEX = -100;
__global__ __launch_bounds__(256) void ce_label_scan_kernel(const int64_t* __restrict__ labels, long total, int C,
                                                            float* __restrict__ stat) {
  float ok = 0.f, bad = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int64_t y = labels[i];
    if (y >= 0 && y < C) ok += 1.f;
    else if (y != MV_IGNORE_INDEX) bad += 1.f;
  }
  ok = wave_sum(ok);
  bad = wave_sum(bad);
  if ((threadIdx.x & 63) == 0) {                 // counts are integers below 2^24 per wave: fp32 atomics are exact and
    if (ok != 0.f) atomicAdd(stat + 1, ok);      // order-independent up to 2^24 labels in total (12.8 M at 256 x 224^2)
    if (bad != 0.f) atomicAdd(stat + 2, bad);
  }
}
__device__ __forceinline__ float ce_inv_count(const float* stat) {
  const float n = stat[1];
  return n > 0.f ? 1.0f / n : __builtin_nanf("");      // every label ignored: torch returns nan as well
}
__device__ __forceinline__ float ce_poison(const float* stat, float v) { return stat[2] > 0.f ? __builtin_nanf("") : v; }

template <typename DT>
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits,
                                                            const int64_t* __restrict__ labels, float* loss_sum,
                                                            DT* dlogits, int ld_dl, int64_t* argmax, long outer, int C,
                                                            long inner, float gscale) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long total = outer * inner;
  const float inv_count = ce_inv_count(loss_sum);
  float block_loss = 0.f;
  for (long smp = (long)blockIdx.x * 4 + wave; smp < total; smp += (long)gridDim.x * 4) {
    const long o = smp / inner, in = smp - o * inner;
    const float* lp = logits + o * C * inner + in;
    float mx = -INFINITY;
    int am = 0;
    for (int c = lane; c < C; c += 64) {
      const float v = lp[(long)c * inner];
      if (v > mx) { mx = v; am = c; }
    }
    // wave arg-max, first index wins on ties (torch.argmax semantics)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      const float om = __shfl_xor(mx, off, 64);
      const int oa = __shfl_xor(am, off, 64);
      if (om > mx || (om == mx && oa < am)) { mx = om; am = oa; }
    }
    float s = 0.f;
    for (int c = lane; c < C; c += 64) s += expf(lp[(long)c * inner] - mx);
    s = wave_sum(s);
    const int64_t yl = labels[smp];
    const bool valid = yl >= 0 && yl < C;         // ignore_index (-100) and out-of-range labels: no loss, no gradient
    const int y = valid ? (int)yl : -1;
    const float lse = mx + logf(s);
    if (lane == 0) {
      if (valid) block_loss += lse - lp[(long)y * inner];
      if (argmax) argmax[smp] = am;
    }
    if (dlogits) {
      const float inv = 1.0f / s;
      if (inner == 1) {
        DT* dp = dlogits + o * ld_dl;
        for (int c = lane; c < ld_dl; c += 64) {
          float gq = 0.f;
          if (c < C && valid) gq = (expf(lp[c] - mx) * inv - (c == y ? 1.f : 0.f)) * inv_count * gscale;
          dp[c] = (DT)gq;
        }
      } else {
        DT* dp = dlogits + o * C * inner + in;
        for (int c = lane; c < C; c += 64)
          dp[(long)c * inner] =
              (DT)(valid ? (expf(lp[(long)c * inner] - mx) * inv - (c == y ? 1.f : 0.f)) * inv_count * gscale : 0.f);
      }
    }
  }
  __shared__ float red[4];
  if (lane == 0) red[wave] = block_loss;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_sum, ce_poison(loss_sum, ((red[0] + red[1]) + (red[2] + red[3])) * inv_count));
}

// segmentation layout (inner > 1): one THREAD per pixel, classes in a register loop, so that consecutive lanes
// read consecutive pixels of one class plane (coalesced); C is small (17)
template <typename DT>
__global__ __launch_bounds__(256) void cross_entropy_pixel_kernel(const float* __restrict__ logits,
                                                                  const int64_t* __restrict__ labels, float* loss_sum,
                                                                  DT* dlogits, int64_t* argmax, long outer, int C,
                                                                  long inner, float gscale) {
  const long total = outer * inner;
  const float inv_count = ce_inv_count(loss_sum);
  float my_loss = 0.f;
  for (long smp = (long)blockIdx.x * 256 + threadIdx.x; smp < total; smp += (long)gridDim.x * 256) {
    const long o = smp / inner, in = smp - o * inner;
    const float* lp = logits + o * C * inner + in;
    float mx = -INFINITY;
    int am = 0;
    for (int c = 0; c < C; ++c) {
      const float v = lp[(long)c * inner];
      if (v > mx) { mx = v; am = c; }
    }
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(lp[(long)c * inner] - mx);
    const int64_t yl = labels[smp];
    const bool valid = yl >= 0 && yl < C;
    const int y = valid ? (int)yl : -1;
    if (valid) my_loss += mx + logf(s) - lp[(long)y * inner];
    if (argmax) argmax[smp] = am;
    if (dlogits) {
      const float inv = 1.0f / s;
      DT* dp = dlogits + o * C * inner + in;
      for (int c = 0; c < C; ++c)
        dp[(long)c * inner] =
            (DT)(valid ? (expf(lp[(long)c * inner] - mx) * inv - (c == y ? 1.f : 0.f)) * inv_count * gscale : 0.f);
    }
  }
  my_loss = wave_sum(my_loss);
  __shared__ float red[4];
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = my_loss;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(loss_sum, ce_poison(loss_sum, ((red[0] + red[1]) + (red[2] + red[3])) * inv_count));
}

// ---- bilinear upsample, align_corners=False (ATen upsample_bilinear2d): src = (dst + 0.5) * scale - 0.5, clamped at 0
__device__ __forceinline__ void bilin_src(int d, float scale, int in_size, int& i0, int& i1, float& l1) {
  float s = ((float)d + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}
__global__ void upsample_fwd_kernel(const float* __restrict__ in, long sb, long sc, long sp, float* __restrict__ out,
                                    int B, int C, int h, int w, int H, int W, float sh, float sw) {
  const long total = (long)B * C * H * W;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int X = (int)(i % W);
    const long t = i / W;
    const int Y = (int)(t % H);
    const long bc = t / H;
    const int c = (int)(bc % C);
    const long b = bc / C;
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_src(Y, sh, h, y0, y1, ly);
    bilin_src(X, sw, w, x0, x1, lx);
    const float* p = in + b * sb + c * sc;
    const float top = p[(long)(y0 * w + x0) * sp] * (1.f - lx) + p[(long)(y0 * w + x1) * sp] * lx;
    const float bot = p[(long)(y1 * w + x0) * sp] * (1.f - lx) + p[(long)(y1 * w + x1) * sp] * lx;
    out[i] = top * (1.f - ly) + bot * ly;
  }
}
// backward in gather form: each input pixel sums the (few) output pixels that reference it -> deterministic.
__global__ void upsample_bwd_kernel(const float* __restrict__ dout, float* __restrict__ din, long sb, long sc, long sp,
                                    int B, int C, int h, int w, int H, int W, float sh, float sw) {
  const long total = (long)B * C * h * w;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    // iterate with c fastest when the small tensor is channels-last (sc == 1) so writes coalesce
    int x, y, c;
    long b;
    if (sc == 1) {
      c = (int)(i % C);
      const long t = i / C;
      x = (int)(t % w);
      const long t2 = t / w;
      y = (int)(t2 % h);
      b = t2 / h;
    } else {
      x = (int)(i % w);
      const long t = i / w;
      y = (int)(t % h);
      const long bc = t / h;
      c = (int)(bc % C);
      b = bc / C;
    }
    // candidate output rows/cols: a superset of those whose (y0|y1) / (x0|x1) can equal y / x
    int Ylo = (int)floorf(((float)y - 1.f + 0.5f) / sh - 0.5f) - 1, Yhi = (int)ceilf(((float)y + 1.f + 0.5f) / sh - 0.5f) + 1;
    int Xlo = (int)floorf(((float)x - 1.f + 0.5f) / sw - 0.5f) - 1, Xhi = (int)ceilf(((float)x + 1.f + 0.5f) / sw - 0.5f) + 1;
    if (Ylo < 0) Ylo = 0;
    if (Xlo < 0) Xlo = 0;
    if (Yhi > H - 1) Yhi = H - 1;
    if (Xhi > W - 1) Xhi = W - 1;
    float s = 0.f;
    const float* dp = dout + (b * C + c) * (long)H * W;
    for (int Y = Ylo; Y <= Yhi; ++Y) {
      int y0, y1; float ly;
      bilin_src(Y, sh, h, y0, y1, ly);
      float wy = 0.f;
      if (y0 == y) wy += 1.f - ly;
      if (y1 == y) wy += ly;
      if (wy == 0.f) continue;
      for (int X = Xlo; X <= Xhi; ++X) {
        int x0, x1; float lx;
        bilin_src(X, sw, w, x0, x1, lx);
        float wx = 0.f;
        if (x0 == x) wx += 1.f - lx;
        if (x1 == x) wx += lx;
        if (wx != 0.f) s += wy * wx * dp[(long)Y * W + X];
      }
    }
    din[b * sb + c * sc + (long)(y * w + x) * sp] = s;
  }
}

// ---- out[i] (+)= sum over s of slabs[s * stride + i], fixed order (deterministic): the reduce of a K-split product ----
// (`add`: optional addend read before the slabs; may alias out)
__global__ __launch_bounds__(256) void sum_slabs_kernel(const float* __restrict__ slabs, long stride, int S,
                                                        const float* add, float* out, long n) {
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 t = add ? reinterpret_cast<const float4*>(add)[i] : make_float4(0.f, 0.f, 0.f, 0.f);
    for (int k = 0; k < S; ++k) {
      const float4 v = reinterpret_cast<const float4*>(slabs + (long)k * stride)[i];
      t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w;
    }
    reinterpret_cast<float4*>(out)[i] = t;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    float t = add ? add[i] : 0.f;
    for (int k = 0; k < S; ++k) t += slabs[(long)k * stride + i];
    out[i] = t;
  }
}

// ---- gradient clipping (torch.nn.utils.clip_grad_norm_, classification/train.py:265-270) over the flat gradient arena:
// two-stage deterministic sum of squares (fp32 per thread, fp64 across threads and blocks), then
// out[0] = total_norm = sqrt(sum) * grad_scale, out[1] = clip coefficient = min(1, max_norm / (total_norm + 1e-6)).
// The coefficient stays on the device and is folded into the AdamW kernel (no pass over the gradients, no host sync).
constexpr int GN_PARTS = 1024;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, long n, double* __restrict__ part) {
  const long n4 = n >> 2;
  float s = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) s += g[i] * g[i];
  __shared__ double red[256];
  red[threadIdx.x] = (double)s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[blockIdx.x] = red[0];
}
__global__ __launch_bounds__(256) void grad_norm_finish_kernel(const double* __restrict__ part, float gscale, float max_norm,
                                                               float* __restrict__ out) {
  __shared__ double red[256];
  double s = 0.0;
  for (int i = threadIdx.x; i < GN_PARTS; i += 256) s += part[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float total = (float)sqrt(red[0]) * gscale;
    const float coef = max_norm / (total + 1e-6f);
    out[0] = total;
    out[1] = coef < 1.f ? coef : 1.f;
  }
}

// ---- dropout (nn.Dropout, vit.py:50,52,75,311): y = x * keep / (1 - p), keep ~ Bernoulli(1 - p) from Philox4x32-10.
// Element i takes word i & 3 of the Philox block with counter (i >> 2, offset) and key seed: the mask is a pure function of
// (seed, offset, i), so the backward regenerates it from the same two numbers instead of storing it.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                              unsigned (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
template <typename T>
__global__ __launch_bounds__(256) void dropout_kernel(const T* __restrict__ x, T* __restrict__ y, long n, unsigned thr,
                                                      float scale, uint64_t seed, uint64_t offset) {
  const long n4 = (n + 3) >> 2;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    unsigned r[4];
    philox4x32_10((unsigned)i, (unsigned)((uint64_t)i >> 32), (unsigned)offset, (unsigned)(offset >> 32), (unsigned)seed,
                  (unsigned)(seed >> 32), r);
    const long e0 = i << 2;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e0 + e < n) y[e0 + e] = (T)(r[e] >= thr ? (float)x[e0 + e] * scale : 0.f);
  }
}

// ---- AdamW (torch.optim.AdamW semantics): p *= 1 - lr*wd; m,v update; p -= lr/bc1 * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                             float* __restrict__ v, long n, float lr, float b1, float b2, float eps, float wd, float bc1,
                             float bc2, float gscale, const float* __restrict__ clip_coef, const float* __restrict__ hyper) {
  if (hyper) {                                // mv_adamw_dev: the per-step scalars live on the device (captured launches stay valid)
    lr = hyper[0];
    bc1 = hyper[1];
    bc2 = hyper[2];
  }
  if (clip_coef) gscale *= clip_coef[0];      // clip_grad_norm_'s min(1, max_norm / (norm + 1e-6)), computed on the device
  const float step = lr / bc1, inv_sqrt_bc2 = rsqrtf(bc2);
  const long n4 = n >> 2;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    float4 P = reinterpret_cast<float4*>(p)[i], M = reinterpret_cast<float4*>(m)[i], V = reinterpret_cast<float4*>(v)[i];
    const float4 G = reinterpret_cast<const float4*>(g)[i];
    float* pp = &P.x; float* mm = &M.x; float* vv = &V.x; const float* gg = &G.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gr = gg[e] * gscale;
      pp[e] *= 1.f - lr * wd;
      mm[e] = b1 * mm[e] + (1.f - b1) * gr;
      vv[e] = b2 * vv[e] + (1.f - b2) * gr * gr;
      pp[e] -= step * mm[e] / (sqrtf(vv[e]) * inv_sqrt_bc2 + eps);
    }
    reinterpret_cast<float4*>(p)[i] = P;
    reinterpret_cast<float4*>(m)[i] = M;
    reinterpret_cast<float4*>(v)[i] = V;
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const float gr = g[i] * gscale;
    float P = p[i] * (1.f - lr * wd);
    const float M = b1 * m[i] + (1.f - b1) * gr, V = b2 * v[i] + (1.f - b2) * gr * gr;
    P -= step * M / (sqrtf(V) * inv_sqrt_bc2 + eps);
    p[i] = P; m[i] = M; v[i] = V;
  }
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int mv_version(void) { return 100; }
extern "C" const char* mv_error_string(int code) {
  switch (code) {
    case MV_OK: return "ok";
    case MV_ERR_SHAPE: return "shape constraint violated";
    case MV_ERR_ALIGN: return "pointer or leading dimension not aligned (16 B / multiple of 8 elements)";
    case MV_ERR_LAUNCH: return "kernel launch failed";
    case MV_ERR_UNSUPPORTED: return "dtype/epilogue combination not supported";
    case MV_ERR_WORKSPACE: return "workspace too small";
    default: return "unknown error";
  }
}

extern "C" int mv_patchify(const float* img, void* out, int out_dtype, int B, int C, int H, int W, int p,
                           mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && C > 0 && p > 0 && H % p == 0 && W % p == 0, MV_ERR_SHAPE);
  MV_REQUIRE(out_dtype == MV_F32 || out_dtype == MV_BF16, MV_ERR_UNSUPPORTED);
  if (B == 0) return MV_OK;
  if (C == 3 && p % 4 == 0 && W % 4 == 0 && mv_aligned16(img)) {
    const long items = (long)B * H * (W / 4);
    if (out_dtype == MV_F32)
      patchify_kernel<float, 3><<<ew_grid(items), 256, 0, S_>>>(img, (float*)out, B, H, W, p);
    else
      patchify_kernel<bf16_t, 3><<<ew_grid(items), 256, 0, S_>>>(img, (bf16_t*)out, B, H, W, p);
  } else {
    const long n = (long)B * C * H * W;
    if (out_dtype == MV_F32)
      patchify_generic_kernel<float><<<ew_grid(n), 256, 0, S_>>>(img, (float*)out, B, C, H, W, p);
    else
      patchify_generic_kernel<bf16_t><<<ew_grid(n), 256, 0, S_>>>(img, (bf16_t*)out, B, C, H, W, p);
  }
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_embed_cls(const float* cls, const float* pos, float* x, int B, int T, int D, mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && T > 0 && D > 0, MV_ERR_SHAPE);
  if (B == 0) return MV_OK;
  embed_cls_kernel<<<ew_grid((long)B * D), 256, 0, S_>>>(cls, pos, x, B, T, D);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_embed_bwd(const float* dx, float* dpos, float* dcls, int accumulate, int B, int T, int D,
                            mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && T > 0 && D > 0, MV_ERR_SHAPE);
  embed_bwd_kernel<<<ew_grid((long)T * D), 256, 0, S_>>>(dx, dpos, dcls, accumulate, B, T, D);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// Backward of the embedding assembly in ONE pass over dx [B, T, D] (round 4; replaces embed_bwd_kernel + gather_patch_rows_kernel:
// 103 + 36 us per ViT-B step, the first of them a 256-deep serial load chain per thread): block (t, 128-column chunk), 32 threads
// x float4 across the chunk, 8 thread rows striding the batch; every dx element is read once, leaves as dy (rows 1..T-1 of each
// image, the dY operand of the patch GEMM's dW product, in the activation dtype) and enters the batch sums dpos[t] (and dcls for
// t = 0); the 8 partial sums meet in LDS in a fixed order (deterministic).
template <typename DT>
__global__ __launch_bounds__(256) void embed_bwd_gather_kernel(const float* __restrict__ dx, DT* __restrict__ dy,
                                                               float* __restrict__ dpos, float* __restrict__ dcls, int B,
                                                               int T, int D) {
  __shared__ float4 red[8][32];
  const int chunks = (D + 127) / 128;
  const int t = blockIdx.x / chunks, ch = blockIdx.x - t * chunks;
  const int cx = threadIdx.x & 31, ry = threadIdx.x >> 5;
  const int c = ch * 128 + cx * 4;
  float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
  if (c < D) {
#pragma unroll 4
    for (int b = ry; b < B; b += 8) {
      const float4 v = *reinterpret_cast<const float4*>(dx + ((long)b * T + t) * D + c);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      if (t > 0 && dy) {
        DT* o = dy + ((long)b * (T - 1) + t - 1) * D + c;
        if constexpr (sizeof(DT) == 4) {
          *reinterpret_cast<float4*>(o) = v;
        } else {
          const bf16x4 w = {(bf16_t)v.x, (bf16_t)v.y, (bf16_t)v.z, (bf16_t)v.w};
          *reinterpret_cast<bf16x4*>(o) = w;
        }
      }
    }
  }
  red[ry][cx] = acc;
  __syncthreads();
  if (ry == 0 && c < D) {
    float4 s = red[0][cx];
#pragma unroll
    for (int r = 1; r < 8; ++r) {
      const float4 u = red[r][cx];
      s.x += u.x; s.y += u.y; s.z += u.z; s.w += u.w;
    }
    if (dpos) *reinterpret_cast<float4*>(dpos + (long)t * D + c) = s;
    if (dcls && t == 0) *reinterpret_cast<float4*>(dcls + c) = s;
  }
}

extern "C" int mv_embed_bwd_gather(const float* dx, void* dy, int dy_dtype, float* dpos, float* dcls, int B, int T, int D,
                                   mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && T > 0 && D > 0 && D % 4 == 0, MV_ERR_SHAPE);
  MV_REQUIRE(dy_dtype == MV_F32 || dy_dtype == MV_BF16, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(mv_aligned16(dx) && mv_aligned16(dy) && mv_aligned16(dpos) && mv_aligned16(dcls), MV_ERR_ALIGN);
  if (B == 0) return MV_OK;
  const int grid = T * ((D + 127) / 128);
  if (dy_dtype == MV_F32)
    embed_bwd_gather_kernel<float><<<grid, 256, 0, S_>>>(dx, (float*)dy, dpos, dcls, B, T, D);
  else
    embed_bwd_gather_kernel<bf16_t><<<grid, 256, 0, S_>>>(dx, (bf16_t*)dy, dpos, dcls, B, T, D);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_gather_patch_rows(const float* src, void* dst, int dst_dtype, int B, int T, int D,
                                    mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && T > 1 && D > 0 && D % 4 == 0, MV_ERR_SHAPE);
  MV_REQUIRE(mv_aligned16(src) && mv_aligned16(dst), MV_ERR_ALIGN);
  if (B == 0) return MV_OK;
  const long n = (long)B * (T - 1) * (D / 4);
  if (dst_dtype == MV_F32)
    gather_patch_rows_kernel<float><<<ew_grid(n), 256, 0, S_>>>(src, (float*)dst, B, T, D);
  else if (dst_dtype == MV_BF16)
    gather_patch_rows_kernel<bf16_t><<<ew_grid(n), 256, 0, S_>>>(src, (bf16_t*)dst, B, T, D);
  else
    return MV_ERR_UNSUPPORTED;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_cast(const void* src, int src_dtype, void* dst, int dst_dtype, long n, mv_stream_t stream) {
  MV_REQUIRE(n >= 0, MV_ERR_SHAPE);
  if (n == 0) return MV_OK;
  MV_REQUIRE(mv_aligned16(src) && mv_aligned16(dst), MV_ERR_ALIGN);
  const int grid = ew_grid((n + 3) / 4);
  if (src_dtype == MV_F32 && dst_dtype == MV_BF16)
    cast_kernel<float, bf16_t><<<grid, 256, 0, S_>>>((const float*)src, (bf16_t*)dst, n);
  else if (src_dtype == MV_BF16 && dst_dtype == MV_F32)
    cast_kernel<bf16_t, float><<<grid, 256, 0, S_>>>((const bf16_t*)src, (float*)dst, n);
  else if (src_dtype == MV_F32 && dst_dtype == MV_F32)
    cast_kernel<float, float><<<grid, 256, 0, S_>>>((const float*)src, (float*)dst, n);
  else if (src_dtype == MV_F32 && dst_dtype == MV_F16)
    cast_f16_kernel<<<grid, 256, 0, S_>>>((const float*)src, (_Float16*)dst, n);
  else
    return MV_ERR_UNSUPPORTED;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// fp32 -> three bf16 pieces for the bf16x6 fp32 product.  x = p0 + p1 + p2 with p0 = bf16(x), p1 = bf16(x - p0),
// p2 = bf16(x - p0 - p1) (each subtraction exact in fp32): 3 x 8 significand bits + signs cover fp32's 24, residual
// <= 2^-26 |x|.  a*b = sum over i + j <= 2 of a_i * b_j to 2^-25 relative (the three dropped cross terms), so one bf16 MFMA
// product over a SIX-segment contraction axis
//     A' = [a0 | a0 | a1 | a0 | a1 | a2]     B' = [b0 | b1 | b0 | b2 | b1 | b0]
// is an fp32-accurate product (bf16 x bf16 is exact in the MFMA's fp32 accumulator) at 1/6 of the bf16 rate, 2.7x the
// f32 MFMA's.  The kernel writes all six segments of one operand, `seg` elements apart: seg = cols with ldo = 6 * cols puts them
// side by side along K (NT operands), seg = rows * ldo stacks them along the rows (TN operands, contraction over rows).
// ---------------------------------------------------------------------------------------------------------------------
// NSEG = 3 ("bf16x3", mv_split2_bf16*): TWO pieces, the first three segments only -- A' = [a0 | a0 | a1], B' = [b0 | b1 | b0] -- i.e.
// the pairings (0,0) (0,1) (1,0): 2^-16 relative per product (the dropped a1 b1 and the third pieces) at half the MFMA work.
template <int ROLE, int NSEG = 6>
__global__ __launch_bounds__(256) void split3_kernel(const float* __restrict__ x, long ldx, bf16_t* __restrict__ out, long ldo,
                                                     long seg, long rows, int cols4) {
  const long total = rows * cols4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / cols4;
    const int c = (int)(idx - r * cols4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
    const float in[4] = {v.x, v.y, v.z, v.w};
    bf16x4 p[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bf16_t p0 = (bf16_t)in[e];
      const float r1 = in[e] - (float)p0;
      const bf16_t p1 = (bf16_t)r1;
      const float r2 = r1 - (float)p1;
      p[0][e] = p0;
      p[1][e] = p1;
      p[2][e] = (bf16_t)r2;
    }
    constexpr int order[2][6] = {{0, 0, 1, 0, 1, 2}, {0, 1, 0, 2, 1, 0}};
    bf16_t* o = out + r * ldo + c;
#pragma unroll
    for (int s = 0; s < NSEG; ++s) *reinterpret_cast<bf16x4*>(o + s * seg) = p[order[ROLE][s]];
  }
}

// The same split behind a producer's last elementwise step, with the consumer's bias gradient as a by-product: one pass
// instead of three or four over the 2304- / 3072-wide tensors of a block.  OP 0: v = x; OP 1: v = gelu(x) (fc1's activation:
// the fp32 activation itself is never stored); OP 2: v = x * gelu'(h) (fc2's dX times the activation derivative).
// A thread owns 4 columns and walks the rows of its row part (coalesced: the block's threads cover consecutive columns), so
// the column sums of v accumulate in registers; partial[part][cols] is reduced by mv_reduce_rows_kernel (fixed order).
template <int OP, int NSEG = 6>
__global__ __launch_bounds__(256) void split3_ex_kernel(const float* __restrict__ x, long ldx, const float* __restrict__ h,
                                                        long ldh, bf16_t* __restrict__ out, long ldo, long seg, long rows,
                                                        int cols4, int rows_per_part, float* __restrict__ partial) {
  const int c4 = blockIdx.x * 256 + threadIdx.x;
  if (c4 >= cols4) return;
  const int c = c4 * 4;
  const long r0 = (long)blockIdx.y * rows_per_part;
  long r1 = r0 + rows_per_part;
  if (r1 > rows) r1 = rows;
  float cs[4] = {0.f, 0.f, 0.f, 0.f};
  for (long r = r0; r < r1; ++r) {
    const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
    float in[4] = {v.x, v.y, v.z, v.w};
    if constexpr (OP == 1) {
#pragma unroll
      for (int e = 0; e < 4; ++e) in[e] = gelu_f(in[e]);
    } else if constexpr (OP == 2) {
      const float4 hv = *reinterpret_cast<const float4*>(h + r * ldh + c);
      in[0] *= dgelu_f(hv.x); in[1] *= dgelu_f(hv.y); in[2] *= dgelu_f(hv.z); in[3] *= dgelu_f(hv.w);
    }
    bf16x4 p[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      cs[e] += in[e];
      const bf16_t p0 = (bf16_t)in[e];
      const float e1 = in[e] - (float)p0;
      const bf16_t p1 = (bf16_t)e1;
      p[0][e] = p0;
      p[1][e] = p1;
      p[2][e] = (bf16_t)(e1 - (float)p1);
    }
    constexpr int order[6] = {0, 0, 1, 0, 1, 2};
    bf16_t* o = out + r * ldo + c;
#pragma unroll
    for (int sg = 0; sg < NSEG; ++sg) *reinterpret_cast<bf16x4*>(o + sg * seg) = p[order[sg]];
  }
  if (partial) *reinterpret_cast<float4*>(partial + (long)blockIdx.y * (cols4 * 4) + c) = make_float4(cs[0], cs[1], cs[2], cs[3]);
}

// Operand rows for mv_gemm_nt_f8c (round 4; profiles/r04_fp8_correction_study.txt): x = p0 + p1 + ... with p0 = bf16(x), p1 = bf16(x - p0);
// a row of 4 * cols bytes = [p0 as bf16 | segment 1 | segment 2] with the two 8-bit segments e4m3 (OCP, v_cvt_pk_fp8_f32) of
//   role 0 (left operand):  Q(p0 * 2^e)       | Q(p1 * 2^(e + 8))          role 1 (right operand):  Q(p1 * 2^(e + 8)) | Q(p0 * 2^e)
// so that segment s of one operand meets segment s of the other and both products carry 2^(e_a + e_b + 8).  e is the caller's: the
// largest |x| times 2^e must stay below 448 (the low piece is at most 2^-9 of the high one: the same bound with 2^8 more).
template <int ROLE>
__global__ __launch_bounds__(256) void split_f8c_kernel(const float* __restrict__ x, long ldx, unsigned char* __restrict__ out,
                                                        long ldo_bytes, long rows, int cols4, float s_hi, float s_lo) {
  const long total = rows * cols4;
  for (long idx = (long)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (long)gridDim.x * 256) {
    const long r = idx / cols4;
    const int c = (int)(idx - r * cols4) * 4;
    const float4 v = *reinterpret_cast<const float4*>(x + r * ldx + c);
    const float in[4] = {v.x, v.y, v.z, v.w};
    bf16x4 p0;
    float hi[4], lo[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const bf16_t b = (bf16_t)in[e];
      p0[e] = b;
      hi[e] = (float)b * s_hi;
      lo[e] = (float)(bf16_t)(in[e] - (float)b) * s_lo;
    }
    int qh = __builtin_amdgcn_cvt_pk_fp8_f32(hi[0], hi[1], 0, false);
    qh = __builtin_amdgcn_cvt_pk_fp8_f32(hi[2], hi[3], qh, true);
    int ql = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], 0, false);
    ql = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], ql, true);
    unsigned char* o = out + r * ldo_bytes;
    const long cols = (long)cols4 * 4;
    *reinterpret_cast<bf16x4*>(o + 2L * c) = p0;
    *reinterpret_cast<int*>(o + 2 * cols + c) = ROLE == 0 ? qh : ql;
    *reinterpret_cast<int*>(o + 3 * cols + c) = ROLE == 0 ? ql : qh;
  }
}

inline int split3_parts(long rows) {
  long p = rows / 16;
  return (int)(p < 1 ? 1 : (p > 1024 ? 1024 : p));
}

extern "C" size_t mv_split3_ex_workspace_bytes(long rows, int cols) { return (size_t)split3_parts(rows) * cols * sizeof(float) + 256; }

namespace {
template <int NSEG>
int launch_split_ex(const float* x, long ldx, const float* h, long ldh, int op, void* out, long rows, int cols, float* colsum,
                    float* workspace, size_t workspace_bytes, mv_stream_t stream) {
  MV_REQUIRE(rows >= 0 && cols >= 0 && op >= 0 && op <= 2 && (op != 2 || h), MV_ERR_SHAPE);
  if (rows == 0 || cols == 0) return MV_OK;
  MV_REQUIRE(cols % 4 == 0 && ldx % 4 == 0 && ldx >= cols && (op != 2 || (ldh % 4 == 0 && ldh >= cols)), MV_ERR_ALIGN);
  MV_REQUIRE(mv_aligned16(x) && mv_aligned16(out) && mv_aligned16(h) && mv_aligned16(workspace), MV_ERR_ALIGN);
  const int parts = split3_parts(rows);
  MV_REQUIRE(!colsum || workspace_bytes >= (size_t)parts * cols * sizeof(float), MV_ERR_WORKSPACE);
  const int rpp = (int)((rows + parts - 1) / parts);
  dim3 grid(mv_cdiv(cols / 4, 256), mv_cdiv(rows, rpp));
  float* partial = colsum ? workspace : nullptr;
  const long ldo = (long)NSEG * cols, seg = cols;
  if (op == 0)
    split3_ex_kernel<0, NSEG><<<grid, 256, 0, S_>>>(x, ldx, h, ldh, (bf16_t*)out, ldo, seg, rows, cols / 4, rpp, partial);
  else if (op == 1)
    split3_ex_kernel<1, NSEG><<<grid, 256, 0, S_>>>(x, ldx, h, ldh, (bf16_t*)out, ldo, seg, rows, cols / 4, rpp, partial);
  else
    split3_ex_kernel<2, NSEG><<<grid, 256, 0, S_>>>(x, ldx, h, ldh, (bf16_t*)out, ldo, seg, rows, cols / 4, rpp, partial);
  MV_CHECK_LAUNCH();
  if (colsum) {
    mv_reduce_rows_kernel<<<mv_reduce_rows_grid(cols), 1024, 0, S_>>>(workspace, (int)grid.y, cols, (long)cols, colsum, colsum,
                                                                        colsum, cols, cols, 0);
    MV_CHECK_LAUNCH();
  }
  return MV_OK;
}

template <int NSEG>
int launch_split(const float* x, long ldx, void* out, long ldo, long seg, long rows, int cols, int role, mv_stream_t stream) {
  MV_REQUIRE(rows >= 0 && cols >= 0 && (role == 0 || role == 1), MV_ERR_SHAPE);
  if (rows == 0 || cols == 0) return MV_OK;
  MV_REQUIRE(cols % 4 == 0 && ldx % 4 == 0 && ldo % 4 == 0 && seg % 4 == 0 && ldx >= cols && ldo >= cols, MV_ERR_ALIGN);
  MV_REQUIRE(mv_aligned16(x) && mv_aligned16(out), MV_ERR_ALIGN);
  const int grid = ew_grid(rows * (cols / 4));
  if (role == 0)
    split3_kernel<0, NSEG><<<grid, 256, 0, S_>>>(x, ldx, (bf16_t*)out, ldo, seg, rows, cols / 4);
  else
    split3_kernel<1, NSEG><<<grid, 256, 0, S_>>>(x, ldx, (bf16_t*)out, ldo, seg, rows, cols / 4);
  MV_CHECK_LAUNCH();
  return MV_OK;
}
}  // namespace

extern "C" int mv_split3_bf16_ex(const float* x, long ldx, const float* h, long ldh, int op, void* out, long rows, int cols,
                                 float* colsum, float* workspace, size_t workspace_bytes, mv_stream_t stream) {
  return launch_split_ex<6>(x, ldx, h, ldh, op, out, rows, cols, colsum, workspace, workspace_bytes, stream);
}

extern "C" int mv_split3_bf16(const float* x, long ldx, void* out, long ldo, long seg, long rows, int cols, int role,
                              mv_stream_t stream) {
  return launch_split<6>(x, ldx, out, ldo, seg, rows, cols, role, stream);
}

// bf16x3: the two-piece splits ([rows, 3 cols] side by side for the _ex form; three segments `seg` apart for the plain one)
extern "C" int mv_split2_bf16_ex(const float* x, long ldx, const float* h, long ldh, int op, void* out, long rows, int cols,
                                 float* colsum, float* workspace, size_t workspace_bytes, mv_stream_t stream) {
  return launch_split_ex<3>(x, ldx, h, ldh, op, out, rows, cols, colsum, workspace, workspace_bytes, stream);
}

extern "C" int mv_split2_bf16(const float* x, long ldx, void* out, long ldo, long seg, long rows, int cols, int role,
                              mv_stream_t stream) {
  return launch_split<3>(x, ldx, out, ldo, seg, rows, cols, role, stream);
}

extern "C" int mv_weight_prep(const float* w, void* w_bf16, int ldw, void* wt_bf16, int ldt, int R, int C,
                              mv_stream_t stream) {
  MV_REQUIRE(R > 0 && C > 0 && (!w_bf16 || ldw >= C) && (!wt_bf16 || ldt >= R), MV_ERR_SHAPE);
  const int cols = (w_bf16 && ldw > C) ? ldw : C, rows = (wt_bf16 && ldt > R) ? ldt : R;
  const bool pairs = (!w_bf16 || ((ldw & 1) == 0 && (reinterpret_cast<uintptr_t>(w_bf16) & 3) == 0)) &&
                     (!wt_bf16 || ((ldt & 1) == 0 && (reinterpret_cast<uintptr_t>(wt_bf16) & 3) == 0)) &&
                     (reinterpret_cast<uintptr_t>(w) & 7) == 0;
  if (pairs) {
    dim3 grid(mv_cdiv(cols, 64), mv_cdiv(rows, 64));
    weight_prep64_kernel<<<grid, 256, 0, S_>>>(w, (bf16_t*)w_bf16, ldw, (bf16_t*)wt_bf16, ldt, R, C);
  } else {
    dim3 grid(mv_cdiv(cols, 32), mv_cdiv(rows, 32));
    weight_prep_kernel<<<grid, 256, 0, S_>>>(w, (bf16_t*)w_bf16, ldw, (bf16_t*)wt_bf16, ldt, R, C);
  }
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_weight_prep_batch(const mv_weight_prep_item* items_device, int count, int total_blocks,
                                    mv_stream_t stream) {
  MV_REQUIRE(count >= 0 && total_blocks >= 0 && (count == 0) == (total_blocks == 0), MV_ERR_SHAPE);
  if (count == 0) return MV_OK;
  MV_REQUIRE(items_device != nullptr, MV_ERR_SHAPE);
  weight_prep_batch_kernel<<<total_blocks, 256, 0, S_>>>(items_device, count);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_weight_split(const float* w, void* fwd, void* dx, int R, int C, int nseg, mv_stream_t stream) {
  MV_REQUIRE(R > 0 && C > 0 && R % 2 == 0 && C % 2 == 0 && (nseg == 3 || nseg == 6) && (fwd || dx), MV_ERR_SHAPE);
  MV_REQUIRE(mv_aligned16(w) && (!fwd || mv_aligned16(fwd)) && (!dx || mv_aligned16(dx)), MV_ERR_ALIGN);
  const dim3 grid(mv_cdiv(C, 64), mv_cdiv(R, 64));
  if (nseg == 3)
    weight_split_kernel<3><<<grid, 256, 0, S_>>>(w, (bf16_t*)fwd, (bf16_t*)dx, R, C);
  else
    weight_split_kernel<6><<<grid, 256, 0, S_>>>(w, (bf16_t*)fwd, (bf16_t*)dx, R, C);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_split_f8c(const float* x, long ldx, void* out, long ldo_bytes, long rows, int cols, int role, int exp_hi,
                            mv_stream_t stream) {
  MV_REQUIRE(rows >= 0 && cols >= 0 && (role == 0 || role == 1) && exp_hi > -100 && exp_hi < 100, MV_ERR_SHAPE);
  if (rows == 0 || cols == 0) return MV_OK;
  MV_REQUIRE(cols % 4 == 0 && ldx % 4 == 0 && ldx >= cols && ldo_bytes >= 4L * cols && ldo_bytes % 16 == 0, MV_ERR_ALIGN);
  MV_REQUIRE(mv_aligned16(x) && mv_aligned16(out), MV_ERR_ALIGN);
  const int grid = ew_grid(rows * (cols / 4));
  const float s_hi = ldexpf(1.0f, exp_hi), s_lo = ldexpf(1.0f, exp_hi + 8);
  if (role == 0)
    split_f8c_kernel<0><<<grid, 256, 0, S_>>>(x, ldx, (unsigned char*)out, ldo_bytes, rows, cols / 4, s_hi, s_lo);
  else
    split_f8c_kernel<1><<<grid, 256, 0, S_>>>(x, ldx, (unsigned char*)out, ldo_bytes, rows, cols / 4, s_hi, s_lo);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_gelu_fwd(const void* x, void* y, int dtype, long n, mv_stream_t stream) {
  MV_REQUIRE(n >= 0, MV_ERR_SHAPE);
  if (n == 0) return MV_OK;
  if (dtype == MV_F32)
    gelu_fwd_kernel<float><<<ew_grid(n), 256, 0, S_>>>((const float*)x, (float*)y, n);
  else if (dtype == MV_BF16)
    gelu_fwd_kernel<bf16_t><<<ew_grid(n), 256, 0, S_>>>((const bf16_t*)x, (bf16_t*)y, n);
  else
    return MV_ERR_UNSUPPORTED;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_gelu_bwd(const void* x, const void* dy, void* dx, int dtype, long n, mv_stream_t stream) {
  MV_REQUIRE(n >= 0, MV_ERR_SHAPE);
  if (n == 0) return MV_OK;
  if (dtype == MV_F32)
    gelu_bwd_kernel<float><<<ew_grid(n), 256, 0, S_>>>((const float*)x, (const float*)dy, (float*)dx, n);
  else if (dtype == MV_BF16)
    gelu_bwd_kernel<bf16_t><<<ew_grid(n), 256, 0, S_>>>((const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n);
  else
    return MV_ERR_UNSUPPORTED;
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_add_f32(const float* a, const float* b, float* out, long n, mv_stream_t stream) {
  MV_REQUIRE(n >= 0, MV_ERR_SHAPE);
  if (n == 0) return MV_OK;
  MV_REQUIRE(mv_aligned16(a) && mv_aligned16(b) && mv_aligned16(out), MV_ERR_ALIGN);
  add_kernel<<<ew_grid((n + 3) / 4), 256, 0, S_>>>(a, b, out, n);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_quant_float(const float* x, float* y, long n, int exp_bits, int man_bits, mv_stream_t stream) {
  MV_REQUIRE(n >= 0 && exp_bits >= 2 && exp_bits <= 8 && man_bits >= 0 && man_bits <= 22, MV_ERR_SHAPE);
  if (n == 0) return MV_OK;
  MV_REQUIRE(mv_aligned16(x) && mv_aligned16(y), MV_ERR_ALIGN);
  quant_float_kernel<<<ew_grid((n + 3) / 4), 256, 0, S_>>>(x, y, n, exp_bits, man_bits);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

// float_quantize(5, 10) written as IEEE half: every value of that format (subnormals included, saturation at 65504) is
// exactly representable, so the conversion after the quantiser's own rounding is exact
__global__ __launch_bounds__(256) void quant_float_f16_kernel(const float* __restrict__ x, _Float16* __restrict__ y, long n) {
  const long n4 = n >> 2;
  typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const float4 v = reinterpret_cast<const float4*>(x)[i];
    reinterpret_cast<f16x4_t*>(y)[i] = (f16x4_t){(_Float16)quant_float_one(v.x, 5, 10), (_Float16)quant_float_one(v.y, 5, 10),
                                                 (_Float16)quant_float_one(v.z, 5, 10), (_Float16)quant_float_one(v.w, 5, 10)};
  }
  for (long i = (n4 << 2) + (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256)
    y[i] = (_Float16)quant_float_one(x[i], 5, 10);
}

extern "C" int mv_quant_float_f16(const float* x, void* y, long n, mv_stream_t stream) {
  MV_REQUIRE(n >= 0, MV_ERR_SHAPE);
  if (n == 0) return MV_OK;
  MV_REQUIRE(mv_aligned16(x) && mv_aligned16(y), MV_ERR_ALIGN);
  quant_float_f16_kernel<<<ew_grid((n + 3) / 4), 256, 0, S_>>>(x, (_Float16*)y, n);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_quant_fixed(const float* x, float* y, long n, int wl, int fl, int clamp, int symmetric,
                              mv_stream_t stream) {
  MV_REQUIRE(n >= 0 && wl > 0 && wl <= 32, MV_ERR_SHAPE);
  if (n == 0) return MV_OK;
  const float scale = ldexpf(1.f, fl), inv = ldexpf(1.f, -fl);
  float tmax = ldexpf(1.f, wl - fl - 1) - inv, tmin = -ldexpf(1.f, wl - fl - 1);
  if (symmetric) tmin += inv;
  quant_fixed_kernel<<<ew_grid(n), 256, 0, S_>>>(x, y, n, scale, inv, tmin, tmax, clamp);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_quant_affine(const float* x, float* y, long n, float scale, int zero_point, int qmin, int qmax,
                               mv_stream_t stream) {
  MV_REQUIRE(n >= 0 && scale > 0.f && qmin < qmax, MV_ERR_SHAPE);
  if (n == 0) return MV_OK;
  quant_affine_kernel<<<ew_grid(n), 256, 0, S_>>>(x, y, n, scale, 1.0f / scale, zero_point, qmin, qmax);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

template <typename XT>
static int launch_quant_codes(const XT* x, void* codes, long rows, int cols, int ld, float inv, int zero_point, int qmin,
                              int qmax, int pre_op, hipStream_t s) {
  const bool vec = (cols & 3) == 0 && (ld & 3) == 0 && mv_aligned16(x) && (reinterpret_cast<uintptr_t>(codes) & 7) == 0;
  if (vec) {
    const int grid = rows < 65536 ? (int)rows : 65536;
    if (pre_op)
      quant_affine_codes_vec_kernel<1, XT><<<grid, 256, 0, s>>>(x, (bf16_t*)codes, rows, cols, ld, inv, zero_point, qmin, qmax);
    else
      quant_affine_codes_vec_kernel<0, XT><<<grid, 256, 0, s>>>(x, (bf16_t*)codes, rows, cols, ld, inv, zero_point, qmin, qmax);
  } else if (pre_op) {
    quant_affine_codes_kernel<1, XT><<<ew_grid(rows * ld), 256, 0, s>>>(x, (bf16_t*)codes, rows, cols, ld, inv, zero_point, qmin, qmax);
  } else {
    quant_affine_codes_kernel<0, XT><<<ew_grid(rows * ld), 256, 0, s>>>(x, (bf16_t*)codes, rows, cols, ld, inv, zero_point, qmin, qmax);
  }
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_quant_affine_codes(const void* x, int x_dtype, void* codes, long rows, int cols, int ld, float scale,
                                     int zero_point, int qmin, int qmax, int pre_op, mv_stream_t stream) {
  MV_REQUIRE(rows >= 0 && cols > 0 && ld >= cols && scale > 0.f && qmin < qmax, MV_ERR_SHAPE);
  MV_REQUIRE(qmax - zero_point <= 256 && zero_point - qmin <= 256, MV_ERR_UNSUPPORTED);     // exact in bf16
  MV_REQUIRE((pre_op == 0 || pre_op == 1) && (x_dtype == MV_F32 || x_dtype == MV_BF16), MV_ERR_UNSUPPORTED);
  if (rows == 0) return MV_OK;
  const float inv = 1.0f / scale;
  if (x_dtype == MV_F32)
    return launch_quant_codes((const float*)x, codes, rows, cols, ld, inv, zero_point, qmin, qmax, pre_op, S_);
  return launch_quant_codes((const bf16_t*)x, codes, rows, cols, ld, inv, zero_point, qmin, qmax, pre_op, S_);
}

extern "C" int mv_quant_affine_i8(const void* x, int x_dtype, void* codes, long rows, int cols, int ld, float scale,
                                  int zero_point, int pre_op, mv_stream_t stream) {
  MV_REQUIRE(rows >= 0 && cols > 0 && ld >= cols && ld % 16 == 0 && scale > 0.f, MV_ERR_SHAPE);
  MV_REQUIRE(zero_point >= 0 && zero_point <= 255, MV_ERR_UNSUPPORTED);                     // quint8
  MV_REQUIRE((pre_op == 0 || pre_op == 1) && (x_dtype == MV_F32 || x_dtype == MV_BF16), MV_ERR_UNSUPPORTED);
  MV_REQUIRE(mv_aligned16(codes), MV_ERR_ALIGN);
  if (rows == 0) return MV_OK;
  const float inv = 1.0f / scale;
  const long n = rows * (long)cols;
  if (ld == cols && (n & 1023) == 0 && mv_aligned16(x)) {         // no padding columns: the coalesced flat form
    const int fgrid = ew_grid(n >> 10, 4);
#define MV_QI8F(PRE_, T_) quant_affine_i8_flat_kernel<PRE_, T_><<<fgrid, 256, 0, S_>>>((const T_*)x, (unsigned*)codes, n, inv, zero_point)
    if (x_dtype == MV_F32) { if (pre_op) MV_QI8F(1, float); else MV_QI8F(0, float); }
    else { if (pre_op) MV_QI8F(1, bf16_t); else MV_QI8F(0, bf16_t); }
#undef MV_QI8F
    MV_CHECK_LAUNCH();
    return MV_OK;
  }
  const int grid = rows < 4096 ? (int)rows : 4096;
#define MV_QI8(PRE_, T_) quant_affine_i8_kernel<PRE_, T_><<<grid, 256, 0, S_>>>((const T_*)x, (int8_t*)codes, rows, cols, ld, inv, zero_point)
  if (x_dtype == MV_F32) { if (pre_op) MV_QI8(1, float); else MV_QI8(0, float); }
  else { if (pre_op) MV_QI8(1, bf16_t); else MV_QI8(0, bf16_t); }
#undef MV_QI8
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_minmax(const float* x, long n, float* minmax, mv_stream_t stream) {
  MV_REQUIRE(n >= 0, MV_ERR_SHAPE);
  if (n == 0) return MV_OK;
  // minmax[2..3] (caller allocates 4 floats) hold the order-preserving integer image during the reduction
  unsigned* ord = reinterpret_cast<unsigned*>(minmax + 2);
  minmax_begin_kernel<<<1, 1, 0, S_>>>(minmax, ord);
  minmax_kernel<<<ew_grid(n), 256, 0, S_>>>(x, n, ord);
  minmax_end_kernel<<<1, 1, 0, S_>>>(minmax, ord);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_cross_entropy(const float* logits, const int64_t* labels, float* loss_sum, void* dlogits, int dl_dtype,
                                int ld_dl, int64_t* argmax, long outer, int C, long inner, float grad_scale,
                                mv_stream_t stream) {
  MV_REQUIRE(outer >= 0 && C > 0 && inner >= 1, MV_ERR_SHAPE);
  MV_REQUIRE(!dlogits || dl_dtype == MV_F32 || dl_dtype == MV_BF16, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(!dlogits || inner > 1 || ld_dl >= C, MV_ERR_SHAPE);
  // zeroed by a KERNEL, not hipMemsetAsync: captured into a HIP graph (utils/graph.py) the memset node of ROCm 7.2 did not
  // stay ordered in front of the kernels behind it from the second replay on (stat[2..3] garbage, the loss off or NaN while
  // every gradient stayed right: tools/diag/graph_ce_only.py)
  mv_zero_f32_kernel<<<1, 64, 0, S_>>>(loss_sum, 4);
  const long total = outer * inner;
  if (total == 0) return MV_OK;
  ce_label_scan_kernel<<<ew_grid(total), 256, 0, S_>>>(labels, total, C, loss_sum);
  if (inner == 1) {
    const int grid = ew_grid(total, 4);
    if (dl_dtype == MV_BF16 && dlogits)
      cross_entropy_kernel<bf16_t><<<grid, 256, 0, S_>>>(logits, labels, loss_sum, (bf16_t*)dlogits, ld_dl, argmax, outer,
                                                         C, inner, grad_scale);
    else
      cross_entropy_kernel<float><<<grid, 256, 0, S_>>>(logits, labels, loss_sum, (float*)dlogits, ld_dl, argmax, outer, C,
                                                        inner, grad_scale);
  } else {
    const int grid = ew_grid(total);
    if (dl_dtype == MV_BF16 && dlogits)
      cross_entropy_pixel_kernel<bf16_t><<<grid, 256, 0, S_>>>(logits, labels, loss_sum, (bf16_t*)dlogits, argmax, outer,
                                                               C, inner, grad_scale);
    else
      cross_entropy_pixel_kernel<float><<<grid, 256, 0, S_>>>(logits, labels, loss_sum, (float*)dlogits, argmax, outer, C,
                                                              inner, grad_scale);
  }
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_upsample_bilinear_fwd(const float* small, long sb, long sc, long sp, float* big, int B, int C, int h,
                                        int w, int H, int W, mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && C > 0 && h > 0 && w > 0 && H > 0 && W > 0, MV_ERR_SHAPE);
  if (B == 0) return MV_OK;
  upsample_fwd_kernel<<<ew_grid((long)B * C * H * W), 256, 0, S_>>>(small, sb, sc, sp, big, B, C, h, w, H, W,
                                                                     (float)h / (float)H, (float)w / (float)W);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_upsample_bilinear_bwd(const float* dbig, float* dsmall, long sb, long sc, long sp, int B, int C, int h,
                                        int w, int H, int W, mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && C > 0 && h > 0 && w > 0 && H > 0 && W > 0, MV_ERR_SHAPE);
  if (B == 0) return MV_OK;
  upsample_bwd_kernel<<<ew_grid((long)B * C * h * w), 256, 0, S_>>>(dbig, dsmall, sb, sc, sp, B, C, h, w, H, W,
                                                                     (float)h / (float)H, (float)w / (float)W);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_adamw(float* p, const float* g, float* m, float* v, long n, float lr, float beta1, float beta2,
                        float eps, float weight_decay, float bias_corr1, float bias_corr2, float grad_scale,
                        const float* clip_coef, mv_stream_t stream) {
  MV_REQUIRE(n >= 0, MV_ERR_SHAPE);
  if (n == 0) return MV_OK;
  MV_REQUIRE(mv_aligned16(p) && mv_aligned16(g) && mv_aligned16(m) && mv_aligned16(v), MV_ERR_ALIGN);
  adamw_kernel<<<ew_grid((n + 3) / 4), 256, 0, S_>>>(p, g, m, v, n, lr, beta1, beta2, eps, weight_decay, bias_corr1,
                                                     bias_corr2, grad_scale, clip_coef, nullptr);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_adamw_dev(float* p, const float* g, float* m, float* v, long n, const float* hyper, float beta1, float beta2,
                            float eps, float weight_decay, float grad_scale, const float* clip_coef, mv_stream_t stream) {
  MV_REQUIRE(n >= 0 && hyper != nullptr, MV_ERR_SHAPE);
  if (n == 0) return MV_OK;
  MV_REQUIRE(mv_aligned16(p) && mv_aligned16(g) && mv_aligned16(m) && mv_aligned16(v), MV_ERR_ALIGN);
  adamw_kernel<<<ew_grid((n + 3) / 4), 256, 0, S_>>>(p, g, m, v, n, 0.f, beta1, beta2, eps, weight_decay, 1.f, 1.f, grad_scale,
                                                     clip_coef, hyper);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_sum_slabs(const float* slabs, long stride, int S, float* out, long n, int accumulate, mv_stream_t stream) {
  MV_REQUIRE(S >= 1 && n >= 0 && stride >= n && (stride & 3) == 0, MV_ERR_SHAPE);
  MV_REQUIRE(mv_aligned16(slabs) && mv_aligned16(out), MV_ERR_ALIGN);
  if (n == 0) return MV_OK;
  sum_slabs_kernel<<<ew_grid((n + 3) / 4), 256, 0, S_>>>(slabs, stride, S, accumulate ? out : nullptr, out, n);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_sum_slabs_add(const float* slabs, long stride, int S, const float* add, float* out, long n,
                                mv_stream_t stream) {
  MV_REQUIRE(S >= 1 && n >= 0 && stride >= n && (stride & 3) == 0, MV_ERR_SHAPE);
  MV_REQUIRE(mv_aligned16(slabs) && mv_aligned16(out) && mv_aligned16(add), MV_ERR_ALIGN);
  if (n == 0) return MV_OK;
  sum_slabs_kernel<<<ew_grid((n + 3) / 4), 256, 0, S_>>>(slabs, stride, S, add, out, n);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" size_t mv_grad_norm_workspace_bytes(void) { return GN_PARTS * sizeof(double); }

extern "C" int mv_grad_norm_clip(const float* g, long n, float grad_scale, float max_norm, float* out, void* workspace,
                                 size_t workspace_bytes, mv_stream_t stream) {
  MV_REQUIRE(n >= 0 && max_norm >= 0.f, MV_ERR_SHAPE);
  MV_REQUIRE(mv_aligned16(g) && mv_aligned16(workspace), MV_ERR_ALIGN);
  MV_REQUIRE(workspace_bytes >= mv_grad_norm_workspace_bytes(), MV_ERR_WORKSPACE);
  sumsq_partial_kernel<<<GN_PARTS, 256, 0, S_>>>(g, n, (double*)workspace);
  grad_norm_finish_kernel<<<1, 256, 0, S_>>>((const double*)workspace, grad_scale, max_norm, out);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_dropout(const void* x, void* y, int dtype, long n, float p, uint64_t seed, uint64_t offset,
                          mv_stream_t stream) {
  MV_REQUIRE(n >= 0 && p >= 0.f && p < 1.f, MV_ERR_SHAPE);
  MV_REQUIRE(dtype == MV_F32 || dtype == MV_BF16, MV_ERR_UNSUPPORTED);
  if (n == 0) return MV_OK;
  const unsigned thr = (unsigned)((double)p * 4294967296.0);       // keep  <=>  random u32 >= thr
  const float scale = 1.0f / (1.0f - p);
  const int grid = ew_grid((n + 3) / 4);
  if (dtype == MV_F32)
    dropout_kernel<float><<<grid, 256, 0, S_>>>((const float*)x, (float*)y, n, thr, scale, seed, offset);
  else
    dropout_kernel<bf16_t><<<grid, 256, 0, S_>>>((const bf16_t*)x, (bf16_t*)y, n, thr, scale, seed, offset);
  MV_CHECK_LAUNCH();
  return MV_OK;
}
