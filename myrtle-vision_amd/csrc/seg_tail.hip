// Fused segmentation tail:  CrossEntropyLoss()( Upsample(size=(H,W), mode="bilinear")(small), labels )
// (reference: models/vit.py:355,371 SegmentationDecoder.upsample + segmentation/train.py:188,261-265 criterion / argmax /
// accuracy) WITHOUT materialising the [B, C, H, W] logits (873 MB fp32 at B=256, C=17, 224^2) or their gradient.
//
// HBM-bound by construction.  Algorithmic bytes per output pixel: forward 8 (int64 label) + 4 (lse out) + 1 (pred out);
// backward 2 x (8 + 4) (every output row lies in the footprint of two source rows).  The small map of one image
// (h*w*C fp32 = 13 KB for 14x14x17) is staged in LDS by every workgroup and is L2-resident.
//
//   forward : one thread per output pixel (4 pixels per thread, lanes on consecutive x): interpolate the C logits from
//             LDS (same expression as upsample_fwd_kernel), max / first-index arg-max / sum-exp in channel order (same as
//             cross_entropy_pixel_kernel), write lse + pred, block-reduce loss and #correct into per-block partials;
//             a one-block finishing kernel sums the partials in a fixed order (deterministic).
//   backward: gather form, deterministic.  One workgroup per (image, source row sy); thread <-> output column x
//             accumulates  sum_Y wy(sy,Y) * (softmax_c(Y,x) - [label==c])  over the <= 2/scale footprint rows in
//             registers, parks the C sums in LDS, then thread <-> (sx, c) folds the footprint columns with wx(sx,X).
//             Writes d(small) directly in the layout the following GEMMs read (fp32 [M,C] or zero-padded bf16 [M,ld]).
#include "mv_common.h"

namespace {

constexpr int SEG_CMAX = 32;          // classes held in registers by the backward
constexpr int SEG_PX_PER_BLOCK = 1024;

// ATen upsample_bilinear2d, align_corners=False: src = (dst + 0.5) * scale - 0.5, clamped at 0 (same as elementwise.hip)
__device__ __forceinline__ void seg_src(int d, float scale, int in_size, int& i0, int& i1, float& l1) {
  float s = ((float)d + 0.5f) * scale - 0.5f;
  if (s < 0.f) s = 0.f;
  i0 = (int)s;
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + (i0 < in_size - 1 ? 1 : 0);
  l1 = s - (float)i0;
}

__device__ __forceinline__ float seg_interp(const float* s00, const float* s01, const float* s10, const float* s11, int c,
                                            float lx, float ly) {
  const float top = s00[c] * (1.f - lx) + s01[c] * lx;
  const float bot = s10[c] * (1.f - lx) + s11[c] * lx;
  return top * (1.f - ly) + bot * ly;
}

__device__ __forceinline__ void seg_stage_map(float* smap, const float* __restrict__ small, long b, int cells_c) {
  const float* src = small + b * cells_c;
  for (int i = threadIdx.x; i < cells_c; i += blockDim.x) smap[i] = src[i];
}

__global__ __launch_bounds__(256) void seg_ce_fwd_kernel(const float* __restrict__ small,
                                                         const int64_t* __restrict__ labels, float* __restrict__ lse,
                                                         uint8_t* __restrict__ pred, float* __restrict__ partials, int C,
                                                         int h, int w, int H, int W, float sh, float sw) {
  extern __shared__ float smap[];                       // [h*w][C]
  const long b = blockIdx.y;
  seg_stage_map(smap, small, b, h * w * C);
  __syncthreads();
  const int npix = H * W;
  const int p0 = blockIdx.x * SEG_PX_PER_BLOCK;
  float my_loss = 0.f, my_hit = 0.f, my_ok = 0.f, my_bad = 0.f;
  for (int p = p0 + threadIdx.x; p < p0 + SEG_PX_PER_BLOCK && p < npix; p += 256) {
    const int Y = p / W, X = p - Y * W;
    int y0, y1, x0, x1;
    float ly, lx;
    seg_src(Y, sh, h, y0, y1, ly);
    seg_src(X, sw, w, x0, x1, lx);
    const float *s00 = smap + (y0 * w + x0) * C, *s01 = smap + (y0 * w + x1) * C;
    const float *s10 = smap + (y1 * w + x0) * C, *s11 = smap + (y1 * w + x1) * C;
    // nn.CrossEntropyLoss() labels: -100 (ignore_index) is skipped and not counted; anything else outside [0, C) is an
    // error that poisons the loss (never used as an index)
    const int64_t yl = labels[b * npix + p];
    const bool valid = yl >= 0 && yl < C;
    const int y = valid ? (int)yl : -1;
    my_ok += valid ? 1.f : 0.f;
    my_bad += (!valid && yl != -100) ? 1.f : 0.f;
    float mx = -INFINITY, vy = 0.f;
    int am = 0;
    for (int c = 0; c < C; ++c) {
      const float v = seg_interp(s00, s01, s10, s11, c, lx, ly);
      if (v > mx) { mx = v; am = c; }
      if (c == y) vy = v;
    }
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += expf(seg_interp(s00, s01, s10, s11, c, lx, ly) - mx);
    const float l = mx + logf(s);
    lse[b * npix + p] = l;
    pred[b * npix + p] = (uint8_t)am;
    my_loss += valid ? l - vy : 0.f;
    my_hit += (am == y) ? 1.f : 0.f;
  }
  my_loss = wave_sum(my_loss);
  my_hit = wave_sum(my_hit);
  my_ok = wave_sum(my_ok);
  my_bad = wave_sum(my_bad);
  __shared__ float red[16];
  if ((threadIdx.x & 63) == 0) {
    red[threadIdx.x >> 6] = my_loss;
    red[4 + (threadIdx.x >> 6)] = my_hit;
    red[8 + (threadIdx.x >> 6)] = my_ok;
    red[12 + (threadIdx.x >> 6)] = my_bad;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const long blk = b * gridDim.x + blockIdx.x;
    partials[4 * blk] = (red[0] + red[1]) + (red[2] + red[3]);
    partials[4 * blk + 1] = (red[4] + red[5]) + (red[6] + red[7]);
    partials[4 * blk + 2] = (red[8] + red[9]) + (red[10] + red[11]);
    partials[4 * blk + 3] = (red[12] + red[13]) + (red[14] + red[15]);
  }
}

// stats[0] = mean loss over the counted labels (NaN if a label is out of range, or none is counted), stats[1] = pixel accuracy
// over ALL pixels (the reference's (argmax == labels).float().mean()), stats[2] = counted labels, stats[3] = bad labels;
// fixed summation order
__global__ __launch_bounds__(256) void seg_ce_finish_kernel(const float* __restrict__ partials, long nblk, float inv_pixels,
                                                            float* __restrict__ stats) {
  float v[4] = {0.f, 0.f, 0.f, 0.f};
  for (long i = threadIdx.x; i < nblk; i += 256) {
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] += partials[4 * i + k];
  }
  __shared__ float red[16];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    v[k] = wave_sum(v[k]);
    if ((threadIdx.x & 63) == 0) red[4 * k + (threadIdx.x >> 6)] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    float t[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] = (red[4 * k] + red[4 * k + 1]) + (red[4 * k + 2] + red[4 * k + 3]);
    stats[0] = (t[3] > 0.f || t[2] <= 0.f) ? __builtin_nanf("") : t[0] / t[2];
    stats[1] = t[1] * inv_pixels;
    stats[2] = t[2];
    stats[3] = t[3];
  }
}

template <typename DT>
__global__ __launch_bounds__(256) void seg_ce_bwd_kernel(const float* __restrict__ small,
                                                         const int64_t* __restrict__ labels,
                                                         const float* __restrict__ lse, DT* __restrict__ dsmall, int ld_ds,
                                                         float gscale, const float* __restrict__ stats, int C, int h, int w,
                                                         int H, int W, float sh, float sw) {
  extern __shared__ float smem[];
  float* smap = smem;                                   // [h*w][C]
  float* colacc = smem + h * w * C;                     // [W][C]
  const long b = blockIdx.y;
  const int sy = blockIdx.x;
  seg_stage_map(smap, small, b, h * w * C);
  __syncthreads();
  const int npix = H * W;
  // candidate output rows: a superset of those whose y0 or y1 can equal sy (same bound as upsample_bwd_kernel)
  int Ylo = (int)floorf(((float)sy - 1.f + 0.5f) / sh - 0.5f) - 1, Yhi = (int)ceilf(((float)sy + 1.f + 0.5f) / sh - 0.5f) + 1;
  if (Ylo < 0) Ylo = 0;
  if (Yhi > H - 1) Yhi = H - 1;
  for (int X = threadIdx.x; X < W; X += 256) {
    int x0, x1;
    float lx;
    seg_src(X, sw, w, x0, x1, lx);
    float acc[SEG_CMAX];
#pragma unroll
    for (int c = 0; c < SEG_CMAX; ++c) acc[c] = 0.f;
    for (int Y = Ylo; Y <= Yhi; ++Y) {
      int y0, y1;
      float ly;
      seg_src(Y, sh, h, y0, y1, ly);
      float wy = 0.f;
      if (y0 == sy) wy += 1.f - ly;
      if (y1 == sy) wy += ly;
      if (wy == 0.f) continue;                          // uniform over the workgroup
      const float *s00 = smap + (y0 * w + x0) * C, *s01 = smap + (y0 * w + x1) * C;
      const float *s10 = smap + (y1 * w + x0) * C, *s11 = smap + (y1 * w + x1) * C;
      const long pix = b * npix + (long)Y * W + X;
      const int64_t yl = labels[pix];
      if (yl < 0 || yl >= C) continue;                  // ignored (or bad) label: no gradient from this pixel
      const int y = (int)yl;
      const float l = lse[pix];
#pragma unroll
      for (int c = 0; c < SEG_CMAX; ++c)
        if (c < C) acc[c] += wy * (expf(seg_interp(s00, s01, s10, s11, c, lx, ly) - l) - (c == y ? 1.f : 0.f));
    }
#pragma unroll
    for (int c = 0; c < SEG_CMAX; ++c)
      if (c < C) colacc[X * C + c] = acc[c];
  }
  __syncthreads();
  // mean over the COUNTED labels (stats[2] of the forward pass); without stats: over every pixel
  const float cnt = stats ? stats[2] : (float)gridDim.y * (float)npix;
  const float gs = cnt > 0.f ? gscale / cnt : 0.f;
  for (int i = threadIdx.x; i < w * ld_ds; i += 256) {
    const int sx = i / ld_ds, c = i - sx * ld_ds;
    float s = 0.f;
    if (c < C) {
      int Xlo = (int)floorf(((float)sx - 1.f + 0.5f) / sw - 0.5f) - 1, Xhi = (int)ceilf(((float)sx + 1.f + 0.5f) / sw - 0.5f) + 1;
      if (Xlo < 0) Xlo = 0;
      if (Xhi > W - 1) Xhi = W - 1;
      for (int X = Xlo; X <= Xhi; ++X) {
        int x0, x1;
        float lx;
        seg_src(X, sw, w, x0, x1, lx);
        float wx = 0.f;
        if (x0 == sx) wx += 1.f - lx;
        if (x1 == sx) wx += lx;
        if (wx != 0.f) s += wx * colacc[X * C + c];
      }
    }
    dsmall[(b * h * w + (long)sy * w + sx) * ld_ds + c] = (DT)(s * gs);
  }
}

constexpr size_t SEG_LDS_LIMIT = 64 * 1024;

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" long mv_seg_ce_partials(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  return (long)B * (((long)H * W + SEG_PX_PER_BLOCK - 1) / SEG_PX_PER_BLOCK);
}

extern "C" int mv_seg_ce_fwd(const float* small, const int64_t* labels, float* lse, uint8_t* pred, float* partials,
                             float* stats, int B, int C, int h, int w, int H, int W, mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && C > 0 && C <= 256 && h > 0 && w > 0 && H > 0 && W > 0, MV_ERR_SHAPE);
  const size_t lds = (size_t)h * w * C * sizeof(float);
  MV_REQUIRE(lds <= SEG_LDS_LIMIT, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(B <= 65535, MV_ERR_SHAPE);
  if (B == 0) {
    mv_zero_f32_kernel<<<1, 64, 0, S_>>>(stats, 4);          // a kernel, not hipMemsetAsync: graph-capture safe (mv_cross_entropy)
    return MV_OK;
  }
  const int bpi = (int)(((long)H * W + SEG_PX_PER_BLOCK - 1) / SEG_PX_PER_BLOCK);
  seg_ce_fwd_kernel<<<dim3(bpi, B), 256, lds, S_>>>(small, labels, lse, pred, partials, C, h, w, H, W,
                                                    (float)h / (float)H, (float)w / (float)W);
  seg_ce_finish_kernel<<<1, 256, 0, S_>>>(partials, (long)bpi * B, 1.0f / ((float)B * (float)H * (float)W), stats);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_seg_ce_bwd(const float* small, const int64_t* labels, const float* lse, const float* stats, void* dsmall,
                             int ds_dtype, int ld_ds, float grad_scale, int B, int C, int h, int w, int H, int W,
                             mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && C > 0 && h > 0 && w > 0 && H > 0 && W > 0 && ld_ds >= C, MV_ERR_SHAPE);
  MV_REQUIRE(ds_dtype == MV_F32 || ds_dtype == MV_BF16, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(C <= SEG_CMAX, MV_ERR_UNSUPPORTED);
  const size_t lds = ((size_t)h * w * C + (size_t)W * C) * sizeof(float);
  MV_REQUIRE(lds <= SEG_LDS_LIMIT, MV_ERR_UNSUPPORTED);
  MV_REQUIRE(B <= 65535, MV_ERR_SHAPE);
  if (B == 0) return MV_OK;
  if (ds_dtype == MV_BF16)
    seg_ce_bwd_kernel<bf16_t><<<dim3(h, B), 256, lds, S_>>>(small, labels, lse, (bf16_t*)dsmall, ld_ds, grad_scale, stats, C, h,
                                                            w, H, W, (float)h / (float)H, (float)w / (float)W);
  else
    seg_ce_bwd_kernel<float><<<dim3(h, B), 256, lds, S_>>>(small, labels, lse, (float*)dsmall, ld_ds, grad_scale, stats, C, h, w,
                                                           H, W, (float)h / (float)H, (float)w / (float)W);
  MV_CHECK_LAUNCH();
  return MV_OK;
}
