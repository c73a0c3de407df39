// Image batch preparation on the GPU: what the reference's DataLoader worker does per image
// with torchvision + Pillow (datasets/resisc45.py:40-69, datasets/dlrsd.py:39-66, transforms/segmentation.py)
//
//   Normalize(ToTensor(hflip?(resize(crop(img, box), size, BILINEAR))))          and NEAREST for segmentation masks
//
// on a batch of decoded uint8 frames that crossed PCIe as bytes (3x fewer than the fp32 tensors the reference ships).
// Bit-exact to Pillow's 8-bit resampler (libImaging/Resample.c): separable triangle filter whose support grows with the
// downscale factor, 22-bit fixed-point coefficients, horizontal pass first with its result rounded and clipped to
// uint8, then the vertical pass.  The coefficient tables are built by the host in double precision exactly as
// precompute_coeffs does (myrtle_vision/datasets/device_transforms.py); crop offset, CenterCrop window and flip are
// folded into the tables / the output index, so ONE kernel does the whole chain.
//
// HBM-bound by construction: algorithmic bytes per output pixel = 12 (fp32 x 3 out) + ~3.3 (source bytes, each read
// once from HBM, the <= ks^2-fold re-reads hit L2/L1).
#include "mv_common.h"

namespace {

constexpr int IMG_PRECISION_BITS = 22;     // Resample.c: 32 - 8 - 2

__device__ __forceinline__ int clip8(int v) {
  v >>= IMG_PRECISION_BITS;                // arithmetic shift, as the C code's clip8 lookup index
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// U8 = true: the resampled frame itself, uint8 HWC, no flip / ToTensor / Normalize -- stage one of a
// Resize -> RandomResizedCrop chain (each Pillow resize rounds to uint8, so the two resamplings cannot be merged)
template <bool U8>
__global__ __launch_bounds__(256) void image_prepare_kernel(const uint8_t* __restrict__ src, long img_stride, int Hs, int Ws,
                                                            const int* __restrict__ kh, const int* __restrict__ bh,
                                                            const int* __restrict__ kv, const int* __restrict__ bv, int ks,
                                                            const uint8_t* __restrict__ flip, float m0, float m1, float m2,
                                                            float s0, float s1, float s2, float* __restrict__ out,
                                                            uint8_t* __restrict__ out_u8, int oh, int ow) {
  const int b = blockIdx.z;
  const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (X >= ow || Y >= oh) return;
  const int* khx = kh + ((long)b * ow + X) * ks;
  const int* kvy = kv + ((long)b * oh + Y) * ks;
  const int x0 = bh[((long)b * ow + X) * 2], nx = bh[((long)b * ow + X) * 2 + 1];
  const int y0 = bv[((long)b * oh + Y) * 2], ny = bv[((long)b * oh + Y) * 2 + 1];
  const uint8_t* img = src + (long)b * img_stride;
  const int half = 1 << (IMG_PRECISION_BITS - 1);
  int a0 = half, a1 = half, a2 = half;
  for (int y = 0; y < ny; ++y) {
    const uint8_t* row = img + ((long)(y0 + y) * Ws + x0) * 3;
    int h0 = half, h1 = half, h2 = half;                         // horizontal pass of source row y0 + y at column X
    for (int x = 0; x < nx; ++x) {
      const int k = khx[x];
      h0 += (int)row[3 * x] * k;
      h1 += (int)row[3 * x + 1] * k;
      h2 += (int)row[3 * x + 2] * k;
    }
    const int k = kvy[y];                                        // ... rounded to uint8, then the vertical tap
    a0 += clip8(h0) * k;
    a1 += clip8(h1) * k;
    a2 += clip8(h2) * k;
  }
  if constexpr (U8) {
    uint8_t* o = out_u8 + (((long)b * oh + Y) * ow + X) * 3;
    o[0] = (uint8_t)clip8(a0);
    o[1] = (uint8_t)clip8(a1);
    o[2] = (uint8_t)clip8(a2);
    return;
  }
  const int Xo = flip[b] ? ow - 1 - X : X;                       // RandomHorizontalFlip acts on the resized image
  const long plane = (long)oh * ow;
  float* o = out + (long)b * 3 * plane + (long)Y * ow + Xo;
  // ToTensor: uint8 / 255 in fp32; Normalize: (x - mean) / std in fp32 -- correctly rounded, never contracted
  o[0] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(a0), 255.0f), m0), s0);
  o[plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(a1), 255.0f), m1), s1);
  o[2 * plane] = __fdiv_rn(__fsub_rn(__fdiv_rn((float)clip8(a2), 255.0f), m2), s2);
}

// masks: Image.resize(NEAREST) = a gather through per-axis index tables (Geometry.c ImagingScaleAffine), + label offset
template <bool U8>
__global__ __launch_bounds__(256) void mask_prepare_kernel(const uint8_t* __restrict__ src, long img_stride, int Ws,
                                                           const int* __restrict__ yi, const int* __restrict__ xi,
                                                           const uint8_t* __restrict__ flip, int add,
                                                           int64_t* __restrict__ out, uint8_t* __restrict__ out_u8, int oh,
                                                           int ow) {
  const int b = blockIdx.z;
  const int X = blockIdx.x * 64 + (threadIdx.x & 63), Y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (X >= ow || Y >= oh) return;
  const int sy = yi[(long)b * oh + Y], sx = xi[(long)b * ow + X];
  if constexpr (U8) {
    out_u8[((long)b * oh + Y) * ow + X] = src[(long)b * img_stride + (long)sy * Ws + sx];
    return;
  }
  const int Xo = flip[b] ? ow - 1 - X : X;
  out[((long)b * oh + Y) * ow + Xo] = (int64_t)src[(long)b * img_stride + (long)sy * Ws + sx] + add;
}

}  // namespace

#define S_ ((hipStream_t)stream)

extern "C" int mv_image_prepare(const uint8_t* src, long img_stride, int Hs, int Ws, const int32_t* kh, const int32_t* bh,
                                const int32_t* kv, const int32_t* bv, int ks, const uint8_t* flip, float mean0, float mean1,
                                float mean2, float std0, float std1, float std2, float* out, int B, int out_h, int out_w,
                                mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && Hs > 0 && Ws > 0 && out_h > 0 && out_w > 0 && ks > 0 && ks <= 64, MV_ERR_SHAPE);
  MV_REQUIRE(img_stride >= (long)Hs * Ws * 3 && B <= 65535, MV_ERR_SHAPE);
  if (B == 0) return MV_OK;
  image_prepare_kernel<false><<<dim3(mv_cdiv(out_w, 64), mv_cdiv(out_h, 4), B), 256, 0, S_>>>(
      src, img_stride, Hs, Ws, kh, bh, kv, bv, ks, flip, mean0, mean1, mean2, std0, std1, std2, out, nullptr, out_h, out_w);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_mask_prepare(const uint8_t* src, long img_stride, int Hs, int Ws, const int32_t* yi, const int32_t* xi,
                               const uint8_t* flip, int add, int64_t* out, int B, int out_h, int out_w, mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && Hs > 0 && Ws > 0 && out_h > 0 && out_w > 0, MV_ERR_SHAPE);
  MV_REQUIRE(img_stride >= (long)Hs * Ws && B <= 65535, MV_ERR_SHAPE);
  if (B == 0) return MV_OK;
  mask_prepare_kernel<false><<<dim3(mv_cdiv(out_w, 64), mv_cdiv(out_h, 4), B), 256, 0, S_>>>(src, img_stride, Ws, yi, xi, flip, add,
                                                                                            out, nullptr, out_h, out_w);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_image_resize_u8(const uint8_t* src, long img_stride, int Hs, int Ws, const int32_t* kh, const int32_t* bh,
                                  const int32_t* kv, const int32_t* bv, int ks, uint8_t* out, int B, int out_h, int out_w,
                                  mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && Hs > 0 && Ws > 0 && out_h > 0 && out_w > 0 && ks > 0 && ks <= 64, MV_ERR_SHAPE);
  MV_REQUIRE(img_stride >= (long)Hs * Ws * 3 && B <= 65535, MV_ERR_SHAPE);
  if (B == 0) return MV_OK;
  image_prepare_kernel<true><<<dim3(mv_cdiv(out_w, 64), mv_cdiv(out_h, 4), B), 256, 0, S_>>>(
      src, img_stride, Hs, Ws, kh, bh, kv, bv, ks, nullptr, 0.f, 0.f, 0.f, 1.f, 1.f, 1.f, nullptr, out, out_h, out_w);
  MV_CHECK_LAUNCH();
  return MV_OK;
}

extern "C" int mv_mask_resize_u8(const uint8_t* src, long img_stride, int Hs, int Ws, const int32_t* yi, const int32_t* xi,
                                 uint8_t* out, int B, int out_h, int out_w, mv_stream_t stream) {
  MV_REQUIRE(B >= 0 && Hs > 0 && Ws > 0 && out_h > 0 && out_w > 0, MV_ERR_SHAPE);
  MV_REQUIRE(img_stride >= (long)Hs * Ws && B <= 65535, MV_ERR_SHAPE);
  if (B == 0) return MV_OK;
  mask_prepare_kernel<true><<<dim3(mv_cdiv(out_w, 64), mv_cdiv(out_h, 4), B), 256, 0, S_>>>(src, img_stride, Ws, yi, xi, nullptr, 0,
                                                                                           nullptr, out, out_h, out_w);
  MV_CHECK_LAUNCH();
  return MV_OK;
}
