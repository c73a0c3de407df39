// Power/clock probe (gfx950): sustained rate of register-resident MFMA streams, optionally mixed with LDS reads, so that
// tools/clock_watch.sh can record which instruction mix the power governor lets run at which clock.
//   probe_mfma_power MODE [seconds] [waves_per_simd]
//   MODE 0: v_mfma_f32_16x16x32_bf16     1: v_mfma_f32_32x32x16_bf16
//        2: mode 0 + 0.75 ds_read_b64 per MFMA (the 8-phase kernel's ratio)     3: mode 0 on all-zero operands
//        4: mode 0 + 0.375 ds_read_b128 per MFMA (same bytes, half the instructions)
//        5: v_mfma_f32_16x16x32_fp8_fp8 (8-byte operands)   6: v_mfma_f32_16x16x128_f8f6f4 on e4m3 (32-byte operands, unit scales)
//        7: v_mfma_i32_16x16x64_i8 (16-byte operands)       -- round 4: what the 8-bit paths sustain under the power cap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

typedef int i32x4_t __attribute__((ext_vector_type(4)));
typedef int i32x8_t __attribute__((ext_vector_type(8)));
template <int MODE>
__global__ void __launch_bounds__(512) k8(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  const int t = threadIdx.x;
  i32x8_t a[4], b[4];
  for (int i = 0; i < 4; ++i) {
    const i32x4_t lo = __builtin_bit_cast(i32x4_t, src[(t * 4 + i) & 4095]), hi = __builtin_bit_cast(i32x4_t, src[(t * 4 + i + 911) & 4095]);
    const i32x4_t l2 = __builtin_bit_cast(i32x4_t, src[(t * 4 + i + 1777) & 4095]), h2 = __builtin_bit_cast(i32x4_t, src[(t * 4 + i + 2777) & 4095]);
    // keep every byte a finite e4m3 / small int8 value: clear the top exponent bit of each byte
    for (int e = 0; e < 4; ++e) {
      a[i][e] = lo[e] & 0xBFBFBFBF; a[i][4 + e] = hi[e] & 0xBFBFBFBF;
      b[i][e] = l2[e] & 0xBFBFBFBF; b[i][4 + e] = h2[e] & 0xBFBFBFBF;
    }
  }
  if (MODE == 7) {
    i32x4_t acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = i32x4_t{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const i32x4_t av = {a[j & 3][0], a[j & 3][1], a[j & 3][2], a[j & 3][3]}, bv = {b[j >> 2][0], b[j >> 2][1], b[j >> 2][2], b[j >> 2][3]};
        acc[j] = __builtin_amdgcn_mfma_i32_16x16x64_i8(av, bv, acc[j], 0, 0, 0);
      }
    }
    i32x4_t s = acc[0];
    for (int j = 1; j < 16; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + t] = (float)(s[0] + s[1] + s[2] + s[3]);
  } else {
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        if (MODE == 5) {
          const long av = ((long)a[j & 3][1] << 32) | (unsigned)a[j & 3][0], bv = ((long)b[j >> 2][1] << 32) | (unsigned)b[j >> 2][0];
          acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_fp8_fp8(av, bv, acc[j], 0, 0, 0);
        } else {
          acc[j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a[j & 3], b[j >> 2], acc[j], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
      }
    }
    f32x4 s = acc[0];
    for (int j = 1; j < 16; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + t] = s[0] + s[1] + s[2] + s[3];
  }
}

template <int MODE>
__global__ void __launch_bounds__(512) k(const bf16x8* __restrict__ src, float* __restrict__ out, int iters) {
  __shared__ unsigned lds[16384];
  const int t = threadIdx.x;
  for (int i = t; i < 16384; i += blockDim.x) lds[i] = ((const unsigned*)src)[i];
  __syncthreads();
  bf16x8 a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = src[(t * 4 + i) & 4095]; b[i] = src[(t * 4 + i + 1777) & 4095]; }
  if (MODE == 1) {
    f32x16 acc[4];
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(j + r) & 3], b[j], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < 4; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * blockDim.x + t] = s;
  } else {
    f32x4 acc[16];
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const unsigned la = (t & 63) * 8 + (t >> 6) * 4096;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[j & 3], b[j >> 2], acc[j], 0, 0, 0);
        if (MODE == 2 && (j & 3) != 3) {
          asm volatile("ds_read_b64 v[200:201], %0" :: "v"(la + j * 512) : "v200", "v201");
        }
        if (MODE == 4 && (j & 7) < 3) {
          asm volatile("ds_read_b128 v[200:203], %0" :: "v"(la * 2 + j * 1024) : "v200", "v201", "v202", "v203");
        }
      }
      if (MODE == 2 || MODE == 4) asm volatile("s_waitcnt lgkmcnt(0)" ::: "v200", "v201", "v202", "v203");
    }
    f32x4 s = acc[0];
    for (int j = 1; j < 16; ++j) s += acc[j];
    out[blockIdx.x * blockDim.x + t] = s[0] + s[1] + s[2] + s[3];
  }
}

int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const double secs = argc > 2 ? atof(argv[2]) : 5.0;
  const int wps = argc > 3 ? atoi(argv[3]) : 2;                       // waves per SIMD
  const int block = 256 * wps, grid = 256, iters = 20000;
  std::vector<unsigned short> h(16384 * 2);
  srand(1);
  for (auto& v : h) { const float f = (rand() / (float)RAND_MAX - 0.5f); unsigned u; memcpy(&u, &f, 4); v = mode == 3 ? 0 : (unsigned short)(u >> 16); }
  bf16x8* src; float* out;
  if (hipMalloc(&src, h.size() * 2) != hipSuccess || hipMalloc(&out, grid * block * 4) != hipSuccess) return 1;
  hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
  auto launch = [&] {
    switch (mode) {
      case 1: k<1><<<grid, block>>>(src, out, iters); break;
      case 2: k<2><<<grid, block>>>(src, out, iters); break;
      case 4: k<4><<<grid, block>>>(src, out, iters); break;
      case 5: k8<5><<<grid, block>>>(src, out, iters); break;
      case 6: k8<6><<<grid, block>>>(src, out, iters); break;
      case 7: k8<7><<<grid, block>>>(src, out, iters); break;
      default: k<0><<<grid, block>>>(src, out, iters);
    }
  };
  launch(); hipDeviceSynchronize();
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const auto t0 = std::chrono::steady_clock::now();
  double ms_total = 0; long n = 0;
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < secs) {
    hipEventRecord(e0);
    for (int i = 0; i < 4; ++i) launch();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms_total += ms; n += 4;
  }
  // per launch: grid * waves * iters * 16 MFMAs(16x16x32: 16*16*32*2 flop)  |  8 MFMAs(32x32x16: 32*32*16*2 flop)
  // (16x16x32: 16*16*32*2 = 16384 flop; 16x16x64 int8: 32768 op; 16x16x128 8-bit: 65536 flop)
  const double per = mode == 1 ? 8.0 * 32768 : mode == 6 ? 16.0 * 65536 : mode == 7 ? 16.0 * 32768 : 16.0 * 16384;
  const double flop = (double)grid * (block / 64) * iters * per;
  printf("mode %d, %d waves/SIMD: %.1f TFLOP/s  (%.3f ms/launch)\n", mode, wps, flop * n / (ms_total * 1e-3) / 1e12, ms_total / n);
  return 0;
}
