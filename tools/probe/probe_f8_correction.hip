// Numerics probe (gfx950) for the next round's candidate arithmetic (profiles/r04_fp8_correction_study.txt):
//     a b  ~=  a0 b0  (bf16 pieces, v_mfma_f32_16x16x32_bf16)  +  Q(a0) Q(b1) + Q(a1) Q(b0)  (e4m3, v_mfma_f32_16x16x128_f8f6f4)
// with ONE power-of-two scale per operand tensor, chosen so that both correction segments share one constant, which rides in the
// instruction's e8m0 scale operand.  One wave per 16 x 16 output tile, operands read straight from global memory: this checks the
// 8-bit instruction's operand layout (any k order is fine as long as A and B use the same one), the OCP e4m3 conversion
// (v_cvt_pk_fp8_f32) and the scale semantics against a host emulation of the same roundings and against fp64 -- not speed.
//   probe_f8_correction [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

// pieces of x: p0 = bf16(x), p1 = bf16(x - p0); q0 / q1 = e4m3(p0 * s0) / e4m3(p1 * s1)
__global__ void split_kernel(const float* __restrict__ x, bf16_t* __restrict__ p0, unsigned char* __restrict__ q0,
                             unsigned char* __restrict__ q1, long n, float s0, float s1) {
  const long i = ((long)blockIdx.x * blockDim.x + threadIdx.x) * 2;
  if (i >= n) return;
  float v[2] = {x[i], x[i + 1]}, h[2], l[2];
  for (int e = 0; e < 2; ++e) {
    const bf16_t b = (bf16_t)v[e];
    p0[i + e] = b;
    h[e] = (float)b;
    l[e] = (float)(bf16_t)(v[e] - h[e]);
  }
  const int w0 = __builtin_amdgcn_cvt_pk_fp8_f32(h[0] * s0, h[1] * s0, 0, false);
  const int w1 = __builtin_amdgcn_cvt_pk_fp8_f32(l[0] * s1, l[1] * s1, 0, false);
  q0[i] = w0 & 255; q0[i + 1] = (w0 >> 8) & 255;
  q1[i] = w1 & 255; q1[i + 1] = (w1 >> 8) & 255;
}

// C[M, N] = a0 b0^T + 2^-(ea) * (Q(a0) Q(b1)^T + Q(a1) Q(b0)^T);   K % 128 == 0, M % 16 == N % 16 == 0
__global__ __launch_bounds__(64) void gemm_kernel(const bf16_t* __restrict__ a0, const unsigned char* __restrict__ qa0,
                                                  const unsigned char* __restrict__ qa1, const bf16_t* __restrict__ b0,
                                                  const unsigned char* __restrict__ qb0, const unsigned char* __restrict__ qb1,
                                                  float* __restrict__ C, int N, int K, int scale_exp) {
  const int lane = threadIdx.x, r = lane & 15, g = lane >> 4;
  const long ra = (long)(blockIdx.y * 16 + r) * K, rb = (long)(blockIdx.x * 16 + r) * K;
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
  // the 8-bit segments first: lane (r, g) holds bytes k = 32 g .. 32 g + 31 of its row for BOTH operands
  const int sc = (scale_exp & 255) * 0x01010101;           // e8m0: 2^(scale_exp - 127) on the A side, 2^0 on the B side
  for (int k = 0; k < K; k += 128) {
    const i32x8 x0 = *reinterpret_cast<const i32x8*>(qa0 + ra + k + 32 * g), y1 = *reinterpret_cast<const i32x8*>(qb1 + rb + k + 32 * g);
    const i32x8 x1 = *reinterpret_cast<const i32x8*>(qa1 + ra + k + 32 * g), y0 = *reinterpret_cast<const i32x8*>(qb0 + rb + k + 32 * g);
    // (the accumulator comes out TRANSPOSED when the operands are swapped: B first, as the product kernels issue it)
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(y1, x0, acc, 0, 0, 0, 0x7F7F7F7F, 0, sc);
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(y0, x1, acc, 0, 0, 0, 0x7F7F7F7F, 0, sc);
  }
  for (int k = 0; k < K; k += 32) {
    const bf16x8 x = *reinterpret_cast<const bf16x8*>(a0 + ra + k + 8 * g), y = *reinterpret_cast<const bf16x8*>(b0 + rb + k + 8 * g);
    acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(y, x, acc, 0, 0, 0);
  }
  // lane (r, g) holds row m = r, columns n = 4 g .. 4 g + 3 of the tile
  float* o = C + (long)(blockIdx.y * 16 + r) * N + blockIdx.x * 16 + 4 * g;
  o[0] = acc[0]; o[1] = acc[1]; o[2] = acc[2]; o[3] = acc[3];
}

static float bf16r(float v) { return (float)(bf16_t)v; }
// host e4m3 (OCP, saturating to 448, round to nearest even) of an already scaled value, returned as float
static float e4m3r(float v) {
  if (v == 0.f || std::isnan(v)) return v;
  const float a = std::fabs(v);
  if (a >= 448.f) return std::copysign(448.f, v);
  int e;
  std::frexp(a, &e);                                     // a = f * 2^e, f in [0.5, 1)
  int q = e - 1;                                         // exponent of the leading bit
  if (q < -6) q = -6;                                    // subnormal spacing 2^-9
  const float step = std::ldexp(1.f, q - 3);
  return std::copysign(std::nearbyint(a / step) * step, v);
}

int main(int argc, char** argv) {
  const int M = argc > 3 ? atoi(argv[1]) : 256, N = argc > 3 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 768;
  std::vector<float> A((size_t)M * K), B((size_t)N * K);
  srand(7);
  auto rnd = [] { float s = 0; for (int i = 0; i < 6; ++i) s += rand() / (float)RAND_MAX - 0.5f; return s; };   // ~gaussian
  for (auto& v : A) v = rnd() * 2.5f;
  for (auto& v : B) v = rnd() * 0.08f;
  auto amax = [](const std::vector<float>& t) { float m = 0; for (float v : t) m = std::fmax(m, std::fabs(v)); return m; };
  // scales: the high pieces' largest magnitude lands in [128, 256); the low pieces are 2^-8 of the high ones, so they get 2^8 more
  const int ea = 7 - (int)std::floor(std::log2(amax(A))), eb = 7 - (int)std::floor(std::log2(amax(B)));
  const float sa0 = std::ldexp(1.f, ea), sa1 = std::ldexp(1.f, ea + 8), sb0 = std::ldexp(1.f, eb), sb1 = std::ldexp(1.f, eb + 8);
  const int total = ea + eb + 8;                          // both segments carry 2^total: divide it out through the scale operand
  float *dA, *dB, *dC; bf16_t *a0, *b0; unsigned char *qa0, *qa1, *qb0, *qb1;
  hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, (size_t)M * N * 4);
  hipMalloc(&a0, A.size() * 2); hipMalloc(&b0, B.size() * 2);
  hipMalloc(&qa0, A.size()); hipMalloc(&qa1, A.size()); hipMalloc(&qb0, B.size()); hipMalloc(&qb1, B.size());
  hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
  split_kernel<<<(unsigned)((A.size() / 2 + 255) / 256), 256>>>(dA, a0, qa0, qa1, (long)A.size(), sa0, sa1);
  split_kernel<<<(unsigned)((B.size() / 2 + 255) / 256), 256>>>(dB, b0, qb0, qb1, (long)B.size(), sb0, sb1);
  gemm_kernel<<<dim3(N / 16, M / 16), 64>>>(a0, qa0, qa1, b0, qb0, qb1, dC, N, K, 127 - total);
  std::vector<float> C((size_t)M * N);
  if (hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost) != hipSuccess) { printf("HIP error\n"); return 1; }
  double e_exact = 0, e_emu = 0, e_x3 = 0, scale = 0;
  for (int m = 0; m < M; m += 3)
    for (int n = 0; n < N; n += 5) {
      double exact = 0, emu = 0, x3 = 0;
      for (int k = 0; k < K; ++k) {
        const float a = A[(size_t)m * K + k], b = B[(size_t)n * K + k];
        const float ah = bf16r(a), al = bf16r(a - ah), bh = bf16r(b), bl = bf16r(b - bh);
        exact += (double)a * b;
        x3 += (double)ah * bh + (double)ah * bl + (double)al * bh;
        emu += (double)ah * bh + ((double)e4m3r(ah * sa0) * e4m3r(bl * sb1) + (double)e4m3r(al * sa1) * e4m3r(bh * sb0)) * std::ldexp(1.0, -total);
      }
      const double got = C[(size_t)m * N + n];
      e_exact = std::fmax(e_exact, std::fabs(got - exact)); e_emu = std::fmax(e_emu, std::fabs(got - emu));
      e_x3 = std::fmax(e_x3, std::fabs(x3 - exact)); scale = std::fmax(scale, std::fabs(exact));
    }
  printf("M %d N %d K %d  scales 2^%d (A) 2^%d (B): max |C| %.3f\n", M, N, K, ea, eb, scale);
  printf("  device vs host emulation of the same roundings : %.3e of max |C|   (fp32 summation order only)\n", e_emu / scale);
  printf("  device vs fp64 exact product                   : %.3e of max |C|\n", e_exact / scale);
  printf("  three bf16 products (bf16x3, host) vs fp64      : %.3e of max |C|\n", e_x3 / scale);
  return e_emu / scale < 1e-5 ? 0 : 2;
}
