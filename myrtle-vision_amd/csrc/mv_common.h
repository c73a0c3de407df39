// Shared device/host helpers for the myrtle_vision HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "../../include/myrtle_vision_hip.h"

#include <atomic>
// hipFuncSetAttribute acts on the CURRENT device.  The launchers raise a kernel's dynamic-LDS limit once and remember it; the memory is
// per (call site, device), so a process that drives several GPUs -- not the one-process-per-GPU launch this library is written for, but
// legal -- does not meet the default 64 KiB limit on its second device.  MV_ONCE_PER_DEVICE(expr): 0 when expr returned 0 on this device
// (evaluated on first use there), -1 otherwise.
#define MV_ONCE_PER_DEVICE(expr_)                                                                              \
  ([&]() -> int {                                                                                              \
    static std::atomic<signed char> st_[32] = {};                                                              \
    int d_ = 0;                                                                                                \
    if (hipGetDevice(&d_) != hipSuccess || d_ < 0 || d_ >= 32) return (expr_) == 0 ? 0 : -1;                   \
    signed char s_ = st_[d_].load(std::memory_order_acquire);                                                  \
    if (s_ == 0) {                                                                                             \
      s_ = (expr_) == 0 ? 1 : -1;                                                                              \
      st_[d_].store(s_, std::memory_order_release);                                                            \
    }                                                                                                          \
    return s_ == 1 ? 0 : -1;                                                                                   \
  }())

// Compute units of the CURRENT device (persistent-kernel grid sizes), remembered per device like the attributes above.
inline int mv_cu_count() {
  static std::atomic<int> n_[32] = {};
  int d = 0;
  const bool have = hipGetDevice(&d) == hipSuccess && d >= 0 && d < 32;
  if (have) {
    const int c = n_[d].load(std::memory_order_acquire);
    if (c > 0) return c;
  }
  int n = 0;
  if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, have ? d : 0) != hipSuccess || n <= 0) n = 256;
  if (have) n_[d].store(n, std::memory_order_release);
  return n;
}

// Workgroups of `kernel_` (threads_ per group, dyn_lds_ bytes of dynamic LDS) that are RESIDENT on the current device at once:
// the grid of a grid-stride kernel whose register count limits occupancy.  A larger grid runs in rounds, and the last round
// leaves most of the chip idle (LayerNorm backward: 1 024 groups against 768 resident ran 1.33 rounds, -22 %).
#define MV_RESIDENT_BLOCKS(kernel_, threads_, dyn_lds_)                                                        \
  ([&]() -> int {                                                                                              \
    static std::atomic<int> n_[32] = {};                                                                       \
    int d_ = 0;                                                                                                \
    const bool have_ = hipGetDevice(&d_) == hipSuccess && d_ >= 0 && d_ < 32;                                  \
    if (have_) {                                                                                               \
      const int c_ = n_[d_].load(std::memory_order_acquire);                                                   \
      if (c_ > 0) return c_;                                                                                   \
    }                                                                                                          \
    int per_cu_ = 0;                                                                                           \
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu_, kernel_, threads_, dyn_lds_) != hipSuccess || per_cu_ <= 0) \
      per_cu_ = 1;                                                                                             \
    const int r_ = per_cu_ * mv_cu_count();                                                                    \
    if (have_) n_[d_].store(r_, std::memory_order_release);                                                    \
    return r_;                                                                                                 \
  }())

typedef __bf16 bf16_t;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf16x4;
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f32x2;
// 16-byte staging register.  NOT HIP's uint4: that is a struct, and a select between structs goes through
// memory (scratch); a select between native vectors stays in VGPRs.
typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;

#define MV_WAVE 64
#define LDS_PTR(T, p) ((__attribute__((address_space(3))) T*)(p))

// ---- launch plumbing ------------------------------------------------------------------------
#define MV_CHECK_LAUNCH()                                                                          \
  do {                                                                                             \
    hipError_t e__ = hipGetLastError();                                                            \
    if (e__ != hipSuccess) {                                                                       \
      if (getenv("MV_DEBUG"))                                                                      \
        fprintf(stderr, "[myrtle_vision_hip] %s:%d: HIP error %d (%s)\n", __FILE__, __LINE__, (int)e__, \
                hipGetErrorString(e__));                                                           \
      return MV_ERR_LAUNCH;                                                                        \
    }                                                                                              \
  } while (0)

// Argument check.  It also CLEARS the runtime's sticky last-error: hipGetLastError() reports the last error of ANY
// earlier runtime call on this thread (e.g. the host framework probing a pointer), which MV_CHECK_LAUNCH would
// otherwise mistake for a failed launch.  Every entry point runs at least one MV_REQUIRE before it launches.
#define MV_REQUIRE(cond, code)    \
  do {                            \
    (void)hipGetLastError();      \
    if (!(cond)) return (code);   \
  } while (0)

static inline bool mv_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }
static inline int mv_cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// ---- dtype-generic element access -------------------------------------------------------------
template <typename T>
__device__ __forceinline__ float mv_ld(const T* p) { return (float)(*p); }
template <typename T>
__device__ __forceinline__ void mv_st(T* p, float v) { *p = (T)v; }

// wave64 reductions (DPP/shuffle); every lane ends with the full result
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// exact-erf GELU and its derivative (reference: nn.GELU() default, vit.py:49)
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float dgelu_f(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// erf to |error| <= 1.5e-7 (Abramowitz-Stegun 7.1.26) on v_rcp_f32 / v_exp_f32: ~13 VALU operations against libm erff's
// branchy ~40.  Also returns e = exp(-u^2) (the Gaussian factor GELU' needs).
__device__ __forceinline__ float mv_erf_fast(float u, float& e) {
  const float au = fabsf(u);
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, au, 1.0f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  e = __builtin_amdgcn_exp2f(-1.4426950408889634f * u * u);
  return copysignf(fmaf(-p * t, e, 1.0f), u);
}
__device__ __forceinline__ float mv_gelu_fast(float x) {
  float e;
  const float h = 0.5f * x;
  return fmaf(h, mv_erf_fast(x * 0.70710678118654752440f, e), h);     // explicit fma: the same bits in every kernel that inlines it
}

// Per-tensor affine quantiser (torch.fake_quantize_per_tensor_affine / MinMaxObserver qparams): the integer code q - zp of x.
// PRE = 1: GELU (erf form) applied to x first.  ONE definition for the standalone quantiser kernels and for the producers
// that fuse it into their epilogue, so fused and unfused paths agree bit for bit.  Round 3: the GELU in front of an 8-bit
// quantiser is the 1.5e-7-accurate erf above, not libm's erff -- the exact erf was ~60 k VALU cycles per 256 x 256 tile in
// fc1's int8 epilogue, more than that tile's main loop; against a quantiser step of 1/255 of the tensor's range, 1.5e-7 moves
// a code only where the exact value sits within 1.5e-7 of a rounding boundary (about one element in 10^5, by one code:
// tests/test_hip_ops.py), the size of an fp32 rounding in front of the same quantiser.
template <int PRE>
__device__ __forceinline__ float affine_code_one(float x, float inv, float zp, float qmin, float qmax) {
  if (PRE == 1) x = mv_gelu_fast(x);
  float q = rintf(x * inv) + zp;
  q = fminf(fmaxf(q, qmin), qmax);
  return q - zp;
}
// four quint8 values as int8 operands q - 128 of the int8 MFMA, packed little-endian into one dword
template <int PRE>
__device__ __forceinline__ unsigned affine_i8_pack4(float a, float b, float c, float d, float inv, float fz) {
  const float e[4] = {a, b, c, d};
  unsigned w = 0u;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int code = (int)(affine_code_one<PRE>(e[j], inv, fz, 0.f, 255.f) + fz) - 128;
    w |= ((unsigned)code & 0xFFu) << (8 * j);
  }
  return w;
}

// Direct global -> LDS copy (LDS-DMA), 16 bytes per lane: the LDS destination is a WAVE-UNIFORM base (M0) + lane * 16,
// so a wave-instruction fills 1 KiB of consecutive LDS; any swizzle goes on the per-lane SOURCE address.  Counted in
// vmcnt; nothing orders it against ds_reads except an explicit s_waitcnt vmcnt + barrier.
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The same copy issued through inline asm, INVISIBLE to the compiler's s_waitcnt bookkeeping.  Needed wherever LDS is read
// through an intrinsic without alias information (ds_read_b64_tr_b16): seeing a pending LDS-DMA, hipcc orders such a read
// after ALL of them with s_waitcnt vmcnt(0) -- which drained the 3-stage prefetch of gemm_tn_ring_kernel at every stage and
// the next item's K/V prefetch of the persistent attention kernel at every tile.  With the hidden form nothing but the
// caller's own counted s_waitcnt vmcnt(N) + barrier orders the DMA (recipe: cdna_hip_programming.md section 5.7; M0 is
// compiler-reserved, so it is saved, written and restored inside the one statement).
__device__ __forceinline__ void glds16_hidden(const void* gsrc, void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);   // LDS byte address (low 32 bits)
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(gsrc), "s"(dst)
               : "memory");
}

// n floats <- 0 (tiny accumulators in front of atomics; a kernel rather than hipMemsetAsync: see mv_cross_entropy)
static __global__ void mv_zero_f32_kernel(float* __restrict__ p, int n) {
  for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0.f;
}

// The same LDS-DMA piece as buffer_load_dwordx4 ... offen lds, hidden from the compiler like glds16_hidden: a wave-uniform buffer
// descriptor (4 SGPRs: base, extent, raw-buffer format word) + the lane's 32-bit BYTE offset + a wave-uniform byte offset in an
// SGPR.  No 64-bit per-lane address exists, so a kernel keeps one 32-bit offset per staging row instead of a pointer pair and
// advances along the contraction with scalar arithmetic; measured on the 8-phase NT kernel: +3...9 % per ViT-B shape.
typedef __attribute__((ext_vector_type(4))) int mv_srd_t;
__device__ __forceinline__ mv_srd_t mv_make_srd(const void* base, unsigned bytes) {
  const unsigned long long p = (unsigned long long)base;
  return (mv_srd_t){(int)(unsigned)p, (int)((unsigned)(p >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void bufdma16_hidden(mv_srd_t srd, unsigned voff, unsigned soff, void* lds_wave_base) {
  unsigned keep;
  const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %4 offen lds\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(srd), "s"(dst), "s"(soff)
               : "memory");
}

// out[c] (+)= sum over rows r of partial[r * ld + c]: the second stage of the deterministic two-stage reductions
// (LayerNorm dgamma/dbeta/dx column sums, bias-gradient column sums).  One 1024-thread block per 16 columns (so even
// a 768-column reduction spreads over 48+ CUs): lane l takes column l&15 and row phase l>>4 of its wave, the 64 row
// phases of the block walk the rows round-robin, then combine through LDS in a fixed order.
// Columns [0, split0) go to out0, [split0, split1) to out1, the rest to out2.
constexpr int MV_RR_COLS = 16;
static __global__ __launch_bounds__(1024) void mv_reduce_rows_kernel(const float* __restrict__ partial, int nrows,
                                                                     int ncols, long ld, float* out0, float* out1,
                                                                     float* out2, int split0, int split1, int accumulate) {
  __shared__ float red[64][MV_RR_COLS];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x * MV_RR_COLS + (lane & 15), phase = wave * 4 + (lane >> 4);
  float s = 0.f;
  if (c < ncols)
    for (int r = phase; r < nrows; r += 64) s += partial[(long)r * ld + c];
  red[phase][lane & 15] = s;
  __syncthreads();
  if (threadIdx.x < MV_RR_COLS && c < ncols) {
    float t = 0.f;
#pragma unroll 8
    for (int w = 0; w < 64; ++w) t += red[w][threadIdx.x];
    float* o = (c < split0) ? (out0 + c) : (c < split1) ? (out1 + (c - split0)) : (out2 + (c - split1));
    *o = accumulate ? (*o + t) : t;
  }
}
inline int mv_reduce_rows_grid(int ncols) { return (ncols + MV_RR_COLS - 1) / MV_RR_COLS; }
