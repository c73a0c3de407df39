// How fast can ONE compute unit take operand tiles in from L2?  (gfx950; hipcc --offload-arch=gfx950 -O3 -o probe_l2_to_cu probe_l2_to_cu.hip)
// The 256 x 256 GEMM main loops all run at the rate their operands arrive: 64 KB per K-tile of 64 through buffer_load ... lds, measured
// 20 B/clk/CU (DESIGN findings 31, 34).  This probe moves the same bytes with the same addresses (8 waves, each wave-instruction = 8 rows
// x 128 B of a [rows][768] bf16 matrix, 256 + 256 rows per workgroup, 12 K-tiles, repeated) and nothing else:
//   mode 0: buffer_load_dwordx4 ... lds (the kernels' path)        mode 1: global_load_dwordx4 into VGPRs (no LDS write at all)
//   mode 2: mode 0 for the A rows + mode 1 for the B rows (two paths at once)
//   hot = 1: every workgroup reads the same 512 rows (L2-resident)  hot = 0: workgroup b reads A rows 256 (b % 197), B rows 256 (b % 12)
// Prints bytes per shader cycle per CU (s_memtime) and the aggregate rate.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define LDS_PTR(T, p) (reinterpret_cast<__attribute__((address_space(3))) T*>(reinterpret_cast<size_t>(p)))

template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const char* __restrict__ A, const char* __restrict__ B, int lda_bytes, int hot, int reps,
                                            unsigned long long* __restrict__ cycles, unsigned* __restrict__ sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int ra = hot ? 0 : 256 * (blockIdx.x % 197), rb = hot ? 0 : 256 * (blockIdx.x % 12);
  // wave w stages rows 32 w .. 32 w + 31 of each operand: 4 pieces of 8 rows; lane -> row lane >> 3, 16-byte chunk lane & 7
  const unsigned oa = (unsigned)(ra + 32 * wave + (lane >> 3)) * (unsigned)lda_bytes + 16u * (lane & 7);
  const unsigned ob = (unsigned)(rb + 32 * wave + (lane >> 3)) * (unsigned)lda_bytes + 16u * (lane & 7);
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(A), (short)0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsB = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(B), (short)0, 0x7FFFFFFF, 0x00020000);
  char* const wl = smem + __builtin_amdgcn_readfirstlane(wave) * 8192;      // 8 pieces of 1 KiB per wave and buffer
  u32x4 acc = {0u, 0u, 0u, 0u};
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int r = 0; r < reps; ++r) {
#pragma unroll 1
    for (int kt = 0; kt < 12; ++kt) {
      const int so = kt * 128;                                              // K-tile kt: bytes 128 kt .. of every row
      char* const l = wl + (kt & 1) * 65536;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        if (MODE == 0 || MODE == 2)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, LDS_PTR(void, l + p * 1024), 16, (int)(oa + 8u * p * lda_bytes), so, 0, 0);
        else
          acc ^= *reinterpret_cast<const u32x4*>(A + oa + 8u * p * lda_bytes + so);
        if (MODE == 0)
          __builtin_amdgcn_raw_ptr_buffer_load_lds(rsB, LDS_PTR(void, l + 4096 + p * 1024), 16, (int)(ob + 8u * p * lda_bytes), so, 0, 0);
        else
          acc ^= *reinterpret_cast<const u32x4*>(B + ob + 8u * p * lda_bytes + so);
      }
      if (MODE == 0) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");       // the previous K-tile has landed; this one stays in flight
    }
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) cycles[blockIdx.x] = t1 - t0;
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345678u) sink[0] = 1;
}

int main(int argc, char** argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 40;
  const int M = 50432, K = 768, lda = K * 2;
  char *A, *B; unsigned long long* cyc; unsigned* sink;
  hipMalloc(&A, (size_t)M * lda); hipMalloc(&B, (size_t)3072 * lda); hipMalloc(&cyc, 256 * 8); hipMalloc(&sink, 4);
  hipMemset(A, 1, (size_t)M * lda); hipMemset(B, 2, (size_t)3072 * lda);
  std::vector<unsigned long long> h(256);
  for (int hot = 1; hot >= 0; --hot)
    for (int mode = 0; mode < 3; ++mode) {
      auto launch = [&] {
        if (mode == 0) { hipFuncSetAttribute((const void*)k<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); k<0><<<256, 512, 131072>>>(A, B, lda, hot, reps, cyc, sink); }
        if (mode == 1) { hipFuncSetAttribute((const void*)k<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); k<1><<<256, 512, 131072>>>(A, B, lda, hot, reps, cyc, sink); }
        if (mode == 2) { hipFuncSetAttribute((const void*)k<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072); k<2><<<256, 512, 131072>>>(A, B, lda, hot, reps, cyc, sink); }
      };
      launch(); hipDeviceSynchronize();
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
      double mean = 0; for (auto c : h) mean += (double)c; mean /= 256;
      const double bytes = (double)reps * 12 * 65536;
      printf("%s operands, %s: %6.1f B/clk/CU (s_memtime), %7.2f TB/s over 256 CUs, %.3f ms\n", hot ? "L2-hot " : "per-tile",
             mode == 0 ? "both via buffer_load..lds" : mode == 1 ? "both via global_load -> VGPR" : "A via lds-DMA + B via VGPR ", bytes / mean,
             bytes * 256 / (ms * 1e-3) / 1e12, ms);
    }
  return 0;
}
