// Hardware check of v_permlane16_swap_b32 (gfx950) as used by pair_swap_bf16 in csrc/gemm_bf16.hip:
// r = permlane16_swap(a, b): which lanes of r[0] / r[1] come from a / b of which lane.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o) {
  const unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
  const u32x2 r = __builtin_amdgcn_permlane16_swap(a, b, false, false);
  o[threadIdx.x] = r[0];
  o[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned *d, h[128];
  if (hipMalloc(&d, sizeof(h)) != hipSuccess) return 1;
  k<<<1, 64>>>(d);
  if (hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost) != hipSuccess) return 1;
  for (int l : {0, 5, 16, 21, 32, 37, 48, 53}) printf("lane %2d: r0 = %4u  r1 = %4u\n", l, h[l], h[64 + l]);
  // expectation: lanes with even lane>>4: r0 = a(own), r1 = a(lane+16); odd: r0 = b(lane-16), r1 = b(own)
  int ok = 1;
  for (int l = 0; l < 64; ++l) {
    const int g = l >> 4;
    const unsigned e0 = (g & 1) ? 1000 + (l - 16) : l, e1 = (g & 1) ? 1000 + l : l + 16;
    ok &= (h[l] == e0) && (h[64 + l] == e1);
  }
  printf("permlane16_swap matches the assumed semantics: %s\n", ok ? "YES" : "NO");
  return ok ? 0 : 2;
}
