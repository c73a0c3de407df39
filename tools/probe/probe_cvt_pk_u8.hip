// What does v_cvt_pk_u8_f32 do with halves, negatives, values past 255 and NaN?  (hipcc --offload-arch=gfx950 -o probe probe_cvt_pk_u8.hip)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float* x, unsigned* y, int n) {
  const int i = threadIdx.x;
  if (i < n) y[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 1u, 0xAABBCCDDu);
}
int main() {
  const float h[] = {0.f, 0.4f, 0.5f, 0.6f, 1.5f, 2.5f, 3.5f, 25.5f, 26.49f, 26.5f, 226.5f, 254.5f, 255.f, 255.4f, 255.5f, 300.f, -0.4f, -0.6f, -5.f, NAN, INFINITY, 127.5f, 128.5f};
  const int n = sizeof(h) / sizeof(float);
  float* dx; unsigned* dy; unsigned out[64];
  hipMalloc(&dx, sizeof(h)); hipMalloc(&dy, 64 * 4);
  hipMemcpy(dx, h, sizeof(h), hipMemcpyHostToDevice);
  k<<<1, 64>>>(dx, dy, n);
  hipMemcpy(out, dy, n * 4, hipMemcpyDeviceToHost);
  for (int i = 0; i < n; ++i) printf("%10.3f -> byte1 = %3u   (word %08x)\n", h[i], (out[i] >> 8) & 255u, out[i]);
  return 0;
}
