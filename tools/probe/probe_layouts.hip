// Hardware-layout probe (diagnostic, not product): checks the gfx950 MFMA 16x16x32 bf16
// operand/accumulator lane maps and the ds_read_b64_tr_b16 gather against the maps this
// repo's kernels assume.  Build: hipcc --offload-arch=gfx950 -O2 probe_layouts.hip -o probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) __bf16 bf4;
typedef __attribute__((ext_vector_type(4))) float f4;
#define LDSP(T, p) ((__attribute__((address_space(3))) T*)(p))

static inline unsigned short f2bf(float f) { unsigned u; memcpy(&u, &f, 4); return (unsigned short)(u >> 16); }

// C[16x16] = A[16x32] * B[32x16]; A row-major [16][32], Bt row-major [16 cols][32 k]
__global__ void k_mfma(const unsigned short* A, const unsigned short* Bt, float* C) {
  int l = threadIdx.x;
  bf8 a = *(const bf8*)(A + (l & 15) * 32 + 8 * (l >> 4));
  bf8 b = *(const bf8*)(Bt + (l & 15) * 32 + 8 * (l >> 4));
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}

// tile[32 rows][16 cols] of bf16 in LDS, row-major (32-byte rows).  Group g reads the 4x16 block
// at rows 8g..8g+3 (first) and 8g+4..8g+7 (second).  Output: what each lane received.
__global__ void k_tr(const unsigned short* T, float* out) {
  __shared__ __attribute__((aligned(16))) unsigned short lds[32 * 16];
  int l = threadIdx.x;
  for (int i = l; i < 512; i += 64) lds[i] = T[i];
  __syncthreads();
  int g = l >> 4, i = l & 15, q = i >> 2, p = i & 3;
  bf4 v0 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDSP(bf4, &lds[(8 * g + q) * 16 + 4 * p]));
  bf4 v1 = __builtin_amdgcn_ds_read_tr16_b64_v4bf16(LDSP(bf4, &lds[(8 * g + 4 + q) * 16 + 4 * p]));
  for (int e = 0; e < 4; ++e) { out[l * 8 + e] = (float)v0[e]; out[l * 8 + 4 + e] = (float)v1[e]; }
}

// f32 MFMA 16x16x4: A[16][4], B[4][16]
__global__ void k_mfma_f32(const float* A, const float* B, float* C) {
  int l = threadIdx.x;
  float a = A[(l & 15) * 4 + (l >> 4)];
  float b = B[(l >> 4) * 16 + (l & 15)];
  f4 acc = {0, 0, 0, 0};
  acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
  for (int r = 0; r < 4; ++r) C[((l >> 4) * 4 + r) * 16 + (l & 15)] = acc[r];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 2; } } while (0)

int main() {
  int bad = 0;
  {  // MFMA bf16 map
    unsigned short hA[16 * 32], hB[16 * 32]; float fA[16 * 32], fB[16 * 32], hC[256];
    srand(1);
    for (int i = 0; i < 512; ++i) { fA[i] = (float)(rand() % 7 - 3); fB[i] = (float)(rand() % 5 - 2); hA[i] = f2bf(fA[i]); hB[i] = f2bf(fB[i]); }
    unsigned short *dA, *dB; float* dC;
    CK(hipMalloc(&dA, sizeof hA)); CK(hipMalloc(&dB, sizeof hB)); CK(hipMalloc(&dC, sizeof hC));
    CK(hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice));
    k_mfma<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost));
    int err = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 32; ++k) s += fA[m * 32 + k] * fB[n * 32 + k]; if (s != hC[m * 16 + n]) ++err; }
    printf("mfma_16x16x32_bf16 map: %s (%d mismatches)\n", err ? "FAIL" : "ok", err); bad += err;
  }
  {  // tr read
    unsigned short hT[512]; float hO[512];
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 16; ++c) hT[r * 16 + c] = f2bf((float)(r * 16 + c) * 0.25f);  // exact in bf16? r*16+c<512 -> 9 bits: not all exact
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 16; ++c) hT[r * 16 + c] = f2bf((float)(r * 4) + (float)c * 0.0625f * 0 + (float)(c));  // r*4+c collides; use pair encoding below
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 16; ++c) hT[r * 16 + c] = f2bf((float)(r * 16 + c) <= 255 ? (float)(r * 16 + c) : (float)(r * 16 + c - 256) + 0.5f * 0 - 256.0f * 0 - (float)0);
    // values 0..255 exact in bf16; for rows 16..31 store negative (-(idx-256)-1) -> still exact
    for (int r = 0; r < 32; ++r) for (int c = 0; c < 16; ++c) { int idx = r * 16 + c; float v = idx < 256 ? (float)idx : -(float)(idx - 256) - 1.0f; hT[idx] = f2bf(v); }
    unsigned short* dT; float* dO;
    CK(hipMalloc(&dT, sizeof hT)); CK(hipMalloc(&dO, sizeof hO));
    CK(hipMemcpy(dT, hT, sizeof hT, hipMemcpyHostToDevice));
    k_tr<<<1, 64>>>(dT, dO); CK(hipDeviceSynchronize());
    CK(hipMemcpy(hO, dO, sizeof hO, hipMemcpyDeviceToHost));
    int err = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 8; ++j) {
      int g = l >> 4, i = l & 15; int row = 8 * g + j, col = i; int idx = row * 16 + col;
      float want = idx < 256 ? (float)idx : -(float)(idx - 256) - 1.0f;
      if (hO[l * 8 + j] != want) { if (err < 8) printf("  tr lane %d elem %d got %g want %g\n", l, j, hO[l * 8 + j], want); ++err; }
    }
    printf("ds_read_b64_tr_b16 gather (lane i <- column i, element q <- row q): %s (%d mismatches)\n", err ? "FAIL" : "ok", err); bad += err;
  }
  {  // f32 MFMA
    float hA[64], hB[64], hC[256]; srand(2);
    for (int i = 0; i < 64; ++i) { hA[i] = (float)(rand() % 9 - 4); hB[i] = (float)(rand() % 7 - 3); }
    float *dA, *dB, *dC;
    CK(hipMalloc(&dA, sizeof hA)); CK(hipMalloc(&dB, sizeof hB)); CK(hipMalloc(&dC, sizeof hC));
    CK(hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice)); CK(hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice));
    k_mfma_f32<<<1, 64>>>(dA, dB, dC); CK(hipDeviceSynchronize());
    CK(hipMemcpy(hC, dC, sizeof hC, hipMemcpyDeviceToHost));
    int err = 0;
    for (int m = 0; m < 16; ++m) for (int n = 0; n < 16; ++n) { float s = 0; for (int k = 0; k < 4; ++k) s += hA[m * 4 + k] * hB[k * 16 + n]; if (s != hC[m * 16 + n]) ++err; }
    printf("mfma_16x16x4_f32 map: %s (%d mismatches)\n", err ? "FAIL" : "ok", err); bad += err;
  }
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device: %s, CUs %d, LDS/block %zu, clock %d kHz, gcn %s\n", p.name, p.multiProcessorCount, p.sharedMemPerBlock, p.clockRate, p.gcnArchName);
  printf(bad ? "PROBE FAIL\n" : "PROBE OK\n");
  return bad ? 1 : 0;
}
