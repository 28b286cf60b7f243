// bare MFMA streams on every CU: which instruction shape sustains what under the chip's power management
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__device__ inline float rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) - (1 << 23)) * (1.0f / (1 << 23)); }
template <int NACC, int mode>
__global__ void __launch_bounds__(512) k16(float* out, int iters, unsigned long long* clk) {
  bf8 a[4], b[4];
  unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 8; ++i) {
      if (mode == 0) { a[j][i] = (__bf16)(float)(threadIdx.x % 7 + i); b[j][i] = (__bf16)(float)(threadIdx.x % 5 - i); }
      else { a[j][i] = (__bf16)rnd(seed); b[j][i] = (__bf16)(rnd(seed) * 0.05f); }
    }
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = f4{0, 0, 0, 0};
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[mode == 2 ? i & 3 : 0], b[mode == 2 ? (i >> 2) & 3 : 0], acc[i], 0, 0, 0);
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
template <int NACC, int mode>
__global__ void __launch_bounds__(512) k32(float* out, int iters, unsigned long long* clk) {
  bf8 a[4], b[4];
  unsigned seed = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 8; ++i) {
      if (mode == 0) { a[j][i] = (__bf16)(float)(threadIdx.x % 7 + i); b[j][i] = (__bf16)(float)(threadIdx.x % 5 - i); }
      else { a[j][i] = (__bf16)rnd(seed); b[j][i] = (__bf16)(rnd(seed) * 0.05f); }
    }
  f16v acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mode == 2 ? i & 3 : 0], b[mode == 2 ? (i >> 2) & 1 : 0], acc[i], 0, 0, 0);
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int j = 0; j < 16; ++j) s += acc[i][j];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}
int main() {
  float* out; unsigned long long *clk, h[2];
  CHK(hipMalloc(&out, 256 * 512 * 4)); CHK(hipMalloc(&clk, 16));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
   for (int mode = 0; mode < 3; ++mode)
    for (int shape = 0; shape < 2; ++shape)
      for (int thr = 256; thr <= 512; thr *= 2) {
        const int iters = 6000;
        // 32 accumulators of 16x16 = 8 of 32x32 = 128 registers
        CHK(hipEventRecord(e0));
        for (int k = 0; k < 8; ++k) {
#define L16(M) hipLaunchKernelGGL((k16<32, M>), dim3(256), dim3(thr), 0, 0, out, iters, clk)
#define L32(M) hipLaunchKernelGGL((k32<8, M>), dim3(256), dim3(thr), 0, 0, out, iters, clk)
          if (shape == 0) { if (mode == 0) L16(0); else if (mode == 1) L16(1); else L16(2); }
          else { if (mode == 0) L32(0); else if (mode == 1) L32(1); else L32(2); }
        }
        CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
        float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
        CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
        const double flop = 8.0 * 256 * (thr / 64) * (double)iters * (shape == 0 ? 32 * 16384.0 : 8 * 32768.0);
        printf("rep %d data %d %s waves/SIMD %d: %.2f ms  %.0f TFLOP/s (%.3f of 2500)  clock %.0f MHz\n", rep, mode, shape == 0 ? "16x16x32" : "32x32x16", thr / 256, ms,
               flop / ms / 1e9, flop / ms / 1e9 / 2500, (double)h[0] / ((double)h[1] / 100.0));
      }
  return 0;
}
