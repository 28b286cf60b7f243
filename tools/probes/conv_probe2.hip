// A second step structure for the 3x3 convolution, priced like conv_probe.hip prices k_conv3_ldr16's (random data, all 256 CUs):
// one wave per SIMD with the whole register file (256-lane workgroups, 256 accumulators per lane), 16-channel half-steps on
// v_mfma_f32_32x32x16_bf16 (one tap x 16 channels = one K), every wave issues its own share of the LDS-DMA pieces.
//   tile per CU: 16 rows x 32 pixels x 128 couts; a wave owns 8 rows x 32 pixels x 64 couts = 16 accumulator blocks of 32 x 32
//   half-step:   halo 18 x 34 pixels x 16 ch (19.6 KB) + weights 9 taps x 16 ch x 128 couts (36.9 KB) = 56.5 KB per 4 x 144 MFMAs
//                (= 1 152 MFMAs of 16x16x32: k_conv3_ldr16 feeds 78 KB for the same), 48 fragment reads per wave (66 there)
//   DMA 0: none (fragments from a resident buffer)   1: pieces from an L2-resident block   2: halo streamed from a 4 GB tensor, weights shared
//   MF 0: 32x32x16 (K = 16)   1: the same bytes and reads on 16x16x32 (NOT a valid mapping of a 16-channel half-step: the matrix-pipe
//         side of the comparison only)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(16))) float f16v;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define HPW 36                          // halo pitch in pixels
#define HROWS (18 * HPW)                // 648 pixels of 32 B
#define HBYTES (HROWS * 32)             // 20 736
#define WBYTES (9 * 128 * 32)           // 36 864: [tap][cout][16 ch]
#define BUFB (HBYTES + WBYTES)          // 57 600
#define NPIECE ((BUFB + 4095) / 4096)   // 15 pieces of 4 KB per step (256 lanes x 16 B), the last one partial

__device__ inline void dma16(unsigned lds_dst, const void* p) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" :: "s"(__builtin_amdgcn_readfirstlane(lds_dst)), "v"(p) : "memory");
}

template <int DMA, int MF>
__global__ void __launch_bounds__(256) kx(const unsigned short* __restrict__ src, const unsigned short* __restrict__ wsh, float* out, int nsteps, long long img_stride, int W,
                                          unsigned long long* clk) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUFB + 4096];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  unsigned seed = tid * 2654435761u + blockIdx.x * 40503u + 977u;
  for (int i = tid; i < (2 * BUFB) / 4; i += blockDim.x) {
    seed = seed * 1664525u + 1013904223u;
    const unsigned hi = 0x3C00u + ((seed >> 9) & 0x3FFu) + ((seed >> 3) & 0x8000u), lo = 0x3C00u + ((seed >> 19) & 0x3FFu) + ((seed >> 2) & 0x8000u);
    ((unsigned*)smem)[i] = (hi & 0xFFFFu) << 16 | (lo & 0xFFFFu);
  }
  __syncthreads();
  const int rg = wave >> 1, ch = wave & 1;       // rows 8 rg .. 8 rg + 7, couts 64 ch .. 64 ch + 63
  // A fragment of input row i, shift dx: pixel (lane & 31) + dx of the row, k-group lane >> 5 (8 channels = 16 B)
  const int aoff = ((rg * 8) * HPW + (lane & 31)) * 32 + (lane >> 5) * 16;
  // B fragment of tap t, cout block n (32 couts): cout 64 ch + 32 n + (lane & 31), k-group lane >> 5
  const int boff = HBYTES + ((64 * ch + (lane & 31)) * 32) + (lane >> 5) * 16;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) void*)smem;
  f16v acc[8][2];
  for (int m = 0; m < 8; ++m) for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) acc[m][n][i] = 0.f;
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  int buf = 0;
  long long tile = blockIdx.x;
  // piece plan: slot e = tid + 256 k of the step's 3 600 slots of 16 B; halo slots first (1 296), then weights (2 304)
  for (int step = 0; step < nsteps; ++step) {
    if (DMA) {
      const unsigned lb = lds0 + (unsigned)((buf ^ 1) * BUFB) + (unsigned)(wave * 1024);
      if (DMA == 1) {
        const unsigned short* p = src + (size_t)blockIdx.x * (BUFB / 2) + (size_t)(tid * 8);
#pragma unroll
        for (int kk = 0; kk < NPIECE; ++kk)
          if (kk < NPIECE - 1 || tid * 16 + kk * 4096 < BUFB) dma16(lb + kk * 4096, p + kk * 2048);
      } else {
        const unsigned short* simg = src + (tile % 4096) * img_stride + ((tile / 4096) % 8) * 16 * (long long)W * 32;
#pragma unroll
        for (int kk = 0; kk < NPIECE; ++kk) {
          const int e = tid + 256 * kk;
          if (e < 2 * HROWS) {                       // halo: half a 64-byte pixel row (16 of its 32 channels), two slots per pixel
            const int r = e >> 1, py = r / HPW, px = r - py * HPW;
            dma16(lb + kk * 4096, simg + ((size_t)(py * W + px) * 32 + (e & 1) * 8 + (step & 1) * 16));
          } else if (e * 16 < BUFB) {
            dma16(lb + kk * 4096, wsh + (size_t)(e - 2 * HROWS) * 8);
          }
        }
        if (step & 1) tile += gridDim.x;             // (two half-steps read the two halves of the same pixel rows)
      }
    }
    const unsigned char* sb = smem + buf * BUFB;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      bf8 xq[10];
#pragma unroll
      for (int s = 0; s < 10; ++s) xq[s] = *(const bf8*)(sb + aoff + (s * HPW + dx) * 32);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        bf8 wf[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) wf[n] = *(const bf8*)(sb + boff + ((dy * 3 + dx) * 128 + 32 * n) * 32);
        if (MF == 0) {
#pragma unroll
          for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wf[n], xq[m + dy], acc[m][n], 0, 0, 0);
        } else {
          // the same 2 x 16 384 flop per (m, n) on the 16x16x32 shape: four MFMAs into the four quarters of the block
#pragma unroll
          for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) {
              f4 q0 = {acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]}, q1 = {acc[m][n][4], acc[m][n][5], acc[m][n][6], acc[m][n][7]};
              q0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xq[m + dy], q0, 0, 0, 0);
              q1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xq[m + dy], q1, 0, 0, 0);
              acc[m][n][0] = q0[0]; acc[m][n][1] = q0[1]; acc[m][n][2] = q0[2]; acc[m][n][3] = q0[3];
              acc[m][n][4] = q1[0]; acc[m][n][5] = q1[1]; acc[m][n][6] = q1[2]; acc[m][n][7] = q1[3];
            }
        }
      }
    }
    if (DMA) {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      buf ^= 1;
    }
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  float s = 0;
  for (int m = 0; m < 8; ++m) for (int n = 0; n < 2; ++n) for (int i = 0; i < 16; ++i) s += acc[m][n][i];
  out[blockIdx.x * 256 + tid] = s;
  if (tid == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

__global__ void k_fill(unsigned* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned s = (unsigned)i * 2654435761u + 12345u; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
    p[i] = ((0x3C00u + (s & 0x3FFu) + ((s >> 3) & 0x8000u)) & 0xFFFFu) << 16 | ((0x3C00u + ((s >> 10) & 0x3FFu) + ((s >> 2) & 0x8000u)) & 0xFFFFu);
  }
}

template <int DMA, int MF>
static void run(const char* what, const unsigned short* src, const unsigned short* wsh, float* out, unsigned long long* clk, long long img_stride, int W) {
  const int nsteps = 4000, launches = 4;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHK(hipEventRecord(e0));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((kx<DMA, MF>), dim3(256), dim3(256), 0, 0, src, wsh, out, nsteps, img_stride, W, clk);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double flop = (double)launches * 256 * 4 * nsteps * 144 * 32768.0;
    if (rep) printf("%-72s %7.2f ms  %5.0f TFLOP/s  %.3f of 2500   %4.0f MHz   %.2f us/half-step\n", what, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500, (double)h[0] / ((double)h[1] / 100.0),
                    ms * 1e3 / launches / nsteps);
  }
}

int main() {
  const int W = 512;
  unsigned short *src, *wsh; float* out; unsigned long long* clk;
  const size_t nb = 4ULL << 30;
  CHK(hipMalloc(&src, nb)); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)src, nb / 4);
  CHK(hipMalloc(&wsh, 1 << 20)); hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, (unsigned*)wsh, (size_t)(1 << 18)); CHK(hipDeviceSynchronize());
  CHK(hipMalloc(&out, 256 * 256 * 4)); CHK(hipMalloc(&clk, 16));
  const long long st = (nb / 2 - 8LL * 16 * W * 32 - 18LL * W * 32) / 4096 / 8 * 8;
  run<0, 0>("32x32x16, 1 wave / SIMD, 48 fragment reads per half-step, no fill", src, wsh, out, clk, st, W);
  run<1, 0>("+ 56 KB of LDS-DMA per half-step by the computing waves, L2-resident", src, wsh, out, clk, st, W);
  run<2, 0>("+ halo halves streamed from HBM, weights shared", src, wsh, out, clk, st, W);
  run<0, 1>("the same reads on 16x16x32 (matrix-pipe side only), no fill", src, wsh, out, clk, st, W);
  run<2, 1>("16x16x32 + streamed fill", src, wsh, out, clk, st, W);
  return 0;
}
