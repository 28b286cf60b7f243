// The step structure of k_conv3_ldr16 rebuilt piece by piece around a bare MFMA stream: which part costs what.
//   IDLE  0: 256-lane workgroups, 1 wave per SIMD      1: 512 lanes, waves 4..7 only join the step barrier
//   FRAG  0: operands stay in registers                1: 14 ds_read_b128 per step      2: 66 per step (dx-major, as the kernel)
//   DMA   0: none   1: loaders fill 80 KB per step from an L2-resident block   2: halo rows streamed from a 4 GB tensor + weights from a shared 37 KB block
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef const __attribute__((address_space(1))) void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define PW 36
#define INROWS (18 * PW)
#define ROWS (INROWS + 576)
#define BUFB (ROWS * 64)
#define OFF(row, slot) ((row) * 32 + (((slot) ^ (((row) >> 1) & 2)) << 3))

template <int IDLE, int FRAG, int DMA>
__global__ void __launch_bounds__(512) k(const unsigned short* __restrict__ src, const unsigned short* __restrict__ wsh, float* out, int nsteps, long long img_stride, int W,
                                         unsigned long long* clk) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * BUFB + 2048];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= 4;
  const int li = lane & 15, lk = lane >> 4;
  // fill LDS with pseudo-random bf16 in [-1, 1) so the fragments toggle like data
  unsigned seed = tid * 2654435761u + blockIdx.x * 40503u + 977u;
  for (int i = tid; i < (2 * BUFB) / 4; i += blockDim.x) {
    seed = seed * 1664525u + 1013904223u;
    const unsigned hi = 0x3C00u + ((seed >> 9) & 0x3FFu) + ((seed >> 3) & 0x8000u), lo = 0x3C00u + ((seed >> 19) & 0x3FFu) + ((seed >> 2) & 0x8000u);
    ((unsigned*)smem)[i] = (hi & 0xFFFFu) << 16 | (lo & 0xFFFFu);
  }
  __syncthreads();
  const int xh = wave & 1, rg8 = (wave >> 1) & 1;
  int xoff[2][3];
  const int rowbase = rg8 * 8 * PW + xh * 16 + li;
  for (int sp = 0; sp < 2; ++sp) for (int dx = 0; dx < 3; ++dx) xoff[sp][dx] = OFF(rowbase + sp * PW + dx, lk) * 2;
  const int woff = OFF(INROWS + li, lk) * 2;
  // loader plan
  const int ltid = tid - 256, lw = wave - 4;
  const int r0 = (ltid >> 2) & 63;
  const int q8 = ((ltid & 3) ^ ((r0 >> 1) & 2)) * 8;
  f4 acc[8][4];
  for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) acc[m][n] = f4{0, 0, 0, 0};
  bf8 xq0[10], wf0[4];
  for (int s = 0; s < 10; ++s) xq0[s] = *(const bf8*)(smem + xoff[s & 1][0] + (s & ~1) * PW * 64);
  for (int n = 0; n < 4; ++n) wf0[n] = *(const bf8*)(smem + woff + n * 16 * 64);
  unsigned long long c0 = clock64(), w0 = wall_clock64();
  int buf = 0;
  long long tile = blockIdx.x;
  for (int step = 0; step < nsteps; ++step) {
    if (IDLE) {
      if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if (loader) {
      if (DMA) {
        unsigned char* lbase = smem + (buf ^ 1) * BUFB + lw * 1024;
        if (DMA == 1) {
          const unsigned short* p = src + (size_t)blockIdx.x * (BUFB / 2) + (size_t)(ltid * 8);
#pragma unroll
          for (int kk = 0; kk < 20; ++kk)
            if (kk < 19 || r0 < 8) __builtin_amdgcn_global_load_lds((gptr)(p + kk * 2048), (lptr)(lbase + kk * 4096), 16, 0, 0);
        } else {
          // halo: 18 rows x 36 pixels x 64 B out of an image plane of width W (rows W * 64 B apart); a new tile every step
          const unsigned short* simg = src + (tile % 4096) * img_stride + ((tile / 4096) % 8) * 16 * (long long)W * 32;
#pragma unroll
          for (int kk = 0; kk < 20; ++kk) {
            const int r = r0 + 64 * kk;
            if (kk < 10 || (kk == 10 && r0 < 8)) {
              const int py = r / PW, px = r - py * PW;
              __builtin_amdgcn_global_load_lds((gptr)(simg + ((size_t)(py * W + px) * 32 + q8)), (lptr)(lbase + kk * 4096), 16, 0, 0);
            } else if (kk < 19 || r0 < 8) {
              __builtin_amdgcn_global_load_lds((gptr)(wsh + ((size_t)(r - INROWS) * 32 + q8)), (lptr)(lbase + kk * 4096), 16, 0, 0);
            }
          }
          tile += gridDim.x;
        }
      }
    } else {
      const unsigned char* sb = smem + buf * BUFB;
      if (FRAG == 0) {
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf0[n], xq0[m + t % 3], acc[m][n], 0, 0, 0);
      } else if (FRAG == 1) {
        bf8 xq[10], wf[4];
#pragma unroll
        for (int s = 0; s < 10; ++s) xq[s] = *(const bf8*)(sb + xoff[s & 1][0] + (s & ~1) * PW * 64);
#pragma unroll
        for (int n = 0; n < 4; ++n) wf[n] = *(const bf8*)(sb + woff + n * 16 * 64);
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
          for (int m = 0; m < 8; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xq[m + t % 3], acc[m][n], 0, 0, 0);
      } else {
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          bf8 xq[10];
#pragma unroll
          for (int s = 0; s < 10; ++s) xq[s] = *(const bf8*)(sb + xoff[s & 1][dx] + (s & ~1) * PW * 64);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            const int tap = dy * 3 + dx;
            bf8 wf[4];
#pragma unroll
            for (int n = 0; n < 4; ++n) wf[n] = *(const bf8*)(sb + woff + (tap * 64 + n * 16) * 64);
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
              for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xq[m + dy], acc[m][n], 0, 0, 0);
          }
        }
      }
    }
    if (DMA) buf ^= 1;
  }
  unsigned long long c1 = clock64(), w1 = wall_clock64();
  if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (!loader) {
    float s = 0;
    for (int m = 0; m < 8; ++m) for (int n = 0; n < 4; ++n) s += acc[m][n][0] + acc[m][n][1] + acc[m][n][2] + acc[m][n][3];
    out[blockIdx.x * 256 + tid] = s;
  }
  if (tid == 0 && blockIdx.x == 0) { clk[0] = c1 - c0; clk[1] = w1 - w0; }
}

__global__ void k_fill(unsigned* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned s = (unsigned)i * 2654435761u + 12345u; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
    p[i] = ((0x3C00u + (s & 0x3FFu) + ((s >> 3) & 0x8000u)) & 0xFFFFu) << 16 | ((0x3C00u + ((s >> 10) & 0x3FFu) + ((s >> 2) & 0x8000u)) & 0xFFFFu);
  }
}

template <int IDLE, int FRAG, int DMA>
static void run(const char* what, const unsigned short* src, const unsigned short* wsh, float* out, unsigned long long* clk, long long img_stride, int W) {
  const int nsteps = 2000, launches = 4;
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHK(hipEventRecord(e0));
    for (int l = 0; l < launches; ++l) hipLaunchKernelGGL((k<IDLE, FRAG, DMA>), dim3(256), dim3(IDLE ? 512 : 256), 0, 0, src, wsh, out, nsteps, img_stride, W, clk);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    unsigned long long h[2]; CHK(hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost));
    const double flop = (double)launches * 256 * 4 * nsteps * 288 * 16384.0;
    if (rep) printf("%-58s %7.2f ms  %5.0f TFLOP/s  %.3f of 2500   %4.0f MHz   %.2f us/step\n", what, ms, flop / ms / 1e9, flop / ms / 1e9 / 2500, (double)h[0] / ((double)h[1] / 100.0),
                    ms * 1e3 / launches / nsteps);
  }
}

int main() {
  const int W = 512;                       // level-0-like plane: 512 x 512 pixels x 32 channels x 2 B = 16 MB per (image, chunk)
  const long long img_stride = 512LL * 512 * 32;
  unsigned short *src, *wsh; float* out; unsigned long long* clk;
  const size_t src_elems = 4096ULL * img_stride / 16 + (1 << 24);      // 4096 planes would be 64 GB: wrap to 4 GB
  (void)src_elems;
  const size_t nb = 4ULL << 30;
  CHK(hipMalloc(&src, nb)); hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)src, nb / 4);
  CHK(hipMalloc(&wsh, 1 << 20)); hipLaunchKernelGGL(k_fill, dim3(64), dim3(256), 0, 0, (unsigned*)wsh, (size_t)(1 << 18)); CHK(hipDeviceSynchronize());
  CHK(hipMalloc(&out, 256 * 256 * 4)); CHK(hipMalloc(&clk, 16));
  const long long st = (nb / 2 - 8LL * 16 * W * 32 - 18LL * W * 32) / 4096 / 8 * 8;      // plane stride so that 4096 "images" fit in 4 GB
  run<0, 0, 0>("bare: 256 lanes, operands in registers", src, wsh, out, clk, st, W);
  run<1, 0, 0>("+ 4 idle waves and a barrier per 288 MFMAs", src, wsh, out, clk, st, W);
  run<1, 1, 0>("+ 14 fragment reads per step", src, wsh, out, clk, st, W);
  run<1, 2, 0>("+ 66 fragment reads per step (dx-major)", src, wsh, out, clk, st, W);
  run<1, 2, 1>("+ LDS-DMA 80 KB per step, L2-resident source", src, wsh, out, clk, st, W);
  run<1, 2, 2>("+ LDS-DMA: halo streamed from HBM, weights shared", src, wsh, out, clk, st, W);
  run<1, 0, 2>("LDS-DMA streamed + MFMAs on register operands", src, wsh, out, clk, st, W);
  return 0;
}
