// A 32 -> 32 channel 3x3 conv step with a head-like epilogue as SEVERAL small workgroups per CU (4 waves, 32 x 8 pixel tile, two
// 23 KB halo buffers, weights in registers) instead of one 8-wave workgroup in lockstep: what does the matrix pipe reach?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef __attribute__((ext_vector_type(8))) __bf16 bf8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef const __attribute__((address_space(1))) void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)
#define PW 36
#define TR 8                       // tile rows
#define INROWS ((TR + 2) * PW)     // 360 halo pixels
#define BUFB (INROWS * 64)         // 23 040
#define OFF(row, slot) ((row) * 32 + (((slot) ^ (((row) >> 1) & 2)) << 3))

template <int EPI>      // 0: no epilogue, 1: ReLU + 32-channel dot + 2 shuffles + 4-byte store per row (the fused head), 2: ReLU + convert + 16-byte stores (a 32-channel output)
__global__ void __launch_bounds__(256) k(const unsigned short* __restrict__ src, float* __restrict__ logits, unsigned short* __restrict__ dst, int ntiles, int W, int H, int pad_lds) {
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int xh = wave & 1, rg = wave >> 1;      // 2 x 2 waves: 4 rows x 16 pixels each
  bf8 wreg[9][2];
  unsigned seed = tid * 2654435761u + 977u;
  for (int t = 0; t < 9; ++t) for (int n = 0; n < 2; ++n) for (int e = 0; e < 8; ++e) { seed = seed * 1664525u + 1013904223u; wreg[t][n][e] = (__bf16)(((int)(seed >> 9) - (1 << 22)) * (0.05f / (1 << 22))); }
  float hw[8];
  for (int e = 0; e < 8; ++e) hw[e] = 0.01f * (e + lk);
  int xoff[2][3];
  const int rowbase = rg * 4 * PW + xh * 16 + li;
  for (int sp = 0; sp < 2; ++sp) for (int dx = 0; dx < 3; ++dx) xoff[sp][dx] = OFF(rowbase + sp * PW + dx, lk) * 2;
  const int r0 = tid >> 2;                         // 0..63: LDS row of piece 0
  const int q8 = ((tid & 3) ^ ((r0 >> 1) & 2)) * 8;
  const int tiles_x = W / 32, tiles_y = H / TR;
  auto stage = [&](int t, int buf) {
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y; const long long img = t / (tiles_x * tiles_y);
    const unsigned short* simg = src + img * (long long)H * W * 32;
    unsigned char* lbase = smem + buf * BUFB + wave * 1024;
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) {
      const int r = r0 + 64 * kk;
      if (r < INROWS) {
        const int py = r / PW, px = r - py * PW;
        int gy = ty * TR + py - 1, gx = tx * 32 + px - 1;
        gy = gy < 0 ? 0 : gy >= H ? H - 1 : gy; gx = gx < 0 ? 0 : gx >= W ? W - 1 : gx;
        __builtin_amdgcn_global_load_lds((gptr)(simg + ((size_t)(gy * W + gx) * 32 + q8)), (lptr)(lbase + kk * 4096), 16, 0, 0);
      }
    }
  };
  int t = blockIdx.x, buf = 0;
  if (t < ntiles) stage(t, 0);
  for (; t < ntiles; t += gridDim.x) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                  // this tile has landed; everybody is done with the other buffer
    if (t + (int)gridDim.x < ntiles) stage(t + gridDim.x, buf ^ 1);
    const unsigned char* sb = smem + buf * BUFB;
    f4 acc[4][2];
    for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) acc[m][n] = f4{0.1f, 0.2f, 0.3f, 0.4f};
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      bf8 xq[6];
#pragma unroll
      for (int s = 0; s < 6; ++s) xq[s] = *(const bf8*)(sb + xoff[s & 1][dx] + (s & ~1) * PW * 64);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wreg[dy * 3 + dx][n], xq[m + dy], acc[m][n], 0, 0, 0);
    }
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y; const long long img = t / (tiles_x * tiles_y);
    if (EPI == 1) {
      float* lo = logits + img * (long long)H * W;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float s = 0.0f;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) s = __builtin_fmaf(fmaxf(acc[m][n][r], 0.0f), hw[4 * n + r], s);
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        lo[(size_t)(ty * TR + rg * 4 + m) * W + tx * 32 + xh * 16 + li] = s;
      }
    } else if (EPI == 2) {
      unsigned short* o = dst + img * (long long)H * W * 32;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        bf8 v;
#pragma unroll
        for (int r = 0; r < 8; ++r) v[r] = (__bf16)fmaxf(acc[m][r >> 2][r & 3], 0.0f);
        *(bf8*)(o + ((size_t)(ty * TR + rg * 4 + m) * W + tx * 32 + xh * 16 + li) * 32 + 8 * lk) = v;
      }
    } else {
      float s = 0;
      for (int m = 0; m < 4; ++m) for (int n = 0; n < 2; ++n) s += acc[m][n][0] + acc[m][n][3];
      if (s == 123.456f) logits[0] = s;
    }
    buf ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

__global__ void k_fill(unsigned* p, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    unsigned s = (unsigned)i * 2654435761u + 12345u; s ^= s >> 15; s *= 2246822519u; s ^= s >> 13;
    p[i] = ((0x3C00u + (s & 0x3FFu) + ((s >> 3) & 0x8000u)) & 0xFFFFu) << 16 | ((0x3C00u + ((s >> 10) & 0x3FFu) + ((s >> 2) & 0x8000u)) & 0xFFFFu);
  }
}

template <int EPI>
static void run(const char* what, int wg_per_cu, const unsigned short* src, float* logits, unsigned short* dst) {
  const int H = 512, W = 512, nimg = 64, ntiles = nimg * (H / TR) * (W / 32);
  // dynamic LDS: two buffers + padding so that exactly wg_per_cu workgroups fit in 160 KB
  const int lds = wg_per_cu == 3 ? 2 * BUFB + 1024 : wg_per_cu == 2 ? 70 * 1024 : 120 * 1024;
  CHK(hipFuncSetAttribute((const void*)k<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep) {
    CHK(hipEventRecord(e0));
    for (int l = 0; l < 4; ++l) hipLaunchKernelGGL((k<EPI>), dim3(256 * wg_per_cu), dim3(256), lds, 0, src, logits, dst, ntiles, W, H, 0);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms; CHK(hipEventElapsedTime(&ms, e0, e1));
    const double flop = 4.0 * ntiles * 4 * 72 * 16384.0;
    if (rep) printf("%-44s %d WG/CU  %6.3f ms per layer  %5.0f TFLOP/s  %.3f of 2500\n", what, wg_per_cu, ms / 4, flop / ms / 1e9, flop / ms / 1e9 / 2500);
  }
}

int main() {
  unsigned short *src, *dst; float* logits;
  const size_t nb = 64ULL * 512 * 512 * 64;      // 1.07 GB
  CHK(hipMalloc(&src, nb)); CHK(hipMalloc(&dst, nb)); CHK(hipMalloc(&logits, 64ULL * 512 * 512 * 4));
  hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, (unsigned*)src, nb / 4); CHK(hipDeviceSynchronize());
  for (int w = 1; w <= 3; ++w) run<0>("no epilogue", w, src, logits, dst);
  for (int w = 1; w <= 3; ++w) run<1>("head epilogue (dot + 2 shuffles + 4 B store)", w, src, logits, dst);
  for (int w = 1; w <= 3; ++w) run<2>("32-channel output (convert + 16 B stores)", w, src, logits, dst);
  return 0;
}
