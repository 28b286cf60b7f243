// k_conv_bf16_v2.h -- 3x3 conv, bf16, persistent workgroups with LDS-DMA double buffering (lab version).
//
// Workgroup = 512 lanes = 8 waves, one per CU (152 KB of LDS).  Work item = (image, 32x16-pixel tile, 64-cout group);
// a workgroup walks a contiguous range of items, the (item, 32-channel chunk) stream is flattened and pipelined:
// while the MFMAs of step i read LDS buffer i&1, the `global_load_lds_dwordx4` of step i+1 fill buffer (i+1)&1.
// One raw s_barrier per step; the DMA is retired with a counted s_waitcnt before the barrier.
// LDS image per buffer: 612 halo-pixel rows then 576 weight rows ([tap][64 couts]) of 32 channels (64 B), XOR slot
// swizzle as in k_unet_bf16.h -- applied on the SOURCE address (the DMA writes lane-linear) and on the fragment read.
// Out-of-image halo pixels read a 64-byte page of zeros.
#pragma once
#include "k_unet_bf16.h"

namespace sh {

#define V2_THREADS 512
#define V2_PW 34
#define V2_PH 18
#define V2_INROWS 612
#define V2_ROWS 1188
#define V2_BUF (V2_ROWS * 64)
#define V2_SLOTS (V2_ROWS * 4)
#define V2_BIAS_OFF (2 * V2_BUF)
#define V2_SMEM (2 * V2_BUF + 2048)

typedef const __attribute__((address_space(1))) void* v2_gptr;
typedef __attribute__((address_space(3))) void* v2_lptr;

__global__ void __launch_bounds__(V2_THREADS)
k_conv3_bf16_v2(const __bf16* __restrict__ src0, const __bf16* __restrict__ src1, int C0, int C1,
                const __bf16* __restrict__ wgt, const float* __restrict__ bias, __bf16* __restrict__ dst,
                int H, int W, int Cout, int relu, int nimg, const __bf16* __restrict__ zero_page) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[V2_SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int xh = wave & 1, rg = wave >> 1;
  const int Cin = C0 + C1, nchunk = Cin >> 5;
  const int tiles_x = W / 32, tiles_y = H / 16, ngroups = Cout >> 6;
  const int total = nimg * tiles_x * tiles_y * ngroups;
  const int per = (total + gridDim.x - 1) / gridDim.x;
  const int w_begin = blockIdx.x * per, w_end = min(total, w_begin + per);
  if (w_begin >= w_end) return;

  float* s_bias = (float*)(smem + V2_BIAS_OFF);
  for (int i = tid; i < Cout; i += V2_THREADS) s_bias[i] = bias[i];
  __syncthreads();       // every ordinary load is retired before the first LDS-DMA is issued

  // ---- static staging plan of this lane: slot e = tid + 512 k of the buffer image
  int q8[5], pyx[5], wrel[6];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    int e = tid + 512 * k, r = e >> 2, sp = e & 3;
    int q = sp ^ ((r >> 1) & 2);
    int py = r / V2_PW, px = r - py * V2_PW;
    q8[k] = q * 8;
    pyx[k] = py | (px << 8);
  }
#pragma unroll
  for (int k = 4; k < 10; ++k) {
    int e = tid + 512 * k, r = e >> 2, sp = e & 3;
    int q = sp ^ ((r >> 1) & 2);
    int rw = r - V2_INROWS;
    int tap = rw >> 6, j = rw & 63;
    wrel[k - 4] = (tap * nchunk * Cout + j) * 32 + q * 8;
  }
  const bool in4 = ((tid + 2048) >> 2) < V2_INROWS;     // slot k = 4 of this lane is a halo row (else a weight row)
  const bool w9 = tid + 512 * 9 < V2_SLOTS;

  // ---- issue-side item counters
  int i_g, i_tx, i_ty, i_img;
  {
    int w = w_begin;
    i_g = w % ngroups; w /= ngroups;
    i_tx = w % tiles_x; w /= tiles_x;
    i_ty = w % tiles_y; i_img = w / tiles_y;
  }
  int pixoff[5];
  auto item_lane_setup = [&]() {
#pragma unroll
    for (int k = 0; k < 5; ++k) {
      int py = pyx[k] & 255, px = pyx[k] >> 8;
      int gx = i_tx * 32 + px - 1, gy = i_ty * 16 + py - 1;
      bool ok = gx >= 0 && gx < W && gy >= 0 && gy < H;
      pixoff[k] = ok ? gy * W + gx : -1;
    }
  };
  auto issue = [&](int cc, int buf) {
    const int c0 = cc * 32;
    const bool first = c0 < C0;
    const int Cs = first ? C0 : C1, cb = first ? c0 : c0 - C0;
    const __bf16* simg = (first ? src0 : src1) + (size_t)i_img * H * W * Cs;
    const __bf16* wbase = wgt + ((size_t)cc * Cout + i_g * 64) * 32;
    unsigned char* lbase = smem + buf * V2_BUF + wave * 1024;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const __bf16* p = pixoff[k] >= 0 ? simg + (unsigned)(pixoff[k] * Cs + cb + q8[k]) : zero_page;
      __builtin_amdgcn_global_load_lds((v2_gptr)p, (v2_lptr)(lbase + k * 8192), 16, 0, 0);
    }
    {
      const __bf16* pi = pixoff[4] >= 0 ? simg + (unsigned)(pixoff[4] * Cs + cb + q8[4]) : zero_page;
      const __bf16* pw = wbase + wrel[0];
      const __bf16* p = in4 ? pi : pw;
      __builtin_amdgcn_global_load_lds((v2_gptr)p, (v2_lptr)(lbase + 4 * 8192), 16, 0, 0);
    }
#pragma unroll
    for (int k = 5; k < 9; ++k)
      __builtin_amdgcn_global_load_lds((v2_gptr)(wbase + wrel[k - 4]), (v2_lptr)(lbase + k * 8192), 16, 0, 0);
    if (w9) __builtin_amdgcn_global_load_lds((v2_gptr)(wbase + wrel[5]), (v2_lptr)(lbase + 9 * 8192), 16, 0, 0);
  };

  item_lane_setup();
  issue(0, 0);
  int buf = 0;
  for (int w = w_begin; w < w_end; ++w) {
    const int c_x0 = i_tx * 32, c_y0 = i_ty * 16, c_img = i_img, c_n0 = i_g * 64;
    f32x4 acc[4][4];
#pragma unroll
    for (int n = 0; n < 4; ++n) {
      f32x4 bv = *(const f32x4*)(s_bias + c_n0 + n * 16 + lk * 4);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m][n] = bv;
    }
    for (int cc = 0; cc < nchunk; ++cc) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (cc + 1 < nchunk) {
        issue(cc + 1, buf ^ 1);
      } else if (w + 1 < w_end) {
        if (++i_g == ngroups) { i_g = 0; if (++i_tx == tiles_x) { i_tx = 0; if (++i_ty == tiles_y) { i_ty = 0; ++i_img; } } }
        item_lane_setup();
        issue(0, buf ^ 1);
      }
      const __bf16* sb = (const __bf16*)(smem + buf * V2_BUF);
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
        bf16x8 xf[4], wf[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) { const int row = (rg * 4 + m + dy) * V2_PW + xh * 16 + li + dx; xf[m] = *(const bf16x8*)(sb + UB_OFF(row, lk)); }
#pragma unroll
        for (int n = 0; n < 4; ++n) { const int row = V2_INROWS + tap * 64 + n * 16 + li; wf[n] = *(const bf16x8*)(sb + UB_OFF(row, lk)); }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xf[m], acc[m][n], 0, 0, 0);
      }
      buf ^= 1;
    }
    __bf16* out = dst + (size_t)c_img * H * W * Cout;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int gy = c_y0 + rg * 4 + m, gx = c_x0 + xh * 16 + li;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        bf16x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[m][n][r];
          if (relu) v = fmaxf(v, 0.0f);
          o[r] = (__bf16)v;
        }
        *(bf16x4*)(out + ((size_t)gy * W + gx) * Cout + c_n0 + n * 16 + lk * 4) = o;
      }
    }
  }
}


static __bf16* v2_zero_page() {
  static __bf16* z = nullptr;
  if (!z) { (void)hipMalloc(&z, 256); (void)hipMemset(z, 0, 256); }
  return z;
}

static void launch_conv_v2(const __bf16* src0, const __bf16* src1, int C0, int C1, const __bf16* wpk, const float* bias, __bf16* dst,
                           int H, int W, int Cout, int nimg, int relu, hipStream_t st) {
  const int total = nimg * (W / 32) * (H / 16) * (Cout / 64);
  int grid = total < 256 ? total : 256;
  hipLaunchKernelGGL(k_conv3_bf16_v2, dim3(grid), dim3(V2_THREADS), 0, st, src0, src1, C0, C1, wpk, bias, dst, H, W, Cout, relu, nimg, v2_zero_page());
}

}  // namespace sh
