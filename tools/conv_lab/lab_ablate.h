// lab_ablate.h -- timing-only copies of k_conv_mfma_bf16<9,4> with one cost removed each (results are wrong on purpose).
//   ABL 1: no global loads after the first chunk      2: no LDS stores / barriers after the first chunk
//   ABL 3: 1 + 2                                     4: 3 + no ds_read in the tap loop (MFMA only)
//   ABL 5: everything kept, but no MFMA (loads / stores / reads only)
#pragma once
#include "k_unet_bf16.h"

namespace sh {

template <int ABL>
__global__ void __launch_bounds__(UN_THREADS)
k_conv_ablate(const __bf16* __restrict__ src0, int C0, const __bf16* __restrict__ wgt, const float* __restrict__ bias, __bf16* __restrict__ dst,
              int H, int W, int Cout, int relu) {
  constexpr int TAPS = 9, NT = 4;
  constexpr int HALO = 1;
  constexpr int PW = UN_TW + 2 * HALO, PH = UN_TH + 2 * HALO;
  constexpr int NC = 16 * NT;
  constexpr int IN_PIECES = PH * PW * 4, WT_PIECES = TAPS * NC * 4;
  constexpr int NIN = (IN_PIECES + UN_THREADS - 1) / UN_THREADS, NWT = (WT_PIECES + UN_THREADS - 1) / UN_THREADS;
  __shared__ __attribute__((aligned(16))) __bf16 s_in[PH * PW * UB_PSTR];
  __shared__ __attribute__((aligned(16))) __bf16 s_w[TAPS * NC * UB_PSTR];
  const int Cin = C0;
  const int tiles_x = W / UN_TW;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int n0 = blockIdx.y * NC;
  const int img = blockIdx.z;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int x0 = tx * UN_TW, y0 = ty * UN_TH;
  const __bf16* in0 = src0 + (size_t)img * H * W * C0;
  const int nchunk = Cin / 32;
  const __bf16* wp = wgt;

  int in_pix[NIN], in_lds[NIN], wt_off[NWT], wt_lds[NWT];
#pragma unroll
  for (int k = 0; k < NIN; ++k) {
    int e = tid + k * UN_THREADS;
    int q = e & 3, p = e >> 2;
    int px = p % PW, py = p / PW;
    int gx = x0 + px - HALO, gy = y0 + py - HALO;
    bool ok = e < IN_PIECES && gx >= 0 && gx < W && gy >= 0 && gy < H;
    in_pix[k] = ok ? (gy * W + gx) : -1;
    in_lds[k] = e < IN_PIECES ? UB_OFF(p, q) : -1;
  }
#pragma unroll
  for (int k = 0; k < NWT; ++k) {
    int e = tid + k * UN_THREADS;
    int q = e & 3, r = e >> 2;
    int tap = r / NC, j = r % NC;
    wt_off[k] = e < WT_PIECES ? ((tap * nchunk) * Cout + n0 + j) * 32 + q * 8 : -1;
    wt_lds[k] = e < WT_PIECES ? UB_OFF(r, q) : -1;
  }
  u32x4 rin[NIN], rwt[NWT];
  auto load_chunk = [&](int cc) {
    const int c0 = cc * 32;
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      u32x4 v = (u32x4){0u, 0u, 0u, 0u};
      if (in_pix[k] >= 0) v = *(const u32x4*)(in0 + (size_t)in_pix[k] * C0 + c0 + ((tid + k * UN_THREADS) & 3) * 8);
      rin[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NWT; ++k)
      if (wt_off[k] >= 0) rwt[k] = *(const u32x4*)(wp + (size_t)wt_off[k] + (size_t)cc * Cout * 32);
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int k = 0; k < NIN; ++k) if (in_lds[k] >= 0) *(u32x4*)(s_in + in_lds[k]) = rin[k];
#pragma unroll
    for (int k = 0; k < NWT; ++k) if (wt_lds[k] >= 0) *(u32x4*)(s_w + wt_lds[k]) = rwt[k];
  };

  f32x4 acc[4][NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[n0 + n * 16 + lk * 4 + r];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m][n] = bv;
  }
  constexpr bool NOLOAD = ABL == 1 || ABL == 3 || ABL == 4;
  constexpr bool NOSTORE = ABL == 2 || ABL == 3 || ABL == 4;
  constexpr bool NOREAD = ABL == 4;
  constexpr bool NOMFMA = ABL == 5;

  bf16x8 xf0[4], wf0[NT];
  load_chunk(0);
  for (int cc = 0; cc < nchunk; ++cc) {
    if (!NOSTORE || cc == 0) {
      __syncthreads();
      store_chunk();
      __syncthreads();
    }
    if (cc + 1 < nchunk && !NOLOAD) load_chunk(cc + 1);
    if (NOREAD && cc == 0) {
#pragma unroll
      for (int m = 0; m < 4; ++m) { const int row = (wave * 4 + m) * PW + li; xf0[m] = *(const bf16x8*)(s_in + UB_OFF(row, lk)); }
#pragma unroll
      for (int n = 0; n < NT; ++n) { const int row = n * 16 + li; wf0[n] = *(const bf16x8*)(s_w + UB_OFF(row, lk)); }
    }
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      bf16x8 xf[4], wf[NT];
      if (!NOREAD) {
#pragma unroll
        for (int m = 0; m < 4; ++m) { const int row = (wave * 4 + m + dy) * PW + li + dx; xf[m] = *(const bf16x8*)(s_in + UB_OFF(row, lk)); }
#pragma unroll
        for (int n = 0; n < NT; ++n) { const int row = tap * NC + n * 16 + li; wf[n] = *(const bf16x8*)(s_w + UB_OFF(row, lk)); }
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m) xf[m] = xf0[m];
#pragma unroll
        for (int n = 0; n < NT; ++n) wf[n] = wf0[n];
      }
      if (!NOMFMA) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xf[m], acc[m][n], 0, 0, 0);
      } else {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) {     // keep the fragment reads alive at VALU cost ~0
            acc[m][n][0] += __builtin_bit_cast(float, ((u32x4)__builtin_bit_cast(u32x4, wf[n]))[0] ^ ((u32x4)__builtin_bit_cast(u32x4, xf[m]))[1]);
          }
      }
    }
  }
  __bf16* out = dst + (size_t)img * H * W * Cout;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    int gy = y0 + wave * 4 + m, gx = x0 + li;
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      bf16x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[m][n][r];
        if (relu) v = fmaxf(v, 0.0f);
        o[r] = (__bf16)v;
      }
      *(bf16x4*)(out + ((size_t)gy * W + gx) * Cout + n0 + n * 16 + lk * 4) = o;
    }
  }
}

static void launch_ablate(int a, int tiles, const __bf16* src, int Cin, const __bf16* wpk, const float* bias, __bf16* dst, int H, int W, int Cout, int nimg) {
  dim3 g(tiles, Cout / 64, nimg), b(256);
  switch (a) {
    case 0: hipLaunchKernelGGL(k_conv_ablate<0>, g, b, 0, 0, src, Cin, wpk, bias, dst, H, W, Cout, 1); break;
    case 1: hipLaunchKernelGGL(k_conv_ablate<1>, g, b, 0, 0, src, Cin, wpk, bias, dst, H, W, Cout, 1); break;
    case 2: hipLaunchKernelGGL(k_conv_ablate<2>, g, b, 0, 0, src, Cin, wpk, bias, dst, H, W, Cout, 1); break;
    case 3: hipLaunchKernelGGL(k_conv_ablate<3>, g, b, 0, 0, src, Cin, wpk, bias, dst, H, W, Cout, 1); break;
    case 4: hipLaunchKernelGGL(k_conv_ablate<4>, g, b, 0, 0, src, Cin, wpk, bias, dst, H, W, Cout, 1); break;
    case 5: hipLaunchKernelGGL(k_conv_ablate<5>, g, b, 0, 0, src, Cin, wpk, bias, dst, H, W, Cout, 1); break;
    default: break;
  }
}

}  // namespace sh
