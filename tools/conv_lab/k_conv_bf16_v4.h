// k_conv_bf16_v4.h -- v3 with a halo pitch of 36 pixels: the swizzle bit of a fragment row then depends only on
// (row parity, dx, lane), so every ds_read_b128 is one of 7 per-lane base registers plus an immediate offset.
#pragma once
#include "k_conv_bf16_v2.h"

namespace sh {

#define V4_PW 36
#define V4_INROWS (18 * V4_PW)              // 648
#define V4_ROWS (V4_INROWS + 576)           // 1224
#define V4_BUF (V4_ROWS * 64)               // 78336
#define V4_SLOTS (V4_ROWS * 4)              // 4896
#define V4_BIAS_OFF (2 * V4_BUF)
#define V4_SMEM (2 * V4_BUF + 2048)         // 158720

template <int SPREAD, int PREF, int PIN, int ABL = 0>
__global__ void __launch_bounds__(V2_THREADS)
k_conv3_bf16_v4(const __bf16* __restrict__ src0, const __bf16* __restrict__ src1, int C0, int C1,
                const __bf16* __restrict__ wgt, const float* __restrict__ bias, __bf16* __restrict__ dst,
                int H, int W, int Cout, int relu, int nimg, const __bf16* __restrict__ zero_page) {
  __shared__ __attribute__((aligned(1024))) unsigned char smem[V4_SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int xh = wave & 1, rg = wave >> 1;
  const int Cin = C0 + C1, nchunk = Cin >> 5;
  const int tiles_x = W / 32, tiles_y = H / 16, ngroups = Cout >> 6;
  const int total = nimg * tiles_x * tiles_y * ngroups;
  const int per = (total + gridDim.x - 1) / gridDim.x;
  const int w_begin = blockIdx.x * per, w_end = min(total, w_begin + per);
  if (w_begin >= w_end) return;
  const long long t_core0 = clock64(), t_wall0 = wall_clock64();

  float* s_bias = (float*)(smem + V4_BIAS_OFF);
  for (int i = tid; i < Cout; i += V2_THREADS) s_bias[i] = bias[i];
  __syncthreads();

  // staging plan: slot e_k = tid + 512 k -> row r_k = (tid >> 2) + 128 k; k = 0..4 halo rows, k = 5 mixed, k = 6..9 weight rows
  const int r0 = tid >> 2;
  const int q8 = ((tid & 3) ^ ((r0 >> 1) & 2)) * 8;
  const int rw5 = r0 + 640 - V4_INROWS;
  const int wstep = 2 * nchunk * Cout * 32;
  const int wrel5 = ((rw5 >> 6) * nchunk * Cout + (rw5 & 63)) * 32 + q8;
  const bool in5 = rw5 < 0;
  const bool w9 = tid + 512 * 9 < V4_SLOTS;

  // fragment read offsets (bytes inside a buffer)
  int xoff[2][3], woff;
  {
    const int rowbase = rg * 4 * V4_PW + xh * 16 + li;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) xoff[sp][dx] = UB_OFF(rowbase + sp * V4_PW + dx, lk) * 2;
    woff = UB_OFF(V4_INROWS + li, lk) * 2;
  }

  int i_g, i_tx, i_ty, i_img;
  {
    int w = w_begin;
    i_g = w % ngroups; w /= ngroups;
    i_tx = w % tiles_x; w /= tiles_x;
    i_ty = w % tiles_y; i_img = w / tiles_y;
  }
  int pixoff[6];
  auto item_lane_setup = [&]() {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      int r = r0 + 128 * k;
      int py = r / V4_PW, px = r - py * V4_PW;
      int gx = i_tx * 32 + px - 1, gy = i_ty * 16 + py - 1;
      bool ok = px < 34 && gx >= 0 && gx < W && gy >= 0 && gy < H;
      pixoff[k] = ok ? gy * W + gx : -1;
    }
  };
  const __bf16* n_simg; const __bf16* n_wbase; int n_Cs, n_cb; unsigned char* n_lbase;
  auto describe = [&](int cc, int buf) {
    const int c0 = cc * 32;
    const bool first = c0 < C0;
    n_Cs = first ? C0 : C1; n_cb = (first ? c0 : c0 - C0) + q8;
    if (ABL >= 4) { n_cb = (n_cb - q8) * H * W + q8; }      // timing only: channel-blocked source layout [C/32][H][W][32]
    n_simg = (first ? src0 : src1) + (size_t)i_img * H * W * n_Cs;
    if (ABL >= 4) n_Cs = 32;
    n_wbase = wgt + ((size_t)cc * Cout + i_g * 64) * 32;
    n_lbase = smem + buf * V4_BUF + wave * 1024;
  };
  auto piece = [&](int k) {      // k is a compile-time constant at every call site
    if (k < 5) {
      const __bf16* p = pixoff[k] >= 0 ? n_simg + (unsigned)(pixoff[k] * n_Cs + n_cb) : zero_page;
      __builtin_amdgcn_global_load_lds((v2_gptr)p, (v2_lptr)(n_lbase + k * 8192), 16, 0, 0);
    } else if (k == 5) {
      const __bf16* pi = pixoff[5] >= 0 ? n_simg + (unsigned)(pixoff[5] * n_Cs + n_cb) : zero_page;
      const __bf16* p = in5 ? pi : n_wbase + wrel5;
      __builtin_amdgcn_global_load_lds((v2_gptr)p, (v2_lptr)(n_lbase + 5 * 8192), 16, 0, 0);
    } else if (k < 9) {
      __builtin_amdgcn_global_load_lds((v2_gptr)(n_wbase + (wrel5 + (k - 5) * wstep)), (v2_lptr)(n_lbase + k * 8192), 16, 0, 0);
    } else {
      if (w9) __builtin_amdgcn_global_load_lds((v2_gptr)(n_wbase + (wrel5 + 4 * wstep)), (v2_lptr)(n_lbase + 9 * 8192), 16, 0, 0);
    }
  };

  item_lane_setup();
  describe(0, 0);
#pragma unroll
  for (int k = 0; k < 10; ++k) piece(k);
  int buf = 0;
  bool stores_in_flight = false;
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
      if (stores_in_flight) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");     // the 16 epilogue stores are younger than the DMA
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      stores_in_flight = false;
      __builtin_amdgcn_s_barrier();
      bool has_next = true;
      if (cc + 1 < nchunk) {
        describe(cc + 1, buf ^ 1);
      } else if (w + 1 < w_end) {
        if (++i_g == ngroups) { i_g = 0; if (++i_tx == tiles_x) { i_tx = 0; if (++i_ty == tiles_y) { i_ty = 0; ++i_img; } } }
        item_lane_setup();
        describe(0, buf ^ 1);
      } else has_next = false;
      if (ABL == 2) has_next = false;
      if (!SPREAD && has_next) {
#pragma unroll
        for (int k = 0; k < 10; ++k) piece(k);
      }
      const unsigned char* sb = smem + buf * V4_BUF;
      const unsigned char* xb[2][3];
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) xb[sp][dx] = sb + xoff[sp][dx];
      const unsigned char* wbp = sb + woff;
      auto rd = [&](int tap, bf16x8* xf, bf16x8* wf) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int m = 0; m < 4; ++m) { const int s = m + dy; xf[m] = *(const bf16x8*)(xb[s & 1][dx] + (s & ~1) * V4_PW * 64); }
#pragma unroll
        for (int n = 0; n < 4; ++n) wf[n] = *(const bf16x8*)(wbp + (tap * 64 + n * 16) * 64);
      };
      auto rd3 = [&](int tap, bf16x8* xf, bf16x8* wf) {     // ABL 3: registers only
#pragma unroll
        for (int m = 0; m < 4; ++m) { xf[m] = __builtin_bit_cast(bf16x8, (u32x4){(unsigned)tap, (unsigned)lane, 1u, 2u}); wf[m] = xf[m]; }
      };
      auto mm = [&](const bf16x8* xf, const bf16x8* wf) {
        if (ABL == 1 || ABL == 5) {
#pragma unroll
          for (int m = 0; m < 4; ++m) { asm volatile("" :: "v"(xf[m])); asm volatile("" :: "v"(wf[m])); }
          return;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xf[m], acc[m][n], 0, 0, 0);
      };
      if (PREF) {
        bf16x8 xa[4], wa[4], xb2[4], wb2[4];
        rd(0, xa, wa);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          bf16x8* xc = (tap & 1) ? xb2 : xa; bf16x8* wc = (tap & 1) ? wb2 : wa;
          bf16x8* xn = (tap & 1) ? xa : xb2; bf16x8* wn = (tap & 1) ? wa : wb2;
          if (tap + 1 < 9) rd(tap + 1, xn, wn);
          if (SPREAD && has_next) { piece(tap); if (tap == 8) piece(9); }
          if (PIN) __builtin_amdgcn_sched_barrier(0);
          mm(xc, wc);
          if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          bf16x8 xf[4], wf[4];
          if (ABL == 3) rd3(tap, xf, wf); else rd(tap, xf, wf);
          if (SPREAD && has_next && ABL != 7 && ABL != 8) { piece(tap); if (tap == 8) piece(9); }
          if (PIN) __builtin_amdgcn_sched_barrier(0);
          if (ABL == 6 || ABL == 8) __builtin_amdgcn_s_setprio(1);
          mm(xf, wf);
          if (ABL == 6 || ABL == 8) __builtin_amdgcn_s_setprio(0);
          if (SPREAD && has_next && (ABL == 7 || ABL == 8)) { piece(tap); if (tap == 8) piece(9); }
          if (PIN) __builtin_amdgcn_sched_barrier(0);
        }
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
    stores_in_flight = true;
  }
  if (blockIdx.x == 3 && tid == 0) {     // lab: core clock held during the launch (written behind the 64 zero bytes)
    long long* dbg = (long long*)(zero_page + 64);
    dbg[0] = clock64() - t_core0; dbg[1] = wall_clock64() - t_wall0;
  }
}

template <int SPREAD, int PREF, int PIN, int ABL = 0>
static void launch_conv_v4(const __bf16* src0, const __bf16* src1, int C0, int C1, const __bf16* wpk, const float* bias, __bf16* dst,
                           int H, int W, int Cout, int nimg, int relu, hipStream_t st) {
  const int total = nimg * (W / 32) * (H / 16) * (Cout / 64);
  int grid = total < 256 ? total : 256;
  hipLaunchKernelGGL((k_conv3_bf16_v4<SPREAD, PREF, PIN, ABL>), dim3(grid), dim3(V2_THREADS), 0, st, src0, src1, C0, C1, wpk, bias, dst, H, W, Cout, relu, nimg,
                     v2_zero_page());
}

}  // namespace sh
