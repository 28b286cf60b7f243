// k_conv_bf16_v3.h -- v2 (persistent workgroups, LDS-DMA double buffering) plus schedule options (lab version):
//   SPREAD: the ten LDS-DMA pieces of the next step are issued one per tap, between the MFMA groups
//   PREF:   the fragments of tap t+1 are read before the MFMAs of tap t
//   PIN:    sched_barrier(0) around each MFMA group
// and a counted s_waitcnt that leaves the epilogue stores of the previous item in flight.
#pragma once
#include "k_conv_bf16_v2.h"

namespace sh {

template <int SPREAD, int PREF, int PIN>
__global__ void __launch_bounds__(V2_THREADS)
k_conv3_bf16_v3(const __bf16* __restrict__ src0, const __bf16* __restrict__ src1, int C0, int C1,
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
  __syncthreads();

  // staging plan: slot e_k = tid + 512 k -> row r_k = (tid >> 2) + 128 k, 16-B slot tid & 3.  The swizzle bit (bit 2 of the
  // row) is the same for every k, weight rows advance by two taps per k.
  const int r0 = tid >> 2;
  const int q8 = ((tid & 3) ^ ((r0 >> 1) & 2)) * 8;
  const int rw4 = r0 + 512 - V2_INROWS;
  const int wstep = 2 * nchunk * Cout * 32;
  const int wrel4 = ((rw4 >> 6) * nchunk * Cout + (rw4 & 63)) * 32 + q8;
  const bool in4 = rw4 < 0;
  const bool w9 = tid + 512 * 9 < V2_SLOTS;

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
      int r = r0 + 128 * k;
      int py = r / V2_PW, px = r - py * V2_PW;
      int gx = i_tx * 32 + px - 1, gy = i_ty * 16 + py - 1;
      bool ok = gx >= 0 && gx < W && gy >= 0 && gy < H;
      pixoff[k] = ok ? gy * W + gx : -1;
    }
  };
  const __bf16* n_simg; const __bf16* n_wbase; int n_Cs, n_cb; unsigned char* n_lbase;
  auto describe = [&](int cc, int buf) {
    const int c0 = cc * 32;
    const bool first = c0 < C0;
    n_Cs = first ? C0 : C1; n_cb = (first ? c0 : c0 - C0) + q8;
    n_simg = (first ? src0 : src1) + (size_t)i_img * H * W * n_Cs;
    n_wbase = wgt + ((size_t)cc * Cout + i_g * 64) * 32;
    n_lbase = smem + buf * V2_BUF + wave * 1024;
  };
  auto piece = [&](int k) {      // k is a compile-time constant at every call site
    if (k < 4) {
      const __bf16* p = pixoff[k] >= 0 ? n_simg + (unsigned)(pixoff[k] * n_Cs + n_cb) : zero_page;
      __builtin_amdgcn_global_load_lds((v2_gptr)p, (v2_lptr)(n_lbase + k * 8192), 16, 0, 0);
    } else if (k == 4) {
      const __bf16* pi = pixoff[4] >= 0 ? n_simg + (unsigned)(pixoff[4] * n_Cs + n_cb) : zero_page;
      const __bf16* p = in4 ? pi : n_wbase + wrel4;
      __builtin_amdgcn_global_load_lds((v2_gptr)p, (v2_lptr)(n_lbase + 4 * 8192), 16, 0, 0);
    } else if (k < 9) {
      __builtin_amdgcn_global_load_lds((v2_gptr)(n_wbase + (wrel4 + (k - 4) * wstep)), (v2_lptr)(n_lbase + k * 8192), 16, 0, 0);
    } else {
      if (w9) __builtin_amdgcn_global_load_lds((v2_gptr)(n_wbase + (wrel4 + 5 * wstep)), (v2_lptr)(n_lbase + 9 * 8192), 16, 0, 0);
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
      if (!SPREAD && has_next) {
#pragma unroll
        for (int k = 0; k < 10; ++k) piece(k);
      }
      const __bf16* sb = (const __bf16*)(smem + buf * V2_BUF);
      auto rd = [&](int tap, bf16x8* xf, bf16x8* wf) {
        const int dy = tap / 3, dx = tap % 3;
#pragma unroll
        for (int m = 0; m < 4; ++m) { const int row = (rg * 4 + m + dy) * V2_PW + xh * 16 + li + dx; xf[m] = *(const bf16x8*)(sb + UB_OFF(row, lk)); }
#pragma unroll
        for (int n = 0; n < 4; ++n) { const int row = V2_INROWS + tap * 64 + n * 16 + li; wf[n] = *(const bf16x8*)(sb + UB_OFF(row, lk)); }
      };
      auto mm = [&](const bf16x8* xf, const bf16x8* wf) {
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf[n], xf[m], acc[m][n], 0, 0, 0);
      };
      if (PREF) {
        bf16x8 xa[4], wa[4], xb[4], wb[4];
        rd(0, xa, wa);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          bf16x8* xc = (tap & 1) ? xb : xa; bf16x8* wc = (tap & 1) ? wb : wa;
          bf16x8* xn = (tap & 1) ? xa : xb; bf16x8* wn = (tap & 1) ? wa : wb;
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
          rd(tap, xf, wf);
          if (SPREAD && has_next) { piece(tap); if (tap == 8) piece(9); }
          if (PIN) __builtin_amdgcn_sched_barrier(0);
          mm(xf, wf);
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
}

template <int SPREAD, int PREF, int PIN>
static void launch_conv_v3(const __bf16* src0, const __bf16* src1, int C0, int C1, const __bf16* wpk, const float* bias, __bf16* dst,
                           int H, int W, int Cout, int nimg, int relu, hipStream_t st) {
  const int total = nimg * (W / 32) * (H / 16) * (Cout / 64);
  int grid = total < 256 ? total : 256;
  hipLaunchKernelGGL((k_conv3_bf16_v3<SPREAD, PREF, PIN>), dim3(grid), dim3(V2_THREADS), 0, st, src0, src1, C0, C1, wpk, bias, dst, H, W, Cout, relu, nimg,
                     v2_zero_page());
}

}  // namespace sh
