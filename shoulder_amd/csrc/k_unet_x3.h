// k_unet_x3.h -- SH_UNET_F32X: the f32 network (f32 NHWC tensors in HBM, the first conv / pools / head on the f32 kernels of
// k_unet.h) with its MFMA layers on the 16-bit matrix pipe at f32-grade accuracy.
//
// The exact f32 path (k_conv_mfma_f32) runs `v_mfma_f32_16x16x4_f32` at 1/16 of the f16 rate: 1 040 humeri/s end to end,
// the only configuration that met the north star's 1e-4 mm.  Here every operand is split into two f16 values,
//     x = x_hi + x_lo,   x_hi = f16(x),  x_lo = f16(x - x_hi)            (22 significant bits)
// and a product becomes THREE `v_mfma_f32_16x16x32_f16` into one f32 accumulator:
//     w x  ~  w_hi x_hi + w_hi x_lo + w_lo x_hi                          (the dropped w_lo x_lo is 2^-22 of the product)
// -- three MFMAs at 16x the f32 rate.  Weights are split once per parameter block (k_pack_w_x3), scaled by 2^6 so that
// their low parts stay normal f16 numbers (the accumulator starts at 2^6 bias and the epilogue multiplies by 2^-6: exact);
// activations are split while a halo tile is staged (f32 from HBM -> two f16 images in LDS).  Not the f32 fma chain of
// oracle/unet_chain.c bit for bit -- the logits agree with it to ~1e-5 (tests/test_gpu_unet_x3.py) -- so the mask can differ
// from the exact path's in pixels whose logit is within that distance of zero; bench.py and the tests count them.
//
// Structure: the two-barrier kernel of k_unet_bf16.h (16x16-pixel tile, 16 NT couts per workgroup, 32-channel chunks, the
// global loads of chunk c+1 in flight during the MFMAs of chunk c), A = weights, B = pixels; 256 lanes = one wave per SIMD.
#pragma once
#include "k_unet_bf16.h"

namespace sh {

#define X3_WSCALE 64.0f

// all MFMA layers in one launch: packed[t.w_off + e] for e = ((tap * Cin/32 + chunk) * Cout + cout) * 32 + k, as k_pack_w16_all
__global__ void k_pack_w_x3(const float* __restrict__ P, u16* __restrict__ PH_, u16* __restrict__ PL_, const PackEntry* __restrict__ tab, int nlayers, long long total) {
  _Float16* PH = (_Float16*)PH_;
  _Float16* PL = (_Float16*)PL_;
  for (long long g = blockIdx.x * (long long)blockDim.x + threadIdx.x; g < total; g += (long long)gridDim.x * blockDim.x) {
    int l = 0;
    while (l + 1 < nlayers && tab[l + 1].first <= g) ++l;
    const PackEntry t = tab[l];
    const long long e = g - t.first;
    const int k = (int)(e % 32);
    long long r = e / 32;
    const int co = (int)(r % t.Cout);
    r /= t.Cout;
    const int cc = (int)(r % (t.Cin / 32));
    const int tp = (int)(r / (t.Cin / 32));
    const float w = P[t.w_off + ((long long)tp * t.Cin + cc * 32 + k) * t.Cout + co] * X3_WSCALE;
    const _Float16 hi = (_Float16)w;
    PH[t.w_off + e] = hi;
    PL[t.w_off + e] = (_Float16)(w - (float)hi);
  }
}

// TAPS = 9: 3x3 conv, pad 1.  TAPS = 1: one phase (blockIdx.z % 4 = dy * 2 + dx) of a 2x2 stride-2 transposed conv.
// Input channel c < C0 comes from src0, else from src1 (the decoder's cat([skip, up])); C0, C1 multiples of 32.
// FUSE: 0; UF_POOL -- the 2x2 max pool (input of the next encoder level) written beside the output; UF_HEAD (NT = 2, Cout = 32)
// -- the 1x1 head applied to the accumulators, only the logits leave the kernel (same operation order per logit as the
// UF_HEAD epilogues of the 16-bit kernels); UF_FIRST (Cin = 32 = the first conv's output) -- the layer's input is computed from
// the 1-channel image while the halo tile is staged: per halo pixel and channel the f32 fma chain of k_conv_first (bias, then taps
// 0..8, ReLU), bit for bit, so enc0a's 2.1 GB tensor is neither written nor read.
template <int TAPS, int NT, int FUSE = 0, int DB = 1>
__global__ void __launch_bounds__(UN_THREADS)
k_conv_mfma_x3(const float* __restrict__ src0, const float* __restrict__ src1, int C0, int C1,
               const u16* __restrict__ wh_, const u16* __restrict__ wl_ /*packed [phase][tap][Cin/32][Cout][32] f16: high / low part of 64 w*/,
               const float* __restrict__ bias, float* __restrict__ dst, int H, int W, int Cout, int relu,
               float* __restrict__ pooled /*UF_POOL: [nimg][H/2][W/2][Cout]*/, const float* __restrict__ head_w, const float* __restrict__ head_b,
               float* __restrict__ logits /*UF_HEAD: [nimg][H][W]*/,
               const float* __restrict__ image /*UF_FIRST: [nimg][H][W]*/, const float* __restrict__ w0 /*[9][32]*/, const float* __restrict__ b0 /*[32]*/) {
  using ET = _Float16;
  using v8 = typename E16<ET>::v8;
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  constexpr int PW = UN_TW + 2 * HALO, PH = UN_TH + 2 * HALO;
  constexpr int NC = 16 * NT;
  constexpr int IN_PIECES = PH * PW * 4, WT_PIECES = TAPS * NC * 4;
  constexpr int NIN = (IN_PIECES + UN_THREADS - 1) / UN_THREADS, NWT = (WT_PIECES + UN_THREADS - 1) / UN_THREADS;
  // the input images are double buffered (2 x 2 x 20.7 KB; with the 64-cout weight images 157 KB): chunk c + 1 is split and written
  // into the other pair in the middle of chunk c's taps -- its loads have landed by then -- so only the weights are stored
  // between the two barriers
  constexpr int XI = PH * PW * UB_PSTR;
  __shared__ __attribute__((aligned(16))) ET s_xh[(DB ? 2 : 1) * XI];      // DB = 0: one pair, the input stored between the barriers as well
  __shared__ __attribute__((aligned(16))) ET s_xl[(DB ? 2 : 1) * XI];
  __shared__ __attribute__((aligned(16))) ET s_wh[TAPS * NC * UB_PSTR];
  __shared__ __attribute__((aligned(16))) ET s_wl[TAPS * NC * UB_PSTR];
  __shared__ float s_img[(FUSE & UF_FIRST) ? (UN_TH + 4) * (UN_TW + 4) : 1];
  __shared__ float s_w0[(FUSE & UF_FIRST) ? 10 * 32 : 1];      // [9][32] weights, then the bias
  const ET* wh = (const ET*)wh_;
  const ET* wl = (const ET*)wl_;
  const int Cin = C0 + C1;
  const int tiles_x = W / UN_TW;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int n0 = blockIdx.y * NC;
  const int img = TAPS == 9 ? blockIdx.z : blockIdx.z / 4;
  const int phase = TAPS == 9 ? 0 : blockIdx.z % 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int x0 = tx * UN_TW, y0 = ty * UN_TH;
  const float* in0 = src0 + (size_t)img * H * W * C0;
  const float* in1 = src1 ? src1 + (size_t)img * H * W * C1 : nullptr;
  const int nchunk = Cin / 32;
  const size_t wph = (size_t)phase * TAPS * nchunk * Cout * 32;

  // ---- staging plan, fixed for the whole K loop: piece e = (pixel p, 8-channel slot q)
  int in_pix[NIN], in_lds[NIN], wt_off[NWT], wt_lds[NWT];
#pragma unroll
  for (int k = 0; k < NIN; ++k) {
    const int e = tid + k * UN_THREADS;
    const int q = e & 3, p = e >> 2;
    const int px = p % PW, py = p / PW;
    const int gx = x0 + px - HALO, gy = y0 + py - HALO;
    const bool ok = e < IN_PIECES && gx >= 0 && gx < W && gy >= 0 && gy < H;
    in_pix[k] = ok ? (gy * W + gx) : -1;
    in_lds[k] = e < IN_PIECES ? UB_OFF(p, q) : -1;
  }
#pragma unroll
  for (int k = 0; k < NWT; ++k) {
    const int e = tid + k * UN_THREADS;
    const int q = e & 3, r = e >> 2;
    const int tap = r / NC, j = r % NC;
    wt_off[k] = e < WT_PIECES ? ((tap * nchunk) * Cout + n0 + j) * 32 + q * 8 : -1;
    wt_lds[k] = e < WT_PIECES ? UB_OFF(r, q) : -1;
  }
  f32x4 rin[NIN][2];
  u32x4 rwh[NWT], rwl[NWT];
  auto load_chunk = [&](int cc) {
    const int c0 = cc * 32;
    const bool first = c0 < C0;
    const float* src = first ? in0 : in1;
    const int Cs = first ? C0 : C1, cb = first ? c0 : c0 - C0;
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      f32x4 a = (f32x4){0.f, 0.f, 0.f, 0.f}, b = a;
      if (!(FUSE & UF_FIRST) && in_pix[k] >= 0) {
        const float* s = src + (size_t)in_pix[k] * Cs + cb + ((tid + k * UN_THREADS) & 3) * 8;
        a = *(const f32x4*)s; b = *(const f32x4*)(s + 4);
      }
      rin[k][0] = a; rin[k][1] = b;
    }
#pragma unroll
    for (int k = 0; k < NWT; ++k)
      if (wt_off[k] >= 0) {
        const size_t o = wph + (size_t)wt_off[k] + (size_t)cc * Cout * 32;
        rwh[k] = *(const u32x4*)(wh + o);
        rwl[k] = *(const u32x4*)(wl + o);
      }
  };
  auto store_input = [&](int buf) {
#pragma unroll
    for (int k = 0; k < NIN; ++k)
      if (in_lds[k] >= 0) {
        v8 hi, lo;
        float pv[9];
        if constexpr ((FUSE & UF_FIRST) != 0) {      // the 3x3 image patch of this halo pixel (zero outside the image: enc0a's padding)
          const int e = tid + k * UN_THREADS, p = e >> 2;
          const int px = p % PW, py = p / PW;
#pragma unroll
          for (int t = 0; t < 9; ++t) pv[t] = s_img[(py + t / 3) * (UN_TW + 4) + px + t % 3];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = rin[k][j >> 2][j & 3];
          if constexpr ((FUSE & UF_FIRST) != 0) {
            const int c = ((tid + k * UN_THREADS) & 3) * 8 + j;
            float a = s_w0[9 * 32 + c];
#pragma unroll
            for (int t = 0; t < 9; ++t) a = __builtin_fmaf(pv[t], s_w0[t * 32 + c], a);
            v = in_pix[k] >= 0 ? fmaxf(a, 0.0f) : 0.0f;      // outside the image: enc0b's zero padding
          }
          const ET h = (ET)v;
          hi[j] = h;
          lo[j] = (ET)(v - (float)h);
        }
        *(v8*)(s_xh + buf * XI + in_lds[k]) = hi;
        *(v8*)(s_xl + buf * XI + in_lds[k]) = lo;
      }
  };
  auto store_weights = [&]() {
#pragma unroll
    for (int k = 0; k < NWT; ++k)
      if (wt_lds[k] >= 0) { *(u32x4*)(s_wh + wt_lds[k]) = rwh[k]; *(u32x4*)(s_wl + wt_lds[k]) = rwl[k]; }
  };

  f32x4 acc[4][NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[n0 + n * 16 + lk * 4 + r] * X3_WSCALE;
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m][n] = bv;
  }

  // One prefetch per chunk (input + weights of chunk c + 1 into registers), issued once the weights of chunk c are in LDS; its input
  // half is consumed at tap 4 of chunk c, its weight half between the barriers in front of chunk c + 1.
  if constexpr ((FUSE & UF_FIRST) != 0) {
    constexpr int IW = UN_TW + 4;
    const float* im = image + (size_t)img * H * W;
    for (int e = tid; e < (UN_TH + 4) * IW; e += UN_THREADS) {
      const int gy = y0 - 2 + e / IW, gx = x0 - 2 + e % IW;
      s_img[e] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? im[(size_t)gy * W + gx] : 0.0f;
    }
    for (int e = tid; e < 9 * 32; e += UN_THREADS) s_w0[e] = w0[e];
    if (tid < 32) s_w0[9 * 32 + tid] = b0[tid];
    __syncthreads();
  }
  load_chunk(0);
  store_input(0);
  for (int cc = 0; cc < nchunk; ++cc) {
    const int buf = DB ? (cc & 1) : 0;
    __syncthreads();                  // every wave is done reading the previous chunk's weights (and the input buffer `buf ^ 1`)
    if (!DB && cc > 0) store_input(0);
    store_weights();
    __syncthreads();
    if (cc + 1 < nchunk) load_chunk(cc + 1);      // in flight during the MFMAs below
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap % 3 : 0;
      v8 xh[4], xl[4];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int off = buf * XI + UB_OFF((wave * 4 + m + dy) * PW + li + dx, lk);
        xh[m] = *(const v8*)(s_xh + off); xl[m] = *(const v8*)(s_xl + off);
      }
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        const int off = UB_OFF(tap * NC + n * 16 + li, lk);
        const v8 fh = *(const v8*)(s_wh + off), fl = *(const v8*)(s_wl + off);
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          acc[m][n] = E16<ET>::mfma(fh, xh[m], acc[m][n]);
          acc[m][n] = E16<ET>::mfma(fh, xl[m], acc[m][n]);
          acc[m][n] = E16<ET>::mfma(fl, xh[m], acc[m][n]);
        }
      }
      if (DB && tap == (TAPS == 9 ? 4 : 0) && cc + 1 < nchunk) store_input(buf ^ 1);      // (the other pair: last read in chunk c - 1, a barrier ago)
    }
  }
  if constexpr ((FUSE & UF_HEAD) != 0) {
    // logits = head_b + sum_c head_w[c] * relu(conv[c]): this lane holds 4 NT of the Cout values of its pixels, the other
    // three quarters sit in lanes li + 16, + 32, + 48
    float hw[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) hw[n][r] = head_w[n0 + n * 16 + lk * 4 + r];
    const float hb = head_b[0];
    float* lg = logits + (size_t)img * H * W;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float p = 0.0f;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) p = __builtin_fmaf(fmaxf(acc[m][n][r] * (1.0f / X3_WSCALE), 0.0f), hw[n][r], p);
      p += __shfl_xor(p, 16);
      p += __shfl_xor(p, 32);
      if (lk == 0) lg[(size_t)(y0 + wave * 4 + m) * W + x0 + li] = hb + p;
    }
    return;
  }
  // ---- epilogue: f32 NHWC, this lane's 4 consecutive couts of pixel li of its 4 rows
  const int OW = TAPS == 9 ? W : 2 * W, OH = TAPS == 9 ? H : 2 * H;
  float* out = dst + (size_t)img * OH * OW * Cout;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    const int gy = y0 + wave * 4 + m, gx = x0 + li;
    const int oy = TAPS == 9 ? gy : 2 * gy + (phase >> 1), ox = TAPS == 9 ? gx : 2 * gx + (phase & 1);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      f32x4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[m][n][r] * (1.0f / X3_WSCALE);
        if (relu) v = fmaxf(v, 0.0f);
        o[r] = v;
      }
      *(f32x4*)(out + ((size_t)oy * OW + ox) * Cout + n0 + n * 16 + lk * 4) = o;
    }
  }
  if constexpr ((FUSE & UF_POOL) != 0) {
    // 2x2 max pool of this wave's 4 rows x 16 pixels: rows pair inside the lane, columns pair with lane li ^ 1
    float* po = pooled + (size_t)img * (H / 2) * (W / 2) * Cout;
#pragma unroll
    for (int mp = 0; mp < 2; ++mp)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaxf(acc[2 * mp][n][r], acc[2 * mp + 1][n][r]) * (1.0f / X3_WSCALE);
          v = fmaxf(v, __shfl_xor(v, 1));
          if (relu) v = fmaxf(v, 0.0f);
          o[r] = v;
        }
        if ((li & 1) == 0)
          *(f32x4*)(po + ((size_t)((y0 + wave * 4) / 2 + mp) * (W / 2) + (x0 + li) / 2) * Cout + n0 + n * 16 + lk * 4) = o;
      }
  }
}


// ---- 2x2 stride-2 transposed convolution of the F32X path with the source pixels held in registers ------------------------------
// k_conv_mfma_x3<1, NT> runs one (16x16 source tile, 16 NT couts, phase) per workgroup: the f32 tile is read 4 Cout / (16 NT) times
// and split again every time, a chunk is 12 NT MFMAs per wave between two barriers (up0..up3: 0.74 + 0.46 + 0.37 + 0.29 ms at
// B = 64 for 1.9 + 1.0 + 0.5 + 0.3 GB of tensors).  As in the 16-bit up-convolution (k_unet16_up.h) a workgroup of 8 waves owns a 32 x (4 MT)
// source tile for all 4 Cout outputs: a wave loads its MT rows x 16 pixels x Cin once, splits them into the high / low f16
// fragments (x_hi = f16(x), x_lo = f16(x - x_hi): the arithmetic of the staged form) and keeps both for the whole launch
// (2 x 4 MT NCH registers); the split weights stream through LDS, one (32-cout group, phase) slice at a time, fetched into registers
// while the previous slice is multiplied.  Per accumulator the same MFMAs in the same order (bias 2^6 in the accumulator, chunks
// ascending, w_hi x_hi, w_hi x_lo, w_lo x_hi): bit-identical outputs (tests/test_gpu_unet_x3.py).
#define UXR_THREADS 512

template <int NCH, int MT>
__global__ void __launch_bounds__(UXR_THREADS)
k_upconv_x3r(const float* __restrict__ src /*[img][H W][32 NCH]*/, const u16* __restrict__ wh_, const u16* __restrict__ wl_ /*packed [4][1][NCH][Cout][32]*/,
             const float* __restrict__ bias, float* __restrict__ dst /*[img][2H 2W][Cout]*/, int H, int W, int Cout) {
  using ET = _Float16;
  using v8 = typename E16<ET>::v8;
  const ET* wh = (const ET*)wh_;
  const ET* wl = (const ET*)wl_;
  constexpr int Cin = 32 * NCH;
  constexpr int WROWS = NCH * 32;                                         // LDS rows of a weight slice: [chunk][16 n + i]
  constexpr int WPASS = (WROWS * 4 + UXR_THREADS - 1) / UXR_THREADS;      // 16-byte pieces per thread, slice and half
  __shared__ __attribute__((aligned(16))) ET s_wh[2][WROWS * UB_PSTR];
  __shared__ __attribute__((aligned(16))) ET s_wl[2][WROWS * UB_PSTR];
  const int tiles_x = W / 32;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, img = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int xh_ = wave & 1, rg = wave >> 1;
  const int x0 = tx * 32 + xh_ * 16 + li, y0 = (ty * 4 + rg) * MT;
  const float* in = src + (size_t)img * H * W * Cin;
  const int groups = Cout >> 5, nslice = groups * 4;

  int wt_src[WPASS], wt_lds[WPASS];
#pragma unroll
  for (int k = 0; k < WPASS; ++k) {
    const int e = tid + k * UXR_THREADS, q = e & 3, r = e >> 2, cc = r >> 5, j = r & 31;
    wt_src[k] = e < WROWS * 4 ? (cc * Cout + j) * 32 + q * 8 : -1;
    wt_lds[k] = UB_OFF(r, q);
  }
  u32x4 rwh[WPASS], rwl[WPASS];
  auto load_slice = [&](int t) {      // slice t = (group t >> 2, phase t & 3)
    const size_t o = ((size_t)(t & 3) * NCH * Cout + (size_t)(t >> 2) * 32) * 32;
#pragma unroll
    for (int k = 0; k < WPASS; ++k)
      if (wt_src[k] >= 0) { rwh[k] = *(const u32x4*)(wh + o + wt_src[k]); rwl[k] = *(const u32x4*)(wl + o + wt_src[k]); }
  };
  auto put_slice = [&](int b) {
#pragma unroll
    for (int k = 0; k < WPASS; ++k)
      if (wt_src[k] >= 0) { *(u32x4*)(s_wh[b] + wt_lds[k]) = rwh[k]; *(u32x4*)(s_wl[b] + wt_lds[k]) = rwl[k]; }
  };
  load_slice(0);
  v8 xh[MT][NCH], xl[MT][NCH];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
      const float* s = in + ((size_t)(y0 + m) * W + x0) * Cin + cc * 32 + lk * 8;
      const f32x4 a = *(const f32x4*)s, b = *(const f32x4*)(s + 4);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = j < 4 ? a[j & 3] : b[j & 3];
        const ET h = (ET)v;
        xh[m][cc][j] = h;
        xl[m][cc][j] = (ET)(v - (float)h);
      }
    }
  put_slice(0);
  __syncthreads();

  const int OW = 2 * W, OH = 2 * H;
  float* out = dst + (size_t)img * OH * OW * Cout;
  for (int t = 0; t < nslice; ++t) {
    const int g = t >> 2, dy = (t >> 1) & 1, dx = t & 1;
    if (t + 1 < nslice) load_slice(t + 1);      // in flight during the MFMAs below
    f32x4 acc[MT][2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      f32x4 bv;
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = bias[g * 32 + n * 16 + lk * 4 + r] * X3_WSCALE;
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][n] = bv;
    }
    const ET* swh = s_wh[t & 1];
    const ET* swl = s_wl[t & 1];
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc)
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const int off = UB_OFF(cc * 32 + n * 16 + li, lk);
        const v8 fh = *(const v8*)(swh + off), fl = *(const v8*)(swl + off);
#pragma unroll
        for (int m = 0; m < MT; ++m) {
          acc[m][n] = E16<ET>::mfma(fh, xh[m][cc], acc[m][n]);
          acc[m][n] = E16<ET>::mfma(fh, xl[m][cc], acc[m][n]);
          acc[m][n] = E16<ET>::mfma(fl, xh[m][cc], acc[m][n]);
        }
      }
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const size_t opix = (size_t)(2 * (y0 + m) + dy) * OW + 2 * x0 + dx;
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = acc[m][n][r] * (1.0f / X3_WSCALE);
        *(f32x4*)(out + opix * Cout + g * 32 + n * 16 + lk * 4) = o;
      }
    }
    if (t + 1 < nslice) put_slice((t + 1) & 1);      // its last readers passed the barrier that ended slice t - 1
    __syncthreads();
  }
}

}  // namespace sh
