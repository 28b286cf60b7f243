// k_unet.h -- UNet forward on gfx950 (K19; stands in for the reference's missing unetcrf_anp.onnx,
// anatomic_neck.py:62-76).  Activations NHWC float32 in HBM; conv weights [tap][cin][cout].
//
// f32 path (parity): every output element is ONE float32 fma chain that starts from the bias and
// runs over (16-channel chunk, tap, cin within chunk) in order -- v_mfma_f32_16x16x4_f32 accumulates its 4 k-values as a k-ordered
// fmaf chain, so this K loop with a single accumulator reproduces
// oracle/unet_chain.c bit for bit.  No split-K, no re-association.
//
// conv3x3 / upconv kernel (implicit GEMM, M = 16 pixels of one image row, N = 16 couts, K = 4 cin):
//   workgroup = 256 lanes = 4 waves, output tile 16x16 pixels x (16*NT) couts;
//   wave w owns rows 4w..4w+3 (4 M-tiles) x NT N-tiles -> 4*NT accumulator tiles;
//   per 16-channel chunk: the 18x18 halo tile is staged global->LDS channel-major
//   (plane stride 336 floats: lanes (i,k) of one A read hit 32 distinct banks), the weights
//   [tap][16][16*NT] with k-stride padded by 16 floats likewise; then 9 taps x 4 k-steps of
//   (4 A + NT B ds_read_b32, 4*NT MFMA).
// Roofline: MFMA-bound at f32 (157 TFLOP/s dense peak); bytes are small (activations of a
// 64-image batch stay in the 256 MB Infinity Cache between layers).
#pragma once
#include "k_anp.h"
#include "k_unet16_base.h"

namespace sh {


#define UN_TW 16
#define UN_TH 16
#define UN_CK 16
#define UN_PLANE 336            // (18*18 = 324) padded: 336 % 32 == 16
#define UN_THREADS 256

// TAPS = 9: 3x3 conv, pad 1.  TAPS = 1: one phase (blockIdx.z % 4 = dy*2+dx) of a 2x2 stride-2
// transposed conv: out[2y+dy][2x+dx] = b + sum_ci in[y][x][ci] * w[phase][ci][co].
// Input channel c < C0 comes from src0, else from src1 (skip concat by pointer, anatomic-neck UNet
// decoder: cat([skip, up])).
template <int TAPS, int NT>
__global__ void __launch_bounds__(UN_THREADS)
k_conv_mfma_f32(const float* __restrict__ src0, const float* __restrict__ src1, int C0, int C1,
                const float* __restrict__ wgt /*[TAPS*(upconv?4:1)][Cin][Cout]*/, const float* __restrict__ bias,
                float* __restrict__ dst, int H, int W, int Cout, int relu) {
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  constexpr int PW = UN_TW + 2 * HALO, PH = UN_TH + 2 * HALO;
  constexpr int WK = 16 * NT + 16;                 // padded k-stride of the weight tile
  __shared__ float s_in[UN_CK * UN_PLANE];
  __shared__ float s_w[TAPS * UN_CK * WK];
  const int Cin = C0 + C1;
  const int tiles_x = W / UN_TW;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int n0 = blockIdx.y * 16 * NT;
  const int img = TAPS == 9 ? blockIdx.z : blockIdx.z / 4;
  const int phase = TAPS == 9 ? 0 : blockIdx.z % 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int x0 = tx * UN_TW, y0 = ty * UN_TH;
  const float* in0 = src0 + (size_t)img * H * W * C0;
  const float* in1 = src1 ? src1 + (size_t)img * H * W * C1 : nullptr;
  const float* wp = wgt + (size_t)phase * Cin * Cout;

  f32x4 acc[4][NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    float bv = bias[n0 + n * 16 + li];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m][n] = (f32x4){bv, bv, bv, bv};
  }

  for (int c0 = 0; c0 < Cin; c0 += UN_CK) {
    __syncthreads();
    // ---- stage input halo tile, channel-major: s_in[k][py*PW + px]
    for (int e = tid; e < PH * PW * (UN_CK / 4); e += UN_THREADS) {
      int q = e % (UN_CK / 4), p = e / (UN_CK / 4);
      int px = p % PW, py = p / PW;
      int gx = x0 + px - HALO, gy = y0 + py - HALO;
      int c = c0 + q * 4;
      f32x4 v = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (gx >= 0 && gx < W && gy >= 0 && gy < H) {
        const float* s = (c < C0) ? in0 + ((size_t)gy * W + gx) * C0 + c : in1 + ((size_t)gy * W + gx) * C1 + (c - C0);
        v = *(const f32x4*)s;
      }
      s_in[(q * 4 + 0) * UN_PLANE + p] = v.x;
      s_in[(q * 4 + 1) * UN_PLANE + p] = v.y;
      s_in[(q * 4 + 2) * UN_PLANE + p] = v.z;
      s_in[(q * 4 + 3) * UN_PLANE + p] = v.w;
    }
    // ---- stage weights: s_w[(tap*16 + k)*WK + j]  for j < 16*NT
    for (int e = tid; e < TAPS * UN_CK * (4 * NT); e += UN_THREADS) {
      int j4 = e % (4 * NT), r = e / (4 * NT);          // r = tap*16 + k
      int tap = r / UN_CK, k = r % UN_CK;
      f32x4 v = *(const f32x4*)(wp + ((size_t)tap * Cin + c0 + k) * Cout + n0 + j4 * 4);
      *(f32x4*)(s_w + r * WK + j4 * 4) = v;
    }
    __syncthreads();
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap % 3 : 0;
#pragma unroll
      for (int ks = 0; ks < UN_CK / 4; ++ks) {
        const int k = ks * 4 + lk;
        float a[4], bq[NT];
#pragma unroll
        for (int m = 0; m < 4; ++m) a[m] = s_in[k * UN_PLANE + (wave * 4 + m + dy) * PW + li + dx];
#pragma unroll
        for (int n = 0; n < NT; ++n) bq[n] = s_w[(tap * UN_CK + k) * WK + n * 16 + li];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[m], bq[n], acc[m][n], 0, 0, 0);
      }
    }
  }
  // ---- epilogue: D[row = 4*(lane>>4)+reg (pixel)][col = lane&15 (cout)]
  const int OW = TAPS == 9 ? W : 2 * W, OH = TAPS == 9 ? H : 2 * H;
  float* out = dst + (size_t)img * OH * OW * Cout;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    int gy = y0 + wave * 4 + m;
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int gx = x0 + lk * 4 + r;
        float v = acc[m][n][r];
        if (relu) v = fmaxf(v, 0.0f);
        int oy = TAPS == 9 ? gy : 2 * gy + (phase >> 1), ox = TAPS == 9 ? gx : 2 * gx + (phase & 1);
        out[((size_t)oy * OW + ox) * Cout + n0 + n * 16 + li] = v;
      }
  }
}

// First layer: Cin = 1 (the radius image), Cout = C (<= 64): 9-tap fma chain per output, VALU.
__global__ void k_conv_first(const float* __restrict__ img, const float* __restrict__ wgt /*[9][1][C]*/, const float* __restrict__ bias,
                             float* __restrict__ dst, int H, int W, int C, int nimg) {
  __shared__ float sw[9 * SH_UNET_MAXBASE + SH_UNET_MAXBASE];      // [9][C] weights, then the bias (C <= SH_UNET_MAXBASE, checked by sh_load_unet)
  for (int e = threadIdx.x; e < 9 * C; e += blockDim.x) sw[e] = wgt[e];
  for (int e = threadIdx.x; e < C; e += blockDim.x) sw[9 * SH_UNET_MAXBASE + e] = bias[e];
  __syncthreads();
  size_t total = (size_t)nimg * H * W;
  for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < total; p += (size_t)gridDim.x * blockDim.x) {
    int x = (int)(p % W), y = (int)((p / W) % H);
    size_t im = p / ((size_t)H * W);
    const float* src = img + im * H * W;
    float v[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      int gy = y + t / 3 - 1, gx = x + t % 3 - 1;
      v[t] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? src[(size_t)gy * W + gx] : 0.0f;
    }
    float* o = dst + p * C;
    for (int c = 0; c < C; ++c) {
      float a = sw[9 * SH_UNET_MAXBASE + c];
#pragma unroll
      for (int t = 0; t < 9; ++t) a = __builtin_fmaf(v[t], sw[t * C + c], a);
      o[c] = fmaxf(a, 0.0f);
    }
  }
}

// 2x2 max-pool, NHWC, 4 channels per lane
__global__ void k_maxpool2(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int C, int nimg) {
  const int OH = H / 2, OW = W / 2, C4 = C / 4;
  size_t total = (size_t)nimg * OH * OW * C4;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    int c4 = (int)(e % C4);
    size_t p = e / C4;
    int ox = (int)(p % OW), oy = (int)((p / OW) % OH);
    size_t im = p / ((size_t)OH * OW);
    const float* s = src + ((im * H + 2 * oy) * W + 2 * ox) * C + c4 * 4;
    f32x4 a = *(const f32x4*)s, b = *(const f32x4*)(s + C), c = *(const f32x4*)(s + (size_t)W * C), d = *(const f32x4*)(s + (size_t)W * C + C);
    f32x4 r;
    r.x = fmaxf(fmaxf(a.x, b.x), fmaxf(c.x, d.x));
    r.y = fmaxf(fmaxf(a.y, b.y), fmaxf(c.y, d.y));
    r.z = fmaxf(fmaxf(a.z, b.z), fmaxf(c.z, d.z));
    r.w = fmaxf(fmaxf(a.w, b.w), fmaxf(c.w, d.w));
    *(f32x4*)(dst + ((im * OH + oy) * OW + ox) * C + c4 * 4) = r;
  }
}

// 1x1 head: logit = fma chain over channels from the bias (anatomic_neck.py:76 output)
// 256 pixels per workgroup pass: coalesced 16-byte loads into LDS (row stride C+1 floats, so the
// per-pixel channel walk below is bank-conflict free), then one sequential chain per lane.
// MAXC: 32 (the network's base width: a 33.8 KB tile, four workgroups per CU; with the 66.5 KB tile of the general form two fit and the
// layer -- 2.1 GB in at B = 64 -- ran at 0.36 of the HBM roof) or 64.
template <int MAXC>
__global__ void __launch_bounds__(256)
k_head(const float* __restrict__ src, const float* __restrict__ w, const float* __restrict__ bp, float* __restrict__ logits, int C, size_t npix) {
  constexpr int TILE = 256 * (MAXC + 1);       // floats: 256 pixels per pass up to MAXC channels, fewer pixels per pass above
  __shared__ float tile[TILE];
  __shared__ float sw[SH_UNET_MAXBASE];
  const int tid = threadIdx.x;
  const float b = bp[0];
  for (int i = tid; i < C; i += 256) sw[i] = w[i];
  const int C4 = C / 4, ld = C + 1;
  const int ppx = min(256, TILE / ld);         // pixels per pass
  for (size_t p0 = (size_t)blockIdx.x * ppx; p0 < npix; p0 += (size_t)gridDim.x * ppx) {
    __syncthreads();
    size_t np_ = npix - p0 < (size_t)ppx ? npix - p0 : (size_t)ppx;
    for (int e = tid; e < (int)np_ * C4; e += 256) {
      int px = e / C4, c4 = e % C4;
      f32x4 v = *(const f32x4*)(src + (p0 + px) * C + c4 * 4);
      float* t = tile + px * ld + c4 * 4;
      t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
    }
    __syncthreads();
    if ((size_t)tid < np_) {
      const float* s = tile + tid * ld;
      float a = b;
      for (int c = 0; c < C; ++c) a = __builtin_fmaf(s[c], sw[c], a);
      logits[p0 + tid] = a;
    }
  }
}

}  // namespace sh
