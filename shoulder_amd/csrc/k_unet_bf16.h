// k_unet_bf16.h -- UNet forward with 16-bit activations and weights on v_mfma_f32_16x16x32_{bf16,f16} (throughput paths:
// BASELINE config 3 "UNet bf16", config 5 "fp16 MFMA conv"); f32 accumulate, bias f32.  Every kernel is a template on
// the element type ET (__bf16 or _Float16): same tiling, same LDS image, same instruction count -- the two MFMA forms have
// the same shape and rate -- so "bf16" in the comments below reads "ET".
//
// Transposed implicit GEMM: A = weights (rows = 16 couts), B = activations (cols = 16 pixels of one
// image row), K = 32 input channels of one tap.  D then has the pixel on the lane and 4 consecutive
// couts in the 4 accumulator registers, so the epilogue stores 8 contiguous bytes per lane
// (4 bf16) instead of 2-byte scattered stores.
//   lane l: A frag = W[cout l&15][k = 8(l>>4) .. +7]   B frag = X[k = 8(l>>4) .. +7][pixel l&15]
//   D[row = 4(l>>4)+r (cout)][col = l&15 (pixel)]
// LDS: pixel-major halo tile and cout-major weight tile, 32 channels (64 B) per row, unpadded, with an XOR
// swizzle of the 16-B slot (UB_OFF): every fragment is one aligned, bank-conflict-free ds_read_b128.
// Software pipeline over the 32-channel chunks: the global loads of chunk c+1 (16 B per lane, all
// offsets precomputed once per workgroup) are issued into registers before the MFMAs of chunk c and
// written to LDS after them, so HBM/L2 latency hides under the matrix work.
// Weights are re-packed once per forward to [phase][tap][Cin/32][Cout][32] bf16 (k_pack_w_bf16) so
// the staging loads are 16-byte and coalesced.
#pragma once
#include "k_unet.h"

namespace sh {

// Fusions at the memory-bound ends of the network (the 32-channel level moves 1.07 GB per tensor at B = 64):
//   UF_FIRST  the layer's input is the 1-channel image: the 32-channel activations of the first conv (enc0a) are
//             computed into the halo tile on the matrix cores instead of being read from HBM -- one MFMA per 16
//             halo pixels x 16 channels with K = 9 taps of the image's bf16 high part + 9 taps of its low part
//             (image precision ~2^-17) against the bf16-rounded first-layer weights
//   UF_HEAD   the 1x1 head is applied to the f32 accumulators in the epilogue; only the logits are written
//   UF_POOL   the epilogue also writes the 2x2 max-pooled tensor (input of the next encoder level)
enum { UF_FIRST = 1, UF_HEAD = 2, UF_POOL = 4 };
struct ConvFuse {
  const float* image;      // UF_FIRST: [nimg][H][W] f32
  const float* w0;         // UF_FIRST: first conv weights [9][32] f32
  const float* b0;         // UF_FIRST: first conv bias [32]
  const float* head_w;     // UF_HEAD:  [Cout] f32
  const float* head_b;     // UF_HEAD:  [1]
  float* logits;           // UF_HEAD:  [nimg][H][W] f32
  void* pooled;            // UF_POOL:  [nimg][H/2][W/2][Cout] ET
};

// All MFMA layers in one launch (the forward re-packs every time -- the parameter block may have been re-broadcast -- and
// 21 separate 5-us launches cost more in launch gaps than in work).  tab[l] = {first element, w_off, T, Cin, Cout}.
struct PackEntry { long long first; long long w_off; int T, Cin, Cout, pad; };
template <int EK>
__global__ void k_pack_w16_all(const float* __restrict__ P, u16* __restrict__ PW_, const PackEntry* __restrict__ tab, int nlayers, long long total) {
  using ET = typename EKT<EK>::type;
  ET* PW = (ET*)PW_;

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
    PW[t.w_off + e] = (ET)P[t.w_off + ((long long)tp * t.Cin + cc * 32 + k) * t.Cout + co];
  }
}

template <int EK, int TAPS, int NT, int FUSE = 0>
__global__ void __launch_bounds__(UN_THREADS)
k_conv_mfma16(const u16* __restrict__ src0_, const u16* __restrict__ src1_, int C0, int C1,
              const u16* __restrict__ wgt_, const float* __restrict__ bias, u16* __restrict__ dst_, int H, int W, int Cout, int relu,
              ConvFuse fz) {
  using ET = typename EKT<EK>::type;
  const ET* src0 = (const ET*)src0_;
  const ET* src1 = (const ET*)src1_;
  const ET* wgt = (const ET*)wgt_;
  ET* dst = (ET*)dst_;

  using v8 = typename E16<ET>::v8;
  using v4 = typename E16<ET>::v4;
  constexpr int HALO = TAPS == 9 ? 1 : 0;
  constexpr int PW = UN_TW + 2 * HALO, PH = UN_TH + 2 * HALO;
  constexpr int NC = 16 * NT;
  constexpr int IN_PIECES = PH * PW * 4, WT_PIECES = TAPS * NC * 4;
  constexpr int NIN = (IN_PIECES + UN_THREADS - 1) / UN_THREADS, NWT = (WT_PIECES + UN_THREADS - 1) / UN_THREADS;
  __shared__ __attribute__((aligned(16))) ET s_in[PH * PW * UB_PSTR];
  __shared__ __attribute__((aligned(16))) ET s_w[TAPS * NC * UB_PSTR];
  __shared__ float s_img[(FUSE & UF_FIRST) ? (UN_TH + 4) * (UN_TW + 4) : 1];
  const int Cin = C0 + C1;
  const int tiles_x = W / UN_TW;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int n0 = blockIdx.y * NC;
  const int img = TAPS == 9 ? blockIdx.z : blockIdx.z / 4;
  const int phase = TAPS == 9 ? 0 : blockIdx.z % 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int x0 = tx * UN_TW, y0 = ty * UN_TH;
  const ET* in0 = src0 + (size_t)img * H * W * C0;
  const ET* in1 = src1 ? src1 + (size_t)img * H * W * C1 : nullptr;
  const int nchunk = Cin / 32;
  const ET* wp = wgt + (size_t)phase * TAPS * nchunk * Cout * 32;

  // ---- staging plan, fixed for the whole K loop
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
    const bool first = c0 < C0;
    const ET* src = first ? in0 : in1;
    const int cb = first ? c0 : c0 - C0;
#pragma unroll
    for (int k = 0; k < NIN; ++k) {
      u32x4 v = (u32x4){0u, 0u, 0u, 0u};
      if (!(FUSE & UF_FIRST) && in_pix[k] >= 0) v = *(const u32x4*)(src + act_off((size_t)H * W, (size_t)in_pix[k], cb) + ((tid + k * UN_THREADS) & 3) * 8);
      rin[k] = v;
    }
#pragma unroll
    for (int k = 0; k < NWT; ++k)
      if (wt_off[k] >= 0) rwt[k] = *(const u32x4*)(wp + (size_t)wt_off[k] + (size_t)cc * Cout * 32);
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int k = 0; k < NIN; ++k) if (!(FUSE & UF_FIRST) && in_lds[k] >= 0) *(u32x4*)(s_in + in_lds[k]) = rin[k];
#pragma unroll
    for (int k = 0; k < NWT; ++k) if (wt_lds[k] >= 0) *(u32x4*)(s_w + wt_lds[k]) = rwt[k];
  };
  if (FUSE & UF_FIRST) {
    // enc0a on the halo tile.  k = 0..8: tap k of the image's bf16 high part, k = 9..17: tap k - 9 of its low part
    // (v - hi), k >= 18: zero; the weight fragment repeats w[tap] for both parts.
    constexpr int IW = UN_TW + 4;
    const float* im = fz.image + (size_t)img * H * W;
    for (int e = tid; e < (UN_TH + 4) * IW; e += UN_THREADS) {
      const int gy = y0 - 2 + e / IW, gx = x0 - 2 + e % IW;
      s_img[e] = (gy >= 0 && gy < H && gx >= 0 && gx < W) ? im[(size_t)gy * W + gx] : 0.0f;
    }
    v8 wA[2];
    f32x4 b0v[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = lk * 8 + j, t = k < 9 ? k : (k < 18 ? k - 9 : 0);
        const float wv = fz.w0[t * 32 + n * 16 + li];
        wA[n][j] = k < 18 ? (ET)wv : (ET)0.0f;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) b0v[n][r] = fz.b0[n * 16 + lk * 4 + r];
    }
    __syncthreads();
    for (int g = wave; g < (PH * PW + 15) / 16; g += UN_THREADS / 64) {
      const int p = 16 * g + li, pc = p < PH * PW ? p : PH * PW - 1;
      const int py = pc / PW, px = pc - py * PW;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
      const int base = py * IW + px;          // patch index of tap (0,0) of this pixel
      v8 bf;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = lk * 8 + j, t = k < 9 ? k : (k < 18 ? k - 9 : 0);
        const int t3 = (t * 11) >> 5;         // t / 3 for t <= 8
        const float v = s_img[base + t + (IW - 3) * t3];
        const ET hi = (ET)v;
        const ET lo = (ET)(v - (float)hi);
        bf[j] = k < 9 ? hi : (k < 18 ? lo : (ET)0.0f);
      }
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const f32x4 a = E16<ET>::mfma(wA[n], bf, b0v[n]);
        v4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) o[r] = inside ? (ET)fmaxf(a[r], 0.0f) : (ET)0.0f;      // outside the image: enc0b's zero padding
        if (p < PH * PW) *(v4*)(s_in + UB_OFF(p, n * 2 + (lk >> 1)) + (lk & 1) * 4) = o;
      }
    }
  }

  f32x4 acc[4][NT];
#pragma unroll
  for (int n = 0; n < NT; ++n) {
    f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[n0 + n * 16 + lk * 4 + r];
#pragma unroll
    for (int m = 0; m < 4; ++m) acc[m][n] = bv;
  }

  load_chunk(0);
  for (int cc = 0; cc < nchunk; ++cc) {
    __syncthreads();                  // every wave is done reading the previous chunk
    store_chunk();
    __syncthreads();
    if (cc + 1 < nchunk) load_chunk(cc + 1);      // in flight during the MFMAs below
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int dy = TAPS == 9 ? tap / 3 : 0, dx = TAPS == 9 ? tap % 3 : 0;
      v8 xf[4], wf[NT];
#pragma unroll
      for (int m = 0; m < 4; ++m) { const int row = (wave * 4 + m + dy) * PW + li + dx; xf[m] = *(const v8*)(s_in + UB_OFF(row, lk)); }
#pragma unroll
      for (int n = 0; n < NT; ++n) { const int row = tap * NC + n * 16 + li; wf[n] = *(const v8*)(s_w + UB_OFF(row, lk)); }
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = E16<ET>::mfma(wf[n], xf[m], acc[m][n]);
    }
  }
  const int OW = TAPS == 9 ? W : 2 * W, OH = TAPS == 9 ? H : 2 * H;
  if (FUSE & UF_HEAD) {
    // logits = head_b + sum_c head_w[c] * relu(acc[c]): this lane holds 4 NT of the Cout values of its pixels, the other
    // three quarters sit in lanes li + 16, + 32, + 48
    float hw[NT][4];
#pragma unroll
    for (int n = 0; n < NT; ++n)
#pragma unroll
      for (int r = 0; r < 4; ++r) hw[n][r] = fz.head_w[n0 + n * 16 + lk * 4 + r];
    const float hb = fz.head_b[0];
    float* lg = fz.logits + (size_t)img * H * W;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      float p = 0.0f;
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) p = __builtin_fmaf(fmaxf(acc[m][n][r], 0.0f), hw[n][r], p);
      p += __shfl_xor(p, 16);
      p += __shfl_xor(p, 32);
      if (lk == 0) lg[(size_t)(y0 + wave * 4 + m) * W + x0 + li] = hb + p;
    }
    return;
  }
  ET* out = dst + (size_t)img * OH * OW * Cout;
#pragma unroll
  for (int m = 0; m < 4; ++m) {
    int gy = y0 + wave * 4 + m, gx = x0 + li;
    int oy = TAPS == 9 ? gy : 2 * gy + (phase >> 1), ox = TAPS == 9 ? gx : 2 * gx + (phase & 1);
#pragma unroll
    for (int n = 0; n < NT; ++n) {
      v4 o;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float v = acc[m][n][r];
        if (relu) v = fmaxf(v, 0.0f);
        o[r] = (ET)v;
      }
      *(v4*)(out + act_off((size_t)OH * OW, (size_t)oy * OW + ox, n0 + n * 16 + lk * 4)) = o;
    }
  }
  if (FUSE & UF_POOL) {
    // 2x2 max pool of this wave's 4 rows x 16 pixels: rows pair inside the lane, columns pair with lane li ^ 1
    // (max commutes with the monotone bf16 rounding, so this equals pooling the stored tensor)
    ET* po = (ET*)fz.pooled + (size_t)img * (H / 2) * (W / 2) * Cout;
#pragma unroll
    for (int mp = 0; mp < 2; ++mp)
#pragma unroll
      for (int n = 0; n < NT; ++n) {
        v4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = fmaxf(acc[2 * mp][n][r], acc[2 * mp + 1][n][r]);
          v = fmaxf(v, __shfl_xor(v, 1));
          if (relu) v = fmaxf(v, 0.0f);
          o[r] = (ET)v;
        }
        if ((li & 1) == 0)
          *(v4*)(po + act_off((size_t)(H / 2) * (W / 2), (size_t)((y0 + wave * 4) / 2 + mp) * (W / 2) + (x0 + li) / 2, n0 + n * 16 + lk * 4)) = o;
      }
  }
}

template <int EK>
__global__ void k_conv_first16(const float* __restrict__ img, const float* __restrict__ wgt, const float* __restrict__ bias,
                               u16* __restrict__ dst_, int H, int W, int C, int nimg) {
  using ET = typename EKT<EK>::type;
  ET* dst = (ET*)dst_;

  using v8 = typename E16<ET>::v8;
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
    ET* o = dst + im * H * W * C;
    const size_t pin = p - im * (size_t)H * W;
    for (int c8 = 0; c8 < C; c8 += 8) {
      v8 ov;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        float a = sw[9 * SH_UNET_MAXBASE + c8 + k];
#pragma unroll
        for (int t = 0; t < 9; ++t) a = __builtin_fmaf(v[t], sw[t * C + c8 + k], a);
        ov[k] = (ET)fmaxf(a, 0.0f);
      }
      *(v8*)(o + act_off((size_t)H * W, pin, c8)) = ov;
    }
  }
}

template <int EK>
__global__ void k_maxpool2_16(const u16* __restrict__ src_, u16* __restrict__ dst_, int H, int W, int C, int nimg) {
  using ET = typename EKT<EK>::type;
  const ET* src = (const ET*)src_;
  ET* dst = (ET*)dst_;

  using v8 = typename E16<ET>::v8;
  const int OH = H / 2, OW = W / 2, C8 = C / 8;
  size_t total = (size_t)nimg * OH * OW * C8;
  for (size_t e = blockIdx.x * (size_t)blockDim.x + threadIdx.x; e < total; e += (size_t)gridDim.x * blockDim.x) {
    int c8 = (int)(e % C8);
    size_t p = e / C8;
    int ox = (int)(p % OW), oy = (int)((p / OW) % OH);
    size_t im = p / ((size_t)OH * OW);
    const ET* s = src + im * H * W * C + act_off((size_t)H * W, (size_t)(2 * oy) * W + 2 * ox, c8 * 8);
    v8 a = *(const v8*)s, b = *(const v8*)(s + 32), c = *(const v8*)(s + (size_t)W * 32), d = *(const v8*)(s + (size_t)W * 32 + 32);
    v8 r;
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = (ET)fmaxf(fmaxf((float)a[k], (float)b[k]), fmaxf((float)c[k], (float)d[k]));
    *(v8*)(dst + im * OH * OW * C + act_off((size_t)OH * OW, (size_t)oy * OW + ox, c8 * 8)) = r;
  }
}

template <int EK>
__global__ void k_head16(const u16* __restrict__ src_, const float* __restrict__ w, const float* __restrict__ bp,
                         float* __restrict__ logits, int C, size_t npix, size_t HW) {
  using ET = typename EKT<EK>::type;
  const ET* src = (const ET*)src_;

  using v8 = typename E16<ET>::v8;
  __shared__ float sw[SH_UNET_MAXBASE];
  for (int i = threadIdx.x; i < C; i += blockDim.x) sw[i] = w[i];
  __syncthreads();
  const float b = bp[0];
  for (size_t p = blockIdx.x * (size_t)blockDim.x + threadIdx.x; p < npix; p += (size_t)gridDim.x * blockDim.x) {
    const size_t im = p / HW, pin = p - im * HW;
    const ET* s = src + im * HW * C;
    float a = b;
    for (int c8 = 0; c8 < C; c8 += 8) {
      v8 v = *(const v8*)(s + act_off(HW, pin, c8));
#pragma unroll
      for (int k = 0; k < 8; ++k) a = __builtin_fmaf((float)v[k], sw[c8 + k], a);
    }
    logits[p] = a;
  }
}

}  // namespace sh
