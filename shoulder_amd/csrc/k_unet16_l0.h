// k_unet16_l0.h -- the full-resolution (32-channel) level of the 16-bit UNet as persistent fused kernels.
//
// Level 0 moves 1.07 GB per 32-channel tensor at B = 64 and carries 21 % of the flops: layer by layer it is bound by
// neither roof (round 1: enc0a+enc0b+pool 1.05 ms = 0.12 of the MFMA peak and 0.19 of HBM; up0 / dec0a / dec0b+head
// 1.7 + 3.7 + 1.3 GB of traffic).  Here a 32-channel full-resolution tensor that is produced and consumed inside the level
// never reaches HBM:
//   k_enc0_fused16   image -> [enc0a: 1 -> 32, 3x3] -> LDS -> [enc0b: 32 -> 32, 3x3] -> skip0 (+ 2x2 max pool -> level 1)
// One workgroup (8 waves) per CU walks (image, 32x16-pixel tile) items.  The conv weights stay in LDS for the whole
// launch (the two-barrier kernel re-read 18 KB of weights per 16 KB of output); the first conv of item i+1 is computed on
// the matrix cores into the other LDS halo buffer BETWEEN the MFMA groups of item i's second conv (its fragment building is
// VALU work that the matrix pipe's 16-cycle instructions cover), so one s_barrier per item separates producer and consumer.
// HBM floor: 4 B/px image in, 64 + 16 B/px out = 1.41 GB per launch at B = 64.
//
// First conv on the matrix cores without per-fragment VALU work: the image patch of an item is split once into its ET high
// part and ET low part (v - hi) and written to LDS as 8 copies shifted by 0..7 elements, so that the 8 consecutive patch
// values e[q .. q + 7] any lane needs are ONE aligned ds_read_b128 from copy q & 7.  K layout of the MFMA: lane group
// lk = patch row ty (0..2), element j = column offset (weights w[ty][j] for j < 3, zero beyond; lk = 3 reads zeros), one
// MFMA on the high parts and one on the low parts per 16 pixels x 16 channels (image precision ~2^-17, ET-rounded weights,
// f32 accumulate from the bias -- the arithmetic of k_conv_mfma16<EK, 9, 2, UF_FIRST | UF_POOL>, summed in another order;
// the second conv is the same operation for operation: tests/test_gpu_unet_bf16.py::test_level0_fused_matches_two_barrier_kernel).
#pragma once
#include "k_unet_bf16_dma.h"

namespace sh {

#define L0_THREADS 512
#define L0_PW 36                                   // halo-tile pitch in pixels (34 used), as in k_conv3_dma16
#define L0_INROWS (18 * L0_PW)                     // 648 halo rows of 64 B
#define L0_BUF (L0_INROWS * 64)                    // 41472
#define L0_WOFF (2 * L0_BUF)                       // 82944: [9 taps][32 couts] rows of 64 B
#define L0_WBYTES (288 * 64)
#define L0_IMGOFF (L0_WOFF + L0_WBYTES)            // 101376: two patch slots, each 8 shifted copies of the high part, then of the low part
#define L0_PATCH 744                               // patch elements written: 20 rows x 36 = 720, + zeros up to the reach of the last rows' 8-element reads
#define L0_CPAD 16                                 // bytes in front of a copy's element 0: copy s is written at indices -s .. 743 - s without a bounds test
#define L0_CSTR 1568                               // bytes between copies (16 + 2 * 744 = 1504 used): 1568 % 256 == 32, so the 16 lanes of a fragment read hit distinct 16-B slots
#define L0_SLOT (16 * L0_CSTR)                     // 25088 bytes per patch slot
#define L0_ZEROOFF (L0_IMGOFF + 2 * L0_SLOT)       // 16 zero bytes (fragment of lane group lk = 3)
#define L0_BIASOFF (L0_ZEROOFF + 128)
#define L0_SMEM (L0_BIASOFF + 256)                 // 151936

template <int EK>
__global__ void __launch_bounds__(L0_THREADS)
k_enc0_fused16(const float* __restrict__ image, const float* __restrict__ w0 /*[9][32] f32*/, const float* __restrict__ b0 /*[32]*/,
               const u16* __restrict__ wgt_ /*enc0b, packed [9][1][32][32]*/, const float* __restrict__ bias /*[32]*/,
               u16* __restrict__ skip_, u16* __restrict__ pooled_, int H, int W, int nimg,
               const double* __restrict__ raw /*nullable: the UNSCALED image [nimg][H][W] f64 and ...*/,
               const unsigned long long* __restrict__ mm_enc /*... its minimum / complemented maximum per image, encoded (k_anp_rows): the MinMaxScaler
               arithmetic of k_anp_scale is then applied where the patch is read, and the f32 image (a 201 MB pass of its own at B = 64)
               is never made*/) {
  using ET = typename EKT<EK>::type;
  using v8 = typename E16<ET>::v8;
  const ET* wgt = (const ET*)wgt_;
  ET* skip = (ET*)skip_;
  ET* pooled = (ET*)pooled_;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[L0_SMEM];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int xh = wave & 1, rg = wave >> 1;
  const int tiles_x = W / 32, tiles_y = H / 16;
  const int total = nimg * tiles_x * tiles_y;
  const int per = (total + gridDim.x - 1) / gridDim.x;
  const int w_begin = blockIdx.x * per, w_end = min(total, w_begin + per);
  if (w_begin >= w_end) return;
  const int nitems = w_end - w_begin;

  // ---- once per workgroup: enc0b weights -> LDS (swizzled rows), bias, first-conv fragments in registers
  {
    ET* s_w = (ET*)(smem + L0_WOFF);
    // LDS row 32 tap + 16 n + i holds output channel 8 (i >> 2) + 4 n + (i & 3): lane group lk then owns channels
    // 8 lk .. 8 lk + 7 of its pixels -- one 16-byte store per pixel row instead of two 8-byte ones (k_unet_bf16_dma.h)
    for (int e = tid; e < 288 * 4; e += L0_THREADS) {
      const int q = e & 3, r = e >> 2, j = r & 31;
      const int ch = 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);
      *(u32x4*)(s_w + UB_OFF(r, q)) = *(const u32x4*)(wgt + (size_t)((r & ~31) + ch) * 32 + q * 8);
    }
    float* s_bias = (float*)(smem + L0_BIASOFF);
    if (tid < 32) s_bias[tid] = bias[tid];
  }
  v8 wA[2];
  f32x4 b0v[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float wv = w0[(min(lk, 2) * 3 + min(j, 2)) * 32 + 8 * (li >> 2) + 4 * n + (li & 3)];
      wA[n][j] = (lk < 3 && j < 3) ? (ET)wv : (ET)0.0f;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) b0v[n][r] = b0[8 * lk + 4 * n + r];
  }
  if (tid < 4) ((unsigned*)(smem + L0_ZEROOFF))[tid] = 0u;

  // fragment read offsets of the second conv (bytes inside a halo buffer / the weight region)
  int xoff[2][3], woff;
  {
    const int rowbase = rg * 4 * L0_PW + xh * 16 + li;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) xoff[sp][dx] = UB_OFF(rowbase + sp * L0_PW + dx, lk) * 2;
    woff = UB_OFF(li, lk) * 2;
  }

  auto item_coords = [&](int w, int& img, int& x0, int& y0) {
    const int tx = w % tiles_x; w /= tiles_x;
    const int ty = w % tiles_y; img = w / tiles_y;
    x0 = tx * 32; y0 = ty * 16;
  };
  // image patch of an item: rows y0-2 .. y0+17, columns x0-2 .. x0+33 (zero outside the image), 720 values e[row * 36 + col].
  // From the unscaled image the values stay doubles while the loads are in flight; patch_store applies X * scale_ + min_
  // (k_anp_scale's expression, so the same float) when it splits them.
  struct Patch { float f[2]; double d[2]; double sc, mn; unsigned ok; };
  auto patch_load = [&](int w, Patch& r) {
    int img, x0, y0;
    item_coords(w, img, x0, y0);
    r.ok = 0u;
    if (raw != nullptr) {
      const unsigned long long elo = mm_enc[2 * img], ehi = ~mm_enc[2 * img + 1];      // order-preserving encoding (k_slices.h: enc_f64 / dec_f64)
      const double lo = __longlong_as_double((long long)((elo & 0x8000000000000000ull) ? (elo & 0x7FFFFFFFFFFFFFFFull) : ~elo));
      const double hi = __longlong_as_double((long long)((ehi & 0x8000000000000000ull) ? (ehi & 0x7FFFFFFFFFFFFFFFull) : ~ehi));
      double rng = hi - lo;
      if (rng == 0.0) rng = 1.0;
      r.sc = 1.0 / rng; r.mn = 0.0 - lo * r.sc;
      const double* im = raw + (size_t)img * H * W;
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int e = tid + k * L0_THREADS;
        const int py = e / L0_PW, px = e - py * L0_PW;
        const int gy = y0 - 2 + py, gx = x0 - 2 + px;
        const bool in = e < 20 * L0_PW && gy >= 0 && gy < H && gx >= 0 && gx < W;
        r.d[k] = in ? im[(size_t)gy * W + gx] : 0.0;
        r.ok |= in ? 1u << k : 0u;
      }
      return;
    }
    const float* im = image + (size_t)img * H * W;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + k * L0_THREADS;
      const int py = e / L0_PW, px = e - py * L0_PW;
      const int gy = y0 - 2 + py, gx = x0 - 2 + px;
      r.f[k] = (e < 20 * L0_PW && gy >= 0 && gy < H && gx >= 0 && gx < W) ? im[(size_t)gy * W + gx] : 0.0f;
    }
  };
  // split into high and low parts; copy s holds e[i + s] at index i (i = -s .. 743 - s; the first s entries are padding)
  auto patch_store = [&](int slot, const Patch& r) {
    unsigned char* base = smem + L0_IMGOFF + slot * L0_SLOT + L0_CPAD;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int e = tid + k * L0_THREADS;
      if (k == 0 || e < L0_PATCH) {
        const float val = raw != nullptr ? ((r.ok >> k & 1u) ? (float)(r.d[k] * r.sc + r.mn) : 0.0f) : r.f[k];
        const ET hi = (ET)val;
        const ET lo = (ET)(val - (float)hi);
        unsigned char* pe = base + 2 * e;
#pragma unroll
        for (int sft = 0; sft < 8; ++sft) {
          *(ET*)(pe + sft * (L0_CSTR - 2)) = hi;                       // copy sft, index e - sft
          *(ET*)(pe + sft * (L0_CSTR - 2) + 8 * L0_CSTR) = lo;
        }
      }
    }
  };
  // first conv for 16 halo pixels (group g = wave + 8 t) of the item at (x0, y0): patch slot -> halo buffer.  Everything that
  // depends only on the lane and the group is computed once per launch.
  int g_px[6], g_py[6], g_frag[6], g_out[6];
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const int p = 16 * (wave + 8 * t) + li, pc = p < L0_INROWS ? p : L0_INROWS - 1;
    g_py[t] = pc / L0_PW; g_px[t] = pc - g_py[t] * L0_PW;
    const int q = pc + lk * L0_PW;              // first of the 8 consecutive patch values of this lane: row py + ty, columns px ..
    const int sft = q & 7;
    g_frag[t] = lk < 3 ? L0_IMGOFF + L0_CPAD + sft * L0_CSTR + 2 * (q - sft) : -1;
    g_out[t] = p < L0_INROWS ? UB_OFF(p, lk) * 2 : -1;
  }
  auto enc0a_group = [&](int t, int slot, int hb, int x0, int y0) {
    const int gy = y0 - 1 + g_py[t], gx = x0 - 1 + g_px[t];
    const bool inside = g_px[t] < 34 && gy >= 0 && gy < H && gx >= 0 && gx < W;
    const unsigned char* fp = g_frag[t] >= 0 ? smem + g_frag[t] + slot * L0_SLOT : smem + L0_ZEROOFF;
    const v8 hi = *(const v8*)fp;
    const v8 lo = *(const v8*)(g_frag[t] >= 0 ? fp + 8 * L0_CSTR : fp);
    v8 o;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      f32x4 a = E16<ET>::mfma(wA[n], hi, b0v[n]);
      a = E16<ET>::mfma(wA[n], lo, a);
#pragma unroll
      for (int r = 0; r < 4; ++r) o[4 * n + r] = inside ? (ET)fmaxf(a[r], 0.0f) : (ET)0.0f;      // outside the image: the second conv's zero padding
    }
    if (g_out[t] >= 0) *(v8*)(smem + hb * L0_BUF + g_out[t]) = o;      // channels 8 lk .. 8 lk + 7 = 16-byte slot lk of the pixel's row
  };

  // ---- prologue: patch 0 -> LDS, first conv of item 0 into buffer 0, patch 1 -> LDS
  Patch pr;
  {
    patch_load(w_begin, pr);
    patch_store(0, pr);
    if (nitems > 1) patch_load(w_begin + 1, pr);
    __syncthreads();
    int img, x0, y0;
    item_coords(w_begin, img, x0, y0);
#pragma unroll
    for (int t = 0; t < 6; ++t)
      if (wave + 8 * t < (L0_INROWS + 15) / 16) enc0a_group(t, 0, 0, x0, y0);
    if (nitems > 1) patch_store(1, pr);
    __syncthreads();
  }
  const float* s_bias = (const float*)(smem + L0_BIASOFF);
  for (int i = 0; i < nitems; ++i) {
    int c_img, c_x0, c_y0;
    item_coords(w_begin + i, c_img, c_x0, c_y0);
    int n_img = 0, n_x0 = 0, n_y0 = 0;
    const bool has_next = i + 1 < nitems;
    if (has_next) item_coords(w_begin + i + 1, n_img, n_x0, n_y0);
    const bool has_next2 = i + 2 < nitems;
    if (has_next2) patch_load(w_begin + i + 2, pr);      // in flight during the MFMAs below
    f32x4 acc[4][2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const f32x4 bv = *(const f32x4*)(s_bias + 8 * lk + 4 * n);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m][n] = bv;
    }
    const unsigned char* sb = smem + (i & 1) * L0_BUF;
    const unsigned char* wbp = smem + L0_WOFF + woff;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int dy = tap / 3, dx = tap % 3;
      v8 xf[4], wf[2];
#pragma unroll
      for (int m = 0; m < 4; ++m) { const int s = m + dy; xf[m] = *(const v8*)(sb + xoff[s & 1][dx] + (s & ~1) * L0_PW * 64); }
#pragma unroll
      for (int n = 0; n < 2; ++n) wf[n] = *(const v8*)(wbp + (tap * 32 + n * 16) * 64);
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = E16<ET>::mfma(wf[n], xf[m], acc[m][n]);
      // the next item's first conv, one 16-pixel group per tap: patch slot (i+1)&1 -> halo buffer (i+1)&1
      if (has_next && tap < 6 && wave + 8 * tap < (L0_INROWS + 15) / 16) enc0a_group(tap, (i + 1) & 1, (i + 1) & 1, n_x0, n_y0);
    }
    // epilogue: bias is in the accumulators; ReLU, round, store 4 consecutive couts per lane (+ the 2x2 max pool)
    ET* out = skip + (size_t)c_img * H * W * 32;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int gy = c_y0 + rg * 4 + m, gx = c_x0 + xh * 16 + li;
      v8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = (ET)fmaxf(acc[m][r >> 2][r & 3], 0.0f);
      *(v8*)(out + ((size_t)gy * W + gx) * 32 + 8 * lk) = o;
    }
    ET* po = pooled + (size_t)c_img * (H / 2) * (W / 2) * 32;
#pragma unroll
    for (int mp = 0; mp < 2; ++mp) {
      v8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        float v = fmaxf(acc[2 * mp][r >> 2][r & 3], acc[2 * mp + 1][r >> 2][r & 3]);
        v = fmaxf(v, __shfl_xor(v, 1));
        o[r] = (ET)fmaxf(v, 0.0f);
      }
      if ((li & 1) == 0)
        *(v8*)(po + ((size_t)((c_y0 + rg * 4) / 2 + mp) * (W / 2) + (c_x0 + xh * 16 + li) / 2) * 32 + 8 * lk) = o;
    }
    if (has_next2) patch_store(i & 1, pr);      // patch of item i+2 (slot (i+2)&1; its last reader was the first conv of item i, one barrier ago)
    __syncthreads();
  }
}

}  // namespace sh

namespace sh {

// ---- 2x2 stride-2 transposed convolution (the decoder's up-sampling), both column phases of an output row per workgroup --
// out[2y + dy][2x + dx][co] = b[co] + sum_ci in[y][x][ci] * w[dy * 2 + dx][ci][co]: per output pixel one K = Cin product, no
// spatial reuse -- a memory-bound layer (up0 at B = 64: 0.54 GB in, 1.07 GB out, 0.07 TFLOP).  The two-barrier kernel ran
// it as one launch of 4 x Cout/64 workgroups per tile: every workgroup re-read the input tile, and the two dx phases of an
// output row -- the two 64-byte halves of each 128-byte line -- were written by different workgroups.  Here a workgroup
// (4 waves) owns a 16x16 source tile x 32 output channels x the two dx phases of one row parity dy: the tile is staged
// once per 32-channel chunk and multiplied against both phases' weights (pixel fragments reused), and both halves of
// every output line leave the same wave back to back.  Channels are dealt to the MFMA rows as in k_conv3_dma16 (16-byte stores).
// A workgroup takes one output-row parity dy (both dx): 32 accumulator tiles for all four phases cost 128 VGPRs and left two
// workgroups per CU, which made the kernel the slowest one beside another lane's kernels (0.47 ms average in the two-lane
// bench against 0.27 ms alone); with 16 tiles four to five workgroups fit.
#define UPC_THREADS 256

template <int EK>
__global__ void __launch_bounds__(UPC_THREADS)
k_upconv16(const u16* __restrict__ src_, int Cin, const u16* __restrict__ wgt_ /*packed [4][Cin/32][Cout][32]*/, const float* __restrict__ bias,
           u16* __restrict__ dst_, int H, int W, int Cout) {
  using ET = typename EKT<EK>::type;
  using v8 = typename E16<ET>::v8;
  const ET* wgt = (const ET*)wgt_;
  __shared__ __attribute__((aligned(16))) ET s_in[256 * UB_PSTR];
  __shared__ __attribute__((aligned(16))) ET s_w[64 * UB_PSTR];
  const int tiles_x = W / 16;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x;
  const int n0 = blockIdx.y * 32, img = blockIdx.z >> 1, dy = blockIdx.z & 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int x0 = tx * 16, y0 = ty * 16;
  const int nchunk = Cin / 32;
  const ET* in = (const ET*)src_ + (size_t)img * H * W * Cin;

  // staging plan: 4 input pieces + 1 weight piece of 16 bytes per thread and chunk
  int in_src[4], in_lds[4], wt_src[1], wt_lds[1];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int e = tid + k * UPC_THREADS, q = e & 3, p = e >> 2;
    in_src[k] = ((y0 + (p >> 4)) * W + x0 + (p & 15)) * 32 + q * 8;
    in_lds[k] = UB_OFF(p, q);
  }
  {
    const int e = tid, q = e & 3, r = e >> 2, dx = r >> 5, j = r & 31;
    const int ch = 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);      // LDS row 16 n + i of a phase holds channel 8 (i >> 2) + 4 n + (i & 3)
    wt_src[0] = ((dy * 2 + dx) * nchunk * Cout + n0 + ch) * 32 + q * 8;
    wt_lds[0] = UB_OFF(r, q);
  }
  u32x4 rin[4], rwt[1];
  auto load_chunk = [&](int cc) {
    const ET* s = in + (size_t)cc * H * W * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) rin[k] = *(const u32x4*)(s + in_src[k]);
    rwt[0] = *(const u32x4*)(wgt + (size_t)wt_src[0] + (size_t)cc * Cout * 32);
  };

  f32x4 acc[4][2][2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    f32x4 bv;
#pragma unroll
    for (int r = 0; r < 4; ++r) bv[r] = bias[n0 + 8 * lk + 4 * n + r];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int ph = 0; ph < 2; ++ph) acc[m][ph][n] = bv;
  }
  load_chunk(0);
  for (int cc = 0; cc < nchunk; ++cc) {
    __syncthreads();                  // every wave is done reading the previous chunk
#pragma unroll
    for (int k = 0; k < 4; ++k) *(u32x4*)(s_in + in_lds[k]) = rin[k];
    *(u32x4*)(s_w + wt_lds[0]) = rwt[0];
    __syncthreads();
    if (cc + 1 < nchunk) load_chunk(cc + 1);      // in flight during the MFMAs below
    v8 xf[4];
#pragma unroll
    for (int m = 0; m < 4; ++m) xf[m] = *(const v8*)(s_in + UB_OFF((wave * 4 + m) * 16 + li, lk));
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {      // ph = dx
      v8 wf[2];
#pragma unroll
      for (int n = 0; n < 2; ++n) wf[n] = *(const v8*)(s_w + UB_OFF(ph * 32 + n * 16 + li, lk));
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][ph][n] = E16<ET>::mfma(wf[n], xf[m], acc[m][ph][n]);
    }
  }
  const int OW = 2 * W, OH = 2 * H;
  ET* out = (ET*)dst_ + (size_t)img * OH * OW * Cout + (size_t)blockIdx.y * OH * OW * 32;      // this group's 32-channel plane
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
      const int oy = 2 * (y0 + wave * 4 + m) + dy, ox = 2 * (x0 + li) + ph;
      v8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = (ET)acc[m][ph][r >> 2][r & 3];
      *(v8*)(out + ((size_t)oy * OW + ox) * 32 + 8 * lk) = o;
    }
}


// ---- the same layer with the source pixels held in REGISTERS ---------------------------------------------------------------
// k_upconv16 stages a 16 x 16 source tile per 32-channel chunk and (row parity, 32-cout group): the tile is read 2 Cout / 32 times
// from L2, and a chunk is 16 MFMAs per wave between two workgroup barriers (matrix pipe busy 0.19, 0.3-0.4 of the HBM roof).
// Here a workgroup (8 waves) owns a 32 x 16 source tile for ALL 4 Cout outputs of its pixels: every wave loads its 4 rows x 16
// pixels x Cin once, straight into the MFMA pixel fragments (4 NCH fragments = 64 / 128 VGPRs for Cin = 128 / 256), and keeps
// them for the whole launch; only the weights -- one (32-cout group, phase) slice of NCH x 32 rows at a time, 8 / 16 KB -- go through
// LDS, fetched into registers while the previous slice is multiplied (one barrier per slice of 4 x 2 x NCH MFMAs per wave).
// Input bytes come from HBM exactly once, the two 64-byte halves of an output line (dx = 0, 1) leave the same wave in
// consecutive slices.  Same accumulation (bias in the accumulator, chunks in order) and the same channel dealing as
// k_upconv16: bit-identical results (tests/test_gpu_unet_bf16.py::test_register_resident_upconv_bit_identical).
// Cin = 512 (up3): 2 rows per wave (MT = 2, 128 fragment registers), a 32 x 8 source tile per workgroup.
#define UPR_THREADS 512

template <int EK, int NCH, int MT = 4>      // MT: source rows (16-pixel MFMA tiles) per wave; a workgroup owns 4 MT rows x 32 pixels
__global__ void __launch_bounds__(UPR_THREADS)
k_upconv16r(const u16* __restrict__ src_, const u16* __restrict__ wgt_ /*packed [4][NCH][Cout][32]*/, const float* __restrict__ bias,
            u16* __restrict__ dst_, int H, int W, int Cout) {
  using ET = typename EKT<EK>::type;
  using v8 = typename E16<ET>::v8;
  const ET* wgt = (const ET*)wgt_;
  constexpr int WROWS = NCH * 32;                      // LDS rows of a weight slice: [chunk][16 n + i]
  constexpr int WPASS = WROWS * 4 / UPR_THREADS;       // 16-byte pieces per thread and slice: 1 / 2
  static_assert(WROWS * 4 % UPR_THREADS == 0, "whole passes");
  __shared__ __attribute__((aligned(16))) ET s_w[2][WROWS * UB_PSTR];
  __shared__ __attribute__((aligned(16))) float s_b[512];
  const int tiles_x = W / 32;
  const int tx = blockIdx.x % tiles_x, ty = blockIdx.x / tiles_x, img = blockIdx.y;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (tid < Cout) s_b[tid] = bias[tid];
  const int li = lane & 15, lk = lane >> 4;
  const int xh = wave & 1, rg = wave >> 1;
  const int x0 = tx * 32 + xh * 16 + li, y0 = (ty * 4 + rg) * MT;
  const ET* in = (const ET*)src_ + (size_t)img * H * W * (NCH * 32);
  const int groups = Cout >> 5, nslice = groups * 4;

  // weight slice t = (group g = t >> 2, phase ph = t & 3): piece e -> LDS row r = e >> 2 = 32 cc + j, slot q
  int wt_src[WPASS], wt_lds[WPASS];
#pragma unroll
  for (int k = 0; k < WPASS; ++k) {
    const int e = tid + k * UPR_THREADS, q = e & 3, r = e >> 2, cc = r >> 5, j = r & 31;
    const int ch = 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);      // LDS row 16 n + i holds channel 8 (i >> 2) + 4 n + (i & 3) of the group
    wt_src[k] = (cc * Cout + ch) * 32 + q * 8;
    wt_lds[k] = UB_OFF(r, q);
  }
  u32x4 rwt[WPASS];
  auto load_slice = [&](int t) {
    const ET* wsl = wgt + ((size_t)(t & 3) * NCH * Cout + (size_t)(t >> 2) * 32) * 32;
#pragma unroll
    for (int k = 0; k < WPASS; ++k) rwt[k] = *(const u32x4*)(wsl + wt_src[k]);
  };
  auto put_slice = [&](int b) {
#pragma unroll
    for (int k = 0; k < WPASS; ++k) *(u32x4*)(s_w[b] + wt_lds[k]) = rwt[k];
  };
  load_slice(0);
  // the wave's pixels: fragment (m, cc) = 8 channels 32 cc + 8 lk .. of pixel (y0 + m, x0)
  v8 xf[MT][NCH];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) xf[m][cc] = *(const v8*)(in + ((size_t)cc * H * W + (size_t)(y0 + m) * W + x0) * 32 + 8 * lk);
  // the pixel fragments are USED here, so hipcc's waits for them stand in front of the slice loop: inside it they were counted waits
  // that ended in s_waitcnt vmcnt(0) in the middle of every slice -- the slice's own weight prefetch and the previous slice's stores
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) { u32x4 t = __builtin_bit_cast(u32x4, xf[m][cc]); asm volatile("" : "+v"(t)); xf[m][cc] = __builtin_bit_cast(v8, t); }
  put_slice(0);
  __syncthreads();

  const int OW = 2 * W, OH = 2 * H;
  ET* out0 = (ET*)dst_ + (size_t)img * OH * OW * Cout;
  for (int t = 0; t < nslice; ++t) {
    const int g = t >> 2, dy = (t >> 1) & 1, dx = t & 1;
    if (t + 1 < nslice) load_slice(t + 1);      // in flight during the MFMAs below
    f32x4 acc[MT][2];
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const f32x4 bv = *(const f32x4*)(s_b + g * 32 + 8 * lk + 4 * n);
#pragma unroll
      for (int m = 0; m < MT; ++m) acc[m][n] = bv;
    }
    const ET* sw = s_w[t & 1];
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) {
      v8 wf[2];
#pragma unroll
      for (int n = 0; n < 2; ++n) wf[n] = *(const v8*)(sw + UB_OFF(cc * 32 + n * 16 + li, lk));
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = E16<ET>::mfma(wf[n], xf[m][cc], acc[m][n]);
    }
    ET* out = out0 + (size_t)g * OH * OW * 32;      // this group's 32-channel plane
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      v8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = (ET)acc[m][r >> 2][r & 3];
      *(v8*)(out + ((size_t)(2 * (y0 + m) + dy) * OW + 2 * x0 + dx) * 32 + 8 * lk) = o;
    }
    if (t + 1 < nslice) put_slice((t + 1) & 1);      // its last readers passed the barrier that ended slice t - 1
    __syncthreads();
  }
}

}  // namespace sh
