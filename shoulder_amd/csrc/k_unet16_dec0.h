// k_unet16_dec0.h -- the level-0 decoder's first conv with its up-convolution fused in:
//   dec0a( concat( skip0 [32 ch, H x W],  up0( low [64 ch, H/2 x W/2] ) ) )   ->  32 ch, H x W
// Unfused, up0 writes its 32-channel full-resolution output (1.07 GB at B = 64) and dec0a reads it back: both layers are HBM
// bound there (0.33 + 0.71 ms).  Here the persistent LDS-DMA conv of k_unet_bf16_dma.h (NN = 2, weights resident, dx-major
// taps, work tickets) computes the up-conv half of its input tile itself: the 10 x 18 low-resolution pixels under the
// 18 x 34 halo tile are staged by LDS-DMA, multiplied on the matrix cores against the four phase matrices of up0 (resident
// in LDS), rounded to the element type exactly as k_upconv16 rounds them (bias in the accumulator, chunks in order) and
// written into the conv's second input buffer -- same values as the two launches bit for bit, 2.1 GB less HBM traffic,
// one launch less.
//
// LDS (bytes): two halo buffers 2 x 41 472 | dec0a weights [2 chunks][9 taps][32 rows] 36 864 | up0 weights [4 phases][2 chunks]
// [32 rows] 16 384 | low tile [2 chunks][180 rows] 23 040 (+ 1 536 of padding: every wave issues all three pieces, so the
// counted waits below are exact) | biases 256  = 161 024.
// Per item (a 32 x 16 output tile), three barriers:
//   wait(skip tile of this item in A)                                                barrier
//   step 0: MFMAs of the skip chunk on A;  wait(low tile of this item in L)          barrier   (A and nothing else is free)
//   DMA skip tile of the NEXT item -> A;  up-conv of this item: L -> B               barrier   (L is free)
//   DMA low tile of the NEXT item -> L;   step 1: MFMAs of the up chunk on B;  epilogue stores
#pragma once
#include "k_unet_bf16_dma.h"

namespace sh {

#define D0_A1 (UD_INROWS * 64)                   // 41 472: second halo buffer
#define D0_WCONV (2 * UD_INROWS * 64)            // 82 944
#define D0_WUP (D0_WCONV + 2 * 9 * 32 * 64)      // 119 808
#define D0_LOW (D0_WUP + 4 * 2 * 32 * 64)        // 136 192
#define D0_LOWROWS 180                           // 10 x 18 low-resolution pixels
#define D0_LOWCH (D0_LOWROWS * 64)               // 11 520: one 32-channel chunk of the low tile
#define D0_BIAS (D0_LOW + 3 * 8192)              // 160 768 (the low tile's three DMA pieces write 1 536 slots: 96 of them padding)
#define D0_SMEM (D0_BIAS + 256)                  // 161 024

template <int EK>
__global__ void __launch_bounds__(UD_THREADS)
k_dec0a_up16(const u16* __restrict__ skip_ /*[img][H W][32]*/, const u16* __restrict__ low_ /*[img][2][H/2 W/2][32]*/,
             const u16* __restrict__ wgt_ /*dec0a packed [9][2][32][32]*/, const float* __restrict__ bias,
             const u16* __restrict__ wup_ /*up0 packed [4][2][32][32]*/, const float* __restrict__ upb,
             u16* __restrict__ dst_ /*[img][H W][32]*/, int H, int W, int nimg, const u16* __restrict__ zero_page_,
             unsigned* __restrict__ ticket, const int* __restrict__ tk_tab, int ntk, YieldArg yl) {
  using ET = typename EKT<EK>::type;
  using v8 = typename E16<ET>::v8;
  const ET* skip = (const ET*)skip_;
  const ET* low = (const ET*)low_;
  const ET* wgt = (const ET*)wgt_;
  const ET* wup = (const ET*)wup_;
  ET* dst = (ET*)dst_;
  const ET* zero_page = (const ET*)zero_page_;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[D0_SMEM];
  __shared__ int s_q[2];
  constexpr int NSTORE = 4;                       // dwordx4 stores per wave and item
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int xh = wave & 1, rg = wave >> 1;
  const int H2 = H >> 1, W2 = W >> 1;
  const int tiles_x = W / 32, tiles_y = H / 16;
  const int total = nimg * tiles_x * tiles_y;
  const bool dyn = ticket != nullptr;
  if (dyn && tid == 0) { s_q[0] = ud_take_ticket(ticket, ntk, yl); s_q[1] = ud_take_ticket(ticket, ntk, yl); }

  // ---- once per workgroup: both weight sets and the biases -> LDS
  {
    // dec0a: LDS row (chunk, tap, 16 n + i) <- packed row (tap, chunk, channel 8 (i >> 2) + 4 n + (i & 3)); slot swizzle on the source
    for (int e = tid; e < 2 * 9 * 32 * 4; e += UD_THREADS) {
      const int row = e >> 2, q = e & 3;
      const int cc = row / 288, rem = row - cc * 288, tap = rem >> 5, j = rem & 31;
      const int ch = 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);
      *(u32x4*)(smem + D0_WCONV + e * 16) = *(const u32x4*)(wgt + (size_t)((tap * 2 + cc) * 32 + ch) * 32 + ((q ^ ((row >> 1) & 2)) << 3));
    }
    // up0: LDS row (phase, chunk, 16 n + i) <- packed row (phase, chunk, channel ...), same dealing as k_upconv16
    for (int e = tid; e < 4 * 2 * 32 * 4; e += UD_THREADS) {
      const int row = e >> 2, q = e & 3, j = row & 31;
      const int ch = 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);
      *(u32x4*)(smem + D0_WUP + e * 16) = *(const u32x4*)(wup + (size_t)((row & ~31) + ch) * 32 + ((q ^ ((row >> 1) & 2)) << 3));
    }
    float* sbf = (float*)(smem + D0_BIAS);
    if (tid < 32) sbf[tid] = bias[tid];
    else if (tid < 64) sbf[tid] = upb[tid - 32];
  }
  __syncthreads();       // every ordinary load is retired before the first LDS-DMA is issued
  int qk = 1;
  int w_begin, w_end;
  if (dyn) {
    const int t0 = __builtin_amdgcn_readfirstlane(s_q[0]);
    if (t0 >= ntk) return;
    w_begin = tk_tab[t0]; w_end = tk_tab[t0 + 1];
  } else {
    const int per = (total + gridDim.x - 1) / gridDim.x;
    w_begin = blockIdx.x * per; w_end = min(total, w_begin + per);
    if (w_begin >= w_end) return;
  }
  const float* s_bias = (const float*)(smem + D0_BIAS);

  // ---- staging plans.  Halo tile: slot e_k = tid + 512 k -> row (tid >> 2) + 128 k (k = 0..5; the last 8 rows: 32 lanes of wave 0).
  const int r0 = tid >> 2;
  const int q8 = ((tid & 3) ^ ((r0 >> 1) & 2)) * 8;
  const bool in5 = r0 + 640 < UD_INROWS;
  // low tile: slot e_k = tid + 512 k (k = 0..2) -> (chunk, row, 16-byte slot); 1 440 slots
  int l_row[3], l_src[3];      // row inside the chunk (-1: padding slot), element offset of (chunk plane, swizzled channel slot)
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int e = tid + 512 * k;
    const int cc = e >= D0_LOWROWS * 4 ? 1 : 0, rem = e - cc * D0_LOWROWS * 4;
    l_row[k] = e < 2 * D0_LOWROWS * 4 ? rem >> 2 : -1;
    l_src[k] = cc * (H2 * W2 * 32) + (((rem & 3) ^ ((l_row[k] >> 1) & 2)) << 3);
  }

  // fragment read offsets of the conv (bytes inside a halo buffer / the weight image)
  int xoff[2][3];
  {
    const int rowbase = rg * 4 * UD_PW + xh * 16 + li;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) xoff[sp][dx] = UB_OFF(rowbase + sp * UD_PW + dx, lk) * 2;
  }
  const int woff = UB_OFF(li, lk) * 2;

  int i_tx, i_ty, i_img;      // item being staged
  auto decode = [&](int w) { i_tx = w % tiles_x; w /= tiles_x; i_ty = w % tiles_y; i_img = w / tiles_y; };
  int pixoff[6], lowoff[3];
  auto item_lane_setup = [&]() {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int r = r0 + 128 * k;
      const int py = r / UD_PW, px = r - py * UD_PW;
      const int gx = i_tx * 32 + px - 1, gy = i_ty * 16 + py - 1;
      const bool ok = px < 34 && gx >= 0 && gx < W && gy >= 0 && gy < H;
      pixoff[k] = ok ? gy * W + gx : -1;
    }
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int lr = l_row[k] / 18, lc = l_row[k] - lr * 18;
      const int ly = i_ty * 8 - 1 + lr, lx = i_tx * 16 - 1 + lc;
      lowoff[k] = (l_row[k] >= 0 && ly >= 0 && ly < H2 && lx >= 0 && lx < W2) ? (ly * W2 + lx) * 32 : -1;
    }
  };
  // the DMA pieces are issued one at a time between groups of MFMAs (a run of them back to back stalls the wave on the
  // vector-memory issue port while the matrix pipe idles)
  const unsigned lds_w = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(ud_lptr)smem + (unsigned)(wave * 1024));      // this wave's 1 KB of a piece
  auto skip_piece = [&](int k) {      // piece k of 6 -> halo buffer A (k is a compile-time constant at every call site)
    const ET* simg = skip + (size_t)i_img * H * W * 32;
    const ET* p = pixoff[k] >= 0 ? simg + (unsigned)(pixoff[k] * 32 + q8) : zero_page;
    if (k < 5 || in5) ud_dma16(lds_w + k * 8192, p);
  };
  auto low_piece = [&](int k) {      // piece k of 3 -> low tile
    const ET* limg = low + (size_t)i_img * H2 * W2 * 64;
    const ET* p = lowoff[k] >= 0 ? limg + (unsigned)(lowoff[k] + l_src[k]) : zero_page;
    ud_dma16(lds_w + D0_LOW + k * 8192, p);
  };
  // one conv step: 9 taps x (4 pixel rows x 2 cout tiles) MFMAs on halo buffer `sb`, weight chunk `cc`
  f32x4 acc[4][2];
  auto conv_step = [&](const unsigned char* sb, int cc, int pieces /*0 none, 1 the low tile's, 2 all nine*/) {
    const unsigned char* wbp = smem + D0_WCONV + cc * 9 * 32 * 64 + woff;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      v8 xq[6];
#pragma unroll
      for (int s = 0; s < 6; ++s) xq[s] = *(const v8*)(sb + xoff[s & 1][dx] + (s & ~1) * UD_PW * 64);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int tap = dy * 3 + dx;
        v8 wf[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) wf[n] = *(const v8*)(wbp + (tap * 32 + n * 16) * 64);
        if (pieces == 1 && dx == 0) low_piece(dy);      // (taps 0, 3, 6: the first three MFMA groups of the step)
        if (pieces == 2) { const int slot = dx * 3 + dy; if (slot < 6) skip_piece(slot); else low_piece(slot - 6); }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < 2; ++n) acc[m][n] = E16<ET>::mfma(wf[n], xq[m + dy], acc[m][n]);
      }
    }
  };
  // the up-conv half of the halo tile: 4 phases x 10 groups of 16 pixels (9 x 17 per phase).  A wave pair owns a phase, so its
  // four weight fragments (2 chunks x 2 cout tiles) stay in registers for the whole launch; five groups per wave
  const int ph = wave >> 1, pdy = ph >> 1, pdx = ph & 1;
  v8 uw[2][2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
#pragma unroll
    for (int n = 0; n < 2; ++n) uw[kc][n] = *(const v8*)(smem + D0_WUP + ((ph * 2 + kc) * 32 + n * 16) * 64 + woff);
  auto upconv_into = [&](unsigned char* bbuf, int x0, int y0, bool skip_next) {
#pragma unroll      // (five independent chains per wave: their LDS reads, MFMAs and writes interleave)
    for (int u = 0; u < 5; ++u) {
      const int gi = (wave & 1) * 5 + u;
      const int t = 16 * gi + li;
      const bool valid = t < 153;
      const int a = valid ? t / 17 : 0, bc = valid ? t - a * 17 : 0;
      const int lowrow = (a + (pdy == 0 ? 1 : 0)) * 18 + bc + (pdx == 0 ? 1 : 0);
      if (skip_next) { skip_piece(u); if (u == 4) skip_piece(5); }
      f32x4 ua[2];
#pragma unroll
      for (int n = 0; n < 2; ++n) ua[n] = *(const f32x4*)(s_bias + 32 + 8 * lk + 4 * n);
#pragma unroll
      for (int kc = 0; kc < 2; ++kc) {
        const v8 xf = *(const v8*)(smem + D0_LOW + kc * D0_LOWCH + UB_OFF(lowrow, lk) * 2);
#pragma unroll
        for (int n = 0; n < 2; ++n) ua[n] = E16<ET>::mfma(uw[kc][n], xf, ua[n]);
      }
      const int py = 2 * a + (pdy == 0 ? 1 : 0), px = 2 * bc + (pdx == 0 ? 1 : 0);
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
      v8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = inside ? (ET)ua[r >> 2][r & 3] : (ET)0.0f;      // outside the image: the conv's zero padding
      if (valid) *(v8*)(bbuf + UB_OFF(py * UD_PW + px, lk) * 2) = o;
    }
  };

  decode(w_begin);
  item_lane_setup();
#pragma unroll
  for (int k = 0; k < 6; ++k) skip_piece(k);
#pragma unroll
  for (int k = 0; k < 3; ++k) low_piece(k);
  bool stores_in_flight = false;
  for (int w = w_begin;;) {
    const int c_x0 = i_tx * 32, c_y0 = i_ty * 16, c_img = i_img;
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const f32x4 bv = *(const f32x4*)(s_bias + 8 * lk + 4 * n);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m][n] = bv;
    }
#if defined(SH_D0_VARIANT) && SH_D0_VARIANT == 2
    // two barriers per item: step 0 and the up-conv in one phase, all nine DMA pieces of the next item during step 1
    if (stores_in_flight) ud_wait_vm<NSTORE>();
    else ud_wait_vm<0>();
    stores_in_flight = false;
    __builtin_amdgcn_s_barrier();      // skip tile (A) and low tile (L) of this item have landed
    conv_step(smem, 0, 0);
    bool more = true;
    if (w + 1 < w_end) { ++w; if (++i_tx == tiles_x) { i_tx = 0; if (++i_ty == tiles_y) { i_ty = 0; ++i_img; } } }
    else if (dyn) {
      const int nt = __builtin_amdgcn_readfirstlane(s_q[qk]);
      if (nt < ntk) {
        if (tid == 0) s_q[qk ^ 1] = ud_take_ticket(ticket, ntk, yl);
        qk ^= 1;
        w = tk_tab[nt]; w_end = tk_tab[nt + 1];
        decode(w);
      } else more = false;
    } else more = false;
    if (more) item_lane_setup();
    upconv_into(smem + D0_A1, c_x0, c_y0, false);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();      // B is complete; A and L are free
    conv_step(smem + D0_A1, 1, more ? 2 : 0);
#else
    // in flight, oldest first: this item's skip pieces (6), its low pieces (3), the previous item's epilogue stores (NSTORE)
    if (stores_in_flight) ud_wait_vm<3 + NSTORE>();
    else ud_wait_vm<3>();
    __builtin_amdgcn_s_barrier();      // the skip tile (A) of this item has landed
    conv_step(smem, 0, 0);
    if (stores_in_flight) ud_wait_vm<NSTORE>();
    else ud_wait_vm<0>();
    stores_in_flight = false;
    __builtin_amdgcn_s_barrier();      // every wave is done reading A; the low tile (L) has landed
    // the next item (fixed share or next ticket), its skip tile -> A
    bool more = true;
    if (w + 1 < w_end) { ++w; if (++i_tx == tiles_x) { i_tx = 0; if (++i_ty == tiles_y) { i_ty = 0; ++i_img; } } }
    else if (dyn) {
      const int nt = __builtin_amdgcn_readfirstlane(s_q[qk]);
      if (nt < ntk) {
        if (tid == 0) s_q[qk ^ 1] = ud_take_ticket(ticket, ntk, yl);
        qk ^= 1;
        w = tk_tab[nt]; w_end = tk_tab[nt + 1];
        decode(w);
      } else more = false;
    } else more = false;
    if (more) item_lane_setup();
    upconv_into(smem + D0_A1, c_x0, c_y0, more);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // this wave's part of B is written
    __builtin_amdgcn_s_barrier();      // B is complete, every wave is done reading L
    conv_step(smem + D0_A1, 1, more ? 1 : 0);
#endif
    ET* out = dst + (size_t)c_img * H * W * 32;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int gy = c_y0 + rg * 4 + m, gx = c_x0 + xh * 16 + li;
      v8 o;
#pragma unroll
      for (int r = 0; r < 8; ++r) o[r] = (ET)fmaxf(acc[m][r >> 2][r & 3], 0.0f);
      ud_store16(out + ((size_t)gy * W + gx) * 32 + 8 * lk, o);
    }
    stores_in_flight = true;
    if (!more) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace sh
