// k_unet16_up.h -- 2x2 stride-2 transposed convolutions (the decoder's up-sampling) of the 16-bit UNet.
#pragma once
#include "k_unet_bf16.h"
#include "k_unet16_base.h"

namespace sh {

// ---- 2x2 stride-2 transposed convolution (the decoder's up-sampling), both column phases of an output row per workgroup --
// out[2y + dy][2x + dx][co] = b[co] + sum_ci in[y][x][ci] * w[dy * 2 + dx][ci][co]: per output pixel one K = Cin product, no
// spatial reuse -- a memory-bound layer (up0 at B = 64: 0.54 GB in, 1.07 GB out, 0.07 TFLOP).  The two-barrier kernel ran
// it as one launch of 4 x Cout/64 workgroups per tile: every workgroup re-read the input tile, and the two dx phases of an
// output row -- the two 64-byte halves of each 128-byte line -- were written by different workgroups.  Here a workgroup
// (4 waves) owns a 16x16 source tile x 32 output channels x the two dx phases of one row parity dy: the tile is staged
// once per 32-channel chunk and multiplied against both phases' weights (pixel fragments reused), and both halves of
// every output line leave the same wave back to back.  Channels are dealt to the MFMA rows as in k_conv3_ldr16 (16-byte stores).
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
// Here a workgroup (8 waves) owns a 32 x 4 MT source tile for ALL 4 Cout outputs of its pixels: every wave loads its MT rows x 16
// pixels x Cin once, straight into the MFMA pixel fragments (MT x NCH fragments: 64 / 128 / 128 VGPRs for Cin = 128 / 256 / 512 with
// MT = 4 / 4 / 2), and keeps them for the whole launch; only the weights go through LDS.  Round 3-4's form of it (k_upconv16r)
// synchronised its eight waves after every (32-cout group, phase) slice -- 32 or 64 MFMAs per wave between two barriers, the next
// slice's weights through registers, per-element conversions: matrix pipe busy 0.15-0.23, waves parked at the barrier a third of
// the time.  Here the weights of PPB phases of a group (all four for Cin = 128 / 256; two for Cin = 512, whose 4 x 512 rows would
// not fit twice) are one LDS-DMA transfer (inline assembly, the slot swizzle on the source address) that lands while the previous
// group is multiplied; one barrier per PPB x MT x 2 x NCH MFMAs of a wave; the conversions in pairs; stores as assembly, so that the
// counted wait in front of the barrier retires exactly the transfer (the PPB x MT stores behind it may still be in flight).
// Input bytes come from HBM exactly once, the two 64-byte halves of an output line (dx = 0, 1) leave the same wave in consecutive
// phases.  Same accumulation (bias in the accumulator, chunks in order) and the same channel dealing as k_upconv16: the same tensor
// bit for bit.  Alone at B = 64: up3 0.115 -> 0.078 ms, up2 0.126 -> 0.111, up1 0.188 -> 0.186 (the 0.54 GB it writes).
#define UPR_THREADS 512
template <int N> __device__ inline void up_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }

template <int EK, int NCH, int MT, int PPB, bool PERSIST /*false: one workgroup per item, item = blockIdx.x (no second set of pixel registers)*/>
__global__ void __launch_bounds__(UPR_THREADS)
k_upconv16g(const u16* __restrict__ src_, const u16* __restrict__ wgt_ /*packed [4][NCH][Cout][32]*/, const float* __restrict__ bias,
            u16* __restrict__ dst_, int H, int W, int Cout, int nimg, unsigned* __restrict__ ticket, const int* __restrict__ tk_tab, int ntk) {
  using ET = typename EKT<EK>::type;
  using v8 = typename E16<ET>::v8;
  using v2 = typename E16<ET>::v2;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const ET* wgt = (const ET*)wgt_;
  constexpr int WROWS = NCH * 32;                              // LDS rows of one phase's slice: [chunk][16 n + i]
  constexpr int NPC = PPB * WROWS * 4 / UPR_THREADS;           // 16-byte pieces per thread and transfer
  static_assert(PPB * WROWS * 4 % UPR_THREADS == 0 && PPB * MT <= 32 && PPB % 2 == 0, "whole pieces; the counted wait; whole row parities");
  constexpr int BUFB = PPB * WROWS * 64;                       // bytes of a weight buffer
  // (at least 84 KB: ONE workgroup per CU, like the persistent convolutions -- two per CU spread over all 256 CUs and the reserve is gone)
  __shared__ __attribute__((aligned(1024))) unsigned char s_w[2 * BUFB > 84 * 1024 ? 2 * BUFB : 84 * 1024];
  __shared__ __attribute__((aligned(16))) float s_b[512];
  __shared__ int s_q[2];
  // A workgroup walks items (image, 32 x 4 MT source tile) handed out by work tickets as in k_conv3_ldr16: launched on the grid of the
  // persistent convolutions (the CUs a lane's UNet pass owns while another lane's geometry runs on the reserve), the kernel does not
  // contend with that lane's workgroups for CUs -- as one workgroup per item it took 1.5-1.7 x its time alone inside the two-lane
  // region, the persistent kernels 1.04-1.24 x.  The pixels of the next item are requested at the start of this one (a second set of
  // fragment registers); the weight transfers run on across the items (the first group's weights of the next item land during the
  // last group of this one); when a layer's weights fit the two buffers they are fetched once.
  const int tiles_x = W / 32, tiles = tiles_x * (H / (4 * MT)), nitems = tiles * nimg;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  if (tid < Cout) s_b[tid] = bias[tid];
  const int li = lane & 15, lk = lane >> 4;
  const int xh = wave & 1, rg = wave >> 1;
  const int groups = Cout >> 5;
  constexpr int NT = 4 / PPB;                                  // transfers per group
  const int ntr = groups * NT;
  // transfer u = (group g = u / NT, phases PPB (u % NT) ..): piece e = tid + 512 k -> phase p = e / (4 WROWS), row r = 32 cc + j, LDS slot e & 3;
  // the LDS position is linear in e (1 KB per wave and piece), the slot swizzle of UB_OFF is applied to the SOURCE slot
  unsigned wsrc[NPC];
#pragma unroll
  for (int k = 0; k < NPC; ++k) {
    const int e = tid + k * UPR_THREADS, ph = e / (4 * WROWS), rem = e - ph * 4 * WROWS, r = rem >> 2, cc = r >> 5, j = r & 31;
    const int ch = 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);      // LDS row 16 n + i holds channel 8 (i >> 2) + 4 n + (i & 3) of the group
    wsrc[k] = (unsigned)(((ph * NCH + cc) * Cout + ch) * 32 + (((rem & 3) ^ ((r >> 1) & 2)) << 3));
  }
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(ud_lptr)s_w) + (unsigned)(wave * 1024);
  auto fetch_to = [&](int u, int bufi) __attribute__((always_inline)) {      // transfer u of an item -> weight buffer bufi
    const ET* wsl = wgt + ((size_t)((u % NT) * PPB) * NCH * Cout + (size_t)(u / NT) * 32) * 32;
    const unsigned lb = lds0 + (unsigned)(bufi * BUFB);
#pragma unroll
    for (int k = 0; k < NPC; ++k) ud_dma16(lb + k * 8192, wsl + wsrc[k]);
  };
  auto fetch = [&](int u) __attribute__((always_inline)) { fetch_to(u, u & 1); };
  const bool resident = ntr <= 2;      // the layer's weights fit the two buffers: fetched once
  if (PERSIST && tid == 0) { s_q[0] = ud_take_ticket(ticket); s_q[1] = ud_take_ticket(ticket); }
  fetch(0);
  if (resident && ntr == 2) fetch(1);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  const int t0 = PERSIST ? __builtin_amdgcn_readfirstlane(s_q[0]) : 0;
  if (PERSIST && t0 >= ntk) return;
  if (!PERSIST && (int)blockIdx.x >= nitems) return;
  int w = PERSIST ? tk_tab[t0] : (int)blockIdx.x, wend = PERSIST ? tk_tab[t0 + 1] : w + 1, qk = 1;      // the item in hand, the end of its ticket, the slot of the next ticket
  int uu = 0;                                             // transfers consumed so far: buffer parity
  const int OW = 2 * W, OH = 2 * H;
  // the wave's pixels of an item: fragment (m, cc) = 8 channels 32 cc + 8 lk .. of pixel (y0 + m, x0)
  auto pixels = [&](int item, v8 (&x)[MT][NCH]) __attribute__((always_inline)) {
    const int img = item / tiles, tl = item - img * tiles;
    const int x0 = (tl % tiles_x) * 32 + xh * 16 + li, y0 = ((tl / tiles_x) * 4 + rg) * MT;
    const ET* in = (const ET*)src_ + (size_t)img * H * W * (NCH * 32);
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) x[m][cc] = *(const v8*)(in + ((size_t)cc * H * W + (size_t)(y0 + m) * W + x0) * 32 + 8 * lk);
  };
  v8 xf[MT][NCH], xn[PERSIST ? MT : 1][PERSIST ? NCH : 1];
  pixels(w, xf);
#pragma unroll 1
  for (;;) {
  const int item = w;
  const int img = item / tiles, tl = item - img * tiles;
  const int x0 = (tl % tiles_x) * 32 + xh * 16 + li, y0 = ((tl / tiles_x) * 4 + rg) * MT;
  // the pixel fragments are USED here, so hipcc's wait for them stands here and not inside the transfer loop
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int cc = 0; cc < NCH; ++cc) { u32x4 t = __builtin_bit_cast(u32x4, xf[m][cc]); asm volatile("" : "+v"(t)); xf[m][cc] = __builtin_bit_cast(v8, t); }
  // (everything of this wave is retired here: the pixels, the previous item's stores, a transfer issued during its last group)
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  // the next item: the next one of this ticket, or the first one of the ticket taken an item ago (lane 0 then takes another)
  bool nlive = PERSIST;
  if constexpr (PERSIST) {
    if (w + 1 < wend) ++w;
    else {
      const int nt = __builtin_amdgcn_readfirstlane(s_q[qk]);
      if (nt < ntk) {
        if (tid == 0) s_q[qk ^ 1] = ud_take_ticket(ticket);
        qk ^= 1;
        w = tk_tab[nt]; wend = tk_tab[nt + 1];
      } else nlive = false;
    }
    if (nlive) pixels(w, xn);      // in flight during this item's groups
  }
  asm volatile("" ::: "memory");

  ET* out0 = (ET*)dst_ + (size_t)img * OH * OW * Cout;
#pragma unroll 1
  for (int u = 0; u < ntr; ++u, ++uu) {
    const int g = u / NT;
    // the next transfer (of this item, or the first one of the next item) lands during the MFMAs below; the last readers of its
    // buffer passed the barrier that ended the previous transfer
    const bool more = !resident && (u + 1 < ntr || nlive);
    if (more) fetch_to(u + 1 < ntr ? u + 1 : 0, (uu + 1) & 1);
    const unsigned char* sw = s_w + ((resident ? u : uu) & 1) * BUFB;
    ET* out = out0 + (size_t)g * OH * OW * 32;      // this group's 32-channel plane
    f32x4 bv[2];
#pragma unroll
    for (int n = 0; n < 2; ++n) bv[n] = *(const f32x4*)(s_b + g * 32 + 8 * lk + 4 * n);
#pragma unroll
    for (int pq = 0; pq < PPB / 2; ++pq) {      // a row parity dy: both column phases together, so that the two 64-byte halves of an output line leave back to back
      const int dy = (u % NT) * (PPB / 2) + pq;
      f32x4 acc[2][MT][2];
#pragma unroll
      for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int m = 0; m < MT; ++m) acc[dx][m][n] = bv[n];
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) {
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          v8 wf[2];
#pragma unroll
          for (int n = 0; n < 2; ++n) wf[n] = *(const v8*)(sw + (size_t)((2 * pq + dx) * WROWS) * 64 + UB_OFF(cc * 32 + n * 16 + li, lk) * 2);
#pragma unroll
          for (int m = 0; m < MT; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[dx][m][n] = E16<ET>::mfma(wf[n], xf[m][cc], acc[dx][m][n]);
        }
      }
#pragma unroll
      for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
          u32x4 o;
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const f32x2 a01 = {acc[dx][m][n][0], acc[dx][m][n][1]}, a23 = {acc[dx][m][n][2], acc[dx][m][n][3]};
            o[2 * n] = __builtin_bit_cast(unsigned, __builtin_convertvector(a01, v2));
            o[2 * n + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(a23, v2));
          }
          ud_store16(out + ((size_t)(2 * (y0 + m) + dy) * OW + 2 * x0 + dx) * 32 + 8 * lk, o);
        }
    }
    // the next transfer has landed (it is older than this iteration's PPB x MT stores)
    if (more) up_wait_vm<PPB * MT>();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
  }
  if (!nlive) break;
  if constexpr (PERSIST) {
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int cc = 0; cc < NCH; ++cc) xf[m][cc] = xn[m][cc];
  }
  }      // items
}

}  // namespace sh
