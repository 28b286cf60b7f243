// k_unet16_dec0b3.h -- dec0b (32 -> 32 channels, 3x3) + the 1x1 head with THREE halo buffers: the tile two items ahead is in
// flight while this one is multiplied.
//
// k_conv3_dma16<EK, UF_HEAD, 2, 1, 2> (k_unet_bf16_dma.h) keeps one 41 KB tile in flight per CU: the DMA of item i + 1 is issued
// behind item i's barrier and waited for in front of item i + 1's.  What such a kernel reads per second is (bytes in flight) /
// (memory latency), and the latency is not the kernel's own: beside the other lane's geometry chain every GB that chain moves costs
// the UNet pass ~0.28 ms.  The layer has one 32-channel chunk, its weights live in registers (18 fragments),
// so LDS holds halo tiles only -- three of them here (3 x 48 KB), two in flight at any time.  Same items, same tickets, same
// arithmetic in the same order as the two-buffer kernel: the logits are bit-identical (tests/test_gpu_unet_bf16.py).
//
// Every wave issues all six 1 KB-per-wave DMA pieces of a tile (the rows behind the 648 halo rows read the zero page into padding),
// and a step without an item two ahead issues them from the zero page as well: the vector-memory operations of every wave and
// step are then 6 loads + 4 stores in a fixed order, and the counted wait in front of a step's barrier is exact:
//   issued, oldest first:  pieces(t)  stores(t-2)  pieces(t+1)  stores(t-1)   ->  s_waitcnt vmcnt(14) leaves the younger 14 in flight
//   (t = 0: pieces(0) pieces(1) -> vmcnt(6);  t = 1: pieces(1) pieces(2) stores(0) -> vmcnt(10))
#pragma once
#include <type_traits>
#include "k_unet_bf16_dma.h"

namespace sh {

#define D3_BUFB (6 * 8192)                  // 49 152: six pieces of 8 KB (648 halo rows of 64 B = 41 472 used)
#define D3_BIAS (3 * D3_BUFB)               // 147 456: bias[32] | head_w[32]
#define D3_SMEM (D3_BIAS + 256)
#define D3_WSTAGE (2 * D3_BUFB)             // the weight image [9 taps][32 rows] of 64 B sits in buffer 2 until the fragments are in registers

template <int EK>
__global__ void __launch_bounds__(UD_THREADS)
k_dec0b_head3(const u16* __restrict__ src_ /*[img][H W][32]*/, const u16* __restrict__ wgt_ /*packed [9][1][32][32]*/, const float* __restrict__ bias,
              const float* __restrict__ head_w, const float* __restrict__ head_b, float* __restrict__ logits, int H, int W, int nimg,
              const u16* __restrict__ zero_page_, unsigned* __restrict__ ticket, const int* __restrict__ tk_tab, int ntk, YieldArg yl) {
  using ET = typename EKT<EK>::type;
  using v8 = typename E16<ET>::v8;
  const ET* src = (const ET*)src_;
  const ET* wgt = (const ET*)wgt_;
  const ET* zero_page = (const ET*)zero_page_;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[D3_SMEM];
  __shared__ int s_q[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int xh = wave & 1, rg = wave >> 1;
  const int tiles_x = W / 32, tiles_y = H / 16;
  if (tid == 0) { s_q[0] = ud_take_ticket(ticket, ntk, yl); s_q[1] = ud_take_ticket(ticket, ntk, yl); }

  // ---- once per workgroup: biases, the weight fragments of this lane (row dealing and slot swizzle of k_conv3_dma16, NN = 2)
  float* s_bias = (float*)(smem + D3_BIAS);
  if (tid < 32) s_bias[tid] = bias[tid];
  else if (tid < 64) s_bias[tid] = head_w[tid - 32];
  for (int e = tid; e < 9 * 32 * 4; e += UD_THREADS) {
    const int row = e >> 2, q = e & 3;
    const int tap = row >> 5, j = row & 31;
    const int ch = 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);
    *(u32x4*)(smem + D3_WSTAGE + e * 16) = *(const u32x4*)(wgt + (size_t)(tap * 32 + ch) * 32 + ((q ^ ((row >> 1) & 2)) << 3));
  }
  __syncthreads();
  v8 wreg[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int n = 0; n < 2; ++n) wreg[tap][n] = *(const v8*)(smem + D3_WSTAGE + (tap * 32 + n * 16) * 64 + UB_OFF(lane & 15, lane >> 4) * 2);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __syncthreads();      // every wave holds its fragments: buffer 2 is free; every ordinary load is retired before the first LDS-DMA

  const int t0 = __builtin_amdgcn_readfirstlane(s_q[0]);
  if (t0 >= ntk) return;
  int qk = 1;
  // staging cursor: the item whose tile is issued next (two ahead of the one being multiplied)
  int s_w = tk_tab[t0], s_wend = tk_tab[t0 + 1], s_tx, s_ty, s_img;
  bool s_live = true;
  auto decode = [&](int w) { s_tx = w % tiles_x; w /= tiles_x; s_ty = w % tiles_y; s_img = w / tiles_y; };
  decode(s_w);
  auto advance = [&]() {      // to the next item of the ticket, or the first of the next ticket (its id was written at least a barrier ago)
    if (s_w + 1 < s_wend) { ++s_w; if (++s_tx == tiles_x) { s_tx = 0; if (++s_ty == tiles_y) { s_ty = 0; ++s_img; } } return; }
    const int nt = __builtin_amdgcn_readfirstlane(s_q[qk]);
    if (nt < ntk) {
      if (tid == 0) s_q[qk ^ 1] = ud_take_ticket(ticket, ntk, yl);
      qk ^= 1;
      s_w = tk_tab[nt]; s_wend = tk_tab[nt + 1];
      decode(s_w);
    } else s_live = false;
  };
  // the tile of the cursor's item (or zeros) -> buffer `bf`: six pieces per wave, slot e_k = tid + 512 k -> LDS row (tid >> 2) + 128 k
  const int r0 = tid >> 2;
  const int q8 = ((tid & 3) ^ ((r0 >> 1) & 2)) * 8;
  auto stage = [&](int bf) {
    const ET* simg = src + (size_t)s_img * H * W * 32;
    unsigned char* lb = smem + bf * D3_BUFB + wave * 1024;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int r = r0 + 128 * k;
      const int py = r / UD_PW, px = r - py * UD_PW;
      const int gx = s_tx * 32 + px - 1, gy = s_ty * 16 + py - 1;
      const bool ok = s_live && r < UD_INROWS && px < 34 && gx >= 0 && gx < W && gy >= 0 && gy < H;
      const ET* p = ok ? simg + (unsigned)((gy * W + gx) * 32 + q8) : zero_page;
      __builtin_amdgcn_global_load_lds((ud_gptr)p, (ud_lptr)(lb + k * 8192), 16, 0, 0);
    }
  };
  // fragment read offsets (bytes inside a buffer)
  int xoff[2][3];
  {
    const int rowbase = rg * 4 * UD_PW + xh * 16 + li;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) xoff[sp][dx] = UB_OFF(rowbase + sp * UD_PW + dx, lk) * 2;
  }

  // ---- prologue: items 0 and 1 -> buffers 0 and 1
  int c_x0[3], c_y0[3], c_img[3];      // the items of the three buffers
  bool c_live[3];
#pragma unroll
  for (int k = 0; k < 2; ++k) {
    c_x0[k] = s_tx * 32; c_y0[k] = s_ty * 16; c_img[k] = s_img; c_live[k] = s_live;
    stage(k);
    if (s_live) advance();
    // a ticket id is read a barrier after it was written (in the loop: the step's own barrier); with one-item tickets both of
    // these advances fetch one
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  const float hb = head_b[0];
  int t = 0;
  // one item: the buffer index is a compile-time constant inside (the item arrays stay in registers)
  auto step = [&](auto BF) -> bool {
    {
      constexpr int bf = decltype(BF)::value;
      if (!c_live[bf]) return false;
      if (t == 0) ud_wait_vm<6>();
      else if (t == 1) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // every wave's pieces of this item have landed; every wave is done with the buffer staged next
      {
        constexpr int nb = (bf + 2) % 3;
        c_x0[nb] = s_tx * 32; c_y0[nb] = s_ty * 16; c_img[nb] = s_img; c_live[nb] = s_live;
        stage(nb);      // (zeros when there is no such item: the count of operations per step stays fixed)
        if (s_live) advance();
      }
      const unsigned char* sb = smem + bf * D3_BUFB;
      f32x4 acc[4][2];
#pragma unroll
      for (int n = 0; n < 2; ++n) {
        const f32x4 bv = *(const f32x4*)(s_bias + 8 * lk + 4 * n);
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m][n] = bv;
      }
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        v8 xq[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) xq[s] = *(const v8*)(sb + xoff[s & 1][dx] + (s & ~1) * UD_PW * 64);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = E16<ET>::mfma(wreg[dy * 3 + dx][n], xq[m + dy], acc[m][n]);
      }
      // logit = head_b + sum over the 32 channels of relu(conv) * head_w: 8 in the lane, the rest in lanes li + 16 k (the order of k_conv3_dma16)
      float* lo = logits + (size_t)c_img[bf] * H * W;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float sacc = 0.0f;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc = __builtin_fmaf(fmaxf(acc[m][n][r], 0.0f), s_bias[32 + 8 * lk + 4 * n + r], sacc);
        sacc += __shfl_xor(sacc, 16);
        sacc += __shfl_xor(sacc, 32);
        ud_store4(lo + (size_t)(c_y0[bf] + rg * 4 + m) * W + c_x0[bf] + xh * 16 + li, hb + sacc);      // (all four lanes of a pixel store: the count stays exact)
      }
      ++t;
    }
    return true;
  };
#pragma unroll 1
  for (;;) {
    if (!step(std::integral_constant<int, 0>{})) break;
    if (!step(std::integral_constant<int, 1>{})) break;
    if (!step(std::integral_constant<int, 2>{})) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace sh
