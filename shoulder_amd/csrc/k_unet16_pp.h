// k_unet16_pp.h -- the full-resolution (32-channel) level of the 16-bit UNet, second generation: the two waves of a SIMD take
// TURNS on the matrix pipe ("ping-pong").
//
// The first generation (k_unet16_l0.h / k_unet16_dec0.h / k_unet16_dec0b3.h) runs 8 waves in lockstep: every wave stages, then
// every wave multiplies, then every wave runs its epilogue, one s_barrier per tile.  Its counters (profiles/r04_pmc_sq_b64_bf16.json):
// matrix pipe busy 0.27-0.39, waves parked 0.40-0.46 -- while the 8 waves issue LDS-DMA pieces or convert, round and store a tile,
// no wave of the CU has an MFMA to issue, and the two waves of a SIMD reach that state together.
// Here a workgroup is two GROUPS of four waves (waves w and w + 4 share a SIMD: one wave of each group per SIMD) that run half a
// period apart.  In phase p group (p & 1) is ON: 144 MFMAs per wave on tile p (8 rows x 16 pixels x 32 couts, the whole 32 x 16
// tile for the group), nothing else but its 30 fragment reads; the other group is OFF: it issues the LDS-DMA pieces of tile p + 2
// (the ON group of phase p + 2 is the group that is ON now) and runs the epilogue of the tile it multiplied in phase p - 1.  One
// s_barrier per phase for the whole workgroup.  Three halo buffers: tile p is read in phase p, tile p + 1 has landed, tile p + 2 is
// in flight.  A wave's vector-memory operations are, per OFF phase and in this order, its pieces and then its epilogue stores, so the
// counted wait in front of the barrier that ends its next ON phase -- vmcnt(number of stores) -- retires exactly the pieces.
//
// Vector work is kept off the matrix waves' issue slots: ReLU is an integer max on the f32 bits (fmaxf is two v_max_f32: it
// quiets NaNs first), conversions are packed (v_cvt_pk_bf16_f32 on pairs), the head's sum over the four lane groups is a
// reduce-scatter on v_permlane16_swap / v_permlane32_swap (6 swaps + 6 adds per 8 rows instead of 16 ds_bpermute + 16 adds, and
// two store instructions per tile instead of eight), interior tiles stage from one scalar base + per-lane offsets computed once.
//
// Work items, tickets, LDS image of a halo tile (648 rows of 64 B at pitch 36, XOR slot swizzle on the DMA's source address and on
// the fragment read), channel dealing and the order of every sum are those of the first generation: dec0b + head returns the
// same logits bit for bit (tests/test_gpu_unet_bf16.py).
#pragma once
#include <type_traits>
#include "k_unet16_base.h"

namespace sh {

#define PP_THREADS 512
#define PP_GTHREADS 256                      // lanes of a group (4 waves)
#define PP_NPIECE 11                         // LDS-DMA pieces per staging wave and tile: 256 lanes x 11 = 2816 slots of 16 B >= 648 rows x 4
#define PP_BUFB (PP_NPIECE * 4096)           // 45 056 bytes per halo buffer: piece k of staging wave v at k * 4096 + v * 1024
#ifndef PP_K1
#define PP_K1 11                             // pieces 0 .. PP_K1 - 1 of a tile are issued by the OFF group, the rest by the ON group behind its MFMAs
#endif
#ifndef PP_G1PRIO
#define PP_G1PRIO 0                          // static s_setprio of the second-dispatched group (waves 4 .. 7: the loser of every age-based arbitration)
#endif
#ifndef PP_OFFPRIO
#define PP_OFFPRIO 0                         // s_setprio of a wave in its OFF phase
#endif

template <int N> __device__ inline void pp_wait_vm() {      // s_waitcnt takes an immediate
  if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else if constexpr (N == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
  else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
  else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  else if constexpr (N == 11) asm volatile("s_waitcnt vmcnt(11)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 13) asm volatile("s_waitcnt vmcnt(13)" ::: "memory");
  else static_assert(N < 0, "pp_wait_vm: add the immediate");
}
// wait until at most nb + ns of this wave's vector-memory operations are outstanding (nb in {0, NB}, ns in {0, NS})
template <int NB, int NS> __device__ inline void pp_wait_vm2(bool b, bool s) {
  if (b) { if (s) pp_wait_vm<NB + NS>(); else pp_wait_vm<NB>(); }
  else { if (s) pp_wait_vm<NS>(); else pp_wait_vm<0>(); }
}

__device__ inline float pp_relu(float x) {      // max(x, +0) for every x that is not a NaN: one v_max_i32 (negative floats are negative ints)
  const int i = __builtin_bit_cast(int, x);
  return __builtin_bit_cast(float, i > 0 ? i : 0);
}
// x' = [x0 y0 x2 y2], y' = [x1 y1 x3 y3] by 16-lane rows; returns x' + y': rows 0 and 2 hold x summed over the row pairs (0,1) / (2,3), rows 1 and 3 hold y
__device__ inline float pp_swap16_add(float x, float y) {
  asm volatile("s_nop 1\n\tv_permlane16_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  return x + y;
}
// x' = [x.lo y.lo], y' = [x.hi y.hi] by 32-lane halves; returns x' + y': the low half holds x summed over both halves, the high half y
__device__ inline float pp_swap32_add(float x, float y) {
  asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 1" : "+v"(x), "+v"(y));
  return x + y;
}

// Diagnostic build (-DPP_STAMP; never the product): cycles a wave spends per part of a phase, summed per workgroup and wave into
// pp_stamp[blockIdx][wave][part] (s_memtime ticks = shader cycles; cdna_hip_programming.md section 7, in-kernel stamps).
#ifdef PP_STAMP
#define PP_NSTAMP 8      // 0 ON multiply, 1 ON vm wait, 2 ON barrier, 3 OFF stage, 4 OFF epilogue, 5 OFF barrier, 6 phases, 7 whole loop
__device__ unsigned long long pp_stamp[3 * 256 * 8 * PP_NSTAMP];      // [kernel: 0 dec0b + head, 1 enc0, 2 dec0a][workgroup][wave][part]
#define PP_T(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define PP_ADD(i, a, b) st_[i] += (b) - (a)
#else
#define PP_T(v)
#define PP_ADD(i, a, b)
#endif

// a value loaded from global memory in front of the main loop is USED here, so the compiler's wait for it stands here and not at
// its first use inside the loop (where it would be a vmcnt(0) behind the loop's own LDS-DMA pieces)
template <typename T> __device__ inline void pp_settle(T& v) { asm volatile("" : "+v"(v)); }

// the tile cursor every wave of a workgroup keeps in step: items of the current ticket, the next ticket's id one ticket ahead in LDS
struct PpCursor {
  int w, wend, tx, ty, img, qk;
  bool live;
};

// dec0b (32 -> 32 channels, 3x3, ReLU) + the 1x1 head: only the logits leave the kernel (anatomic_neck.py:67-76, the network's last two layers)
template <int EK>
__global__ void __launch_bounds__(PP_THREADS)
k_dec0b_head_pp(const u16* __restrict__ src_ /*[img][H W][32]*/, const u16* __restrict__ wgt_ /*packed [9][1][32][32]*/, const float* __restrict__ bias,
                const float* __restrict__ head_w, const float* __restrict__ head_b, float* __restrict__ logits, int H, int W, int nimg,
                const u16* __restrict__ zero_page_, unsigned* __restrict__ ticket, const int* __restrict__ tk_tab, int ntk) {
  using ET = typename EKT<EK>::type;
  using v8 = typename E16<ET>::v8;
  const ET* src = (const ET*)src_;
  const ET* wgt = (const ET*)wgt_;
  const ET* zero_page = (const ET*)zero_page_;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[3 * PP_BUFB];
  __shared__ int s_q[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2), wv = wave & 3;
  const int xh = wv & 1, rg8 = wv >> 1;
  const int tiles_x = W / 32, tiles_y = H / 16;
  if (tid == 0) { s_q[0] = ud_take_ticket(ticket); s_q[1] = ud_take_ticket(ticket); }

  // ---- once per workgroup: the weight fragments of this lane (row dealing and slot swizzle of k_conv3_dma16, NN = 2) through buffer 2
  for (int e = tid; e < 9 * 32 * 4; e += PP_THREADS) {
    const int row = e >> 2, q = e & 3;
    const int tap = row >> 5, j = row & 31;
    const int ch = 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);
    *(u32x4*)(smem + 2 * PP_BUFB + e * 16) = *(const u32x4*)(wgt + (size_t)(tap * 32 + ch) * 32 + ((q ^ ((row >> 1) & 2)) << 3));
  }
  f32x4 bv[2];
  float hw[8];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) { bv[n][r] = bias[8 * lk + 4 * n + r]; hw[4 * n + r] = head_w[8 * lk + 4 * n + r]; }
  float hb = head_b[0];
#pragma unroll
  for (int n = 0; n < 2; ++n) pp_settle(bv[n]);
#pragma unroll
  for (int i = 0; i < 8; ++i) pp_settle(hw[i]);
  pp_settle(hb);
  __syncthreads();
  const int t0 = __builtin_amdgcn_readfirstlane(s_q[0]);      // (read in front of the next barrier: lane 0 refills this slot in its first advance())
  v8 wreg[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int n = 0; n < 2; ++n) wreg[tap][n] = *(const v8*)(smem + 2 * PP_BUFB + (tap * 32 + n * 16) * 64 + UB_OFF(li, lk) * 2);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();      // every wave holds its fragments: buffer 2 is free; every ordinary load is retired before the first LDS-DMA

  PpCursor cu;
  {
    if (t0 >= ntk) return;
    cu.w = tk_tab[t0]; cu.wend = tk_tab[t0 + 1]; cu.qk = 1; cu.live = true;
    int w = cu.w;
    cu.tx = w % tiles_x; w /= tiles_x; cu.ty = w % tiles_y; cu.img = w / tiles_y;
  }
  // to the next item of the ticket, or the first of the next ticket (its id was written at least a barrier ago).  `fetcher`: the lane
  // that refills the freed slot -- lane 0 of the first wave of the group that is ON (its vector-memory queue is all but empty at the
  // end of an ON phase; an OFF wave would wait for the pieces it has just issued)
  auto advance = [&](const bool fetcher) __attribute__((always_inline)) {
    if (cu.w + 1 < cu.wend) { ++cu.w; if (++cu.tx == tiles_x) { cu.tx = 0; if (++cu.ty == tiles_y) { cu.ty = 0; ++cu.img; } } return; }
    const int nt = __builtin_amdgcn_readfirstlane(s_q[cu.qk]);
    if (nt < ntk) {
      if (fetcher) s_q[cu.qk ^ 1] = ud_take_ticket(ticket);
      cu.qk ^= 1;
      cu.w = tk_tab[nt]; cu.wend = tk_tab[nt + 1];
      int w = cu.w;
      cu.tx = w % tiles_x; w /= tiles_x; cu.ty = w % tiles_y; cu.img = w / tiles_y;
    } else cu.live = false;
  };

  // ---- staging plan of a group: slot e_k = ltid + 256 k -> LDS row (ltid >> 2) + 64 k (row = py * 36 + px), 16-byte slot ltid & 3.
  // The swizzle bit (bit 2 of the row) is the same for every k.  voff[k]: byte offset of the slot's source from the tile's halo
  // origin (pixel (y0 - 1, x0 - 1)); slots behind the halo rows and the two padding columns re-read the origin pixel (never read back).
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(ud_lptr)smem);
  const unsigned wv1024 = __builtin_amdgcn_readfirstlane((unsigned)(wv * 1024));
  const int ltid = tid & (PP_GTHREADS - 1);
  const int r0 = ltid >> 2;
  const int q8 = ((ltid & 3) ^ ((r0 >> 1) & 2)) * 8;
  int voff[PP_NPIECE];
#pragma unroll
  for (int k = 0; k < PP_NPIECE; ++k) {
    const int r = r0 + 64 * k, py = r / UD_PW, px = r - py * UD_PW;
    voff[k] = (r < UD_INROWS && px < 34) ? ((py * W + px) * 32 + q8) * 2 : ((W + 1) * 32 + q8) * 2;
  }
  auto stage = [&](int bf, auto KB, auto KE) __attribute__((always_inline)) {      // pieces KB .. KE - 1 of the cursor's tile -> buffer bf; the four waves of the calling group
    constexpr int kb = decltype(KB)::value, ke = decltype(KE)::value;
#if defined(PP_ABL) && (PP_ABL & 2)      // diagnostic (wrong results): no LDS-DMA
    return;
#endif
    const ET* simg = src + (size_t)cu.img * H * W * 32;
    const unsigned lb = lds0 + (unsigned)(bf * PP_BUFB) + wv1024;
    if (cu.tx > 0 && cu.tx + 1 < tiles_x && cu.ty > 0 && cu.ty + 1 < tiles_y) {      // interior tile: every halo pixel is inside the image
      const ET* base = simg + ((size_t)(cu.ty * 16 - 1) * W + (cu.tx * 32 - 1)) * 32;
#pragma unroll
      for (int k = kb; k < ke; ++k) ud_dma16_s(lb + k * 4096, (unsigned)voff[k], base);
    } else {
#pragma unroll
      for (int k = kb; k < ke; ++k) {
        const int r = r0 + 64 * k, py = r / UD_PW, px = r - py * UD_PW;
        const int gx = cu.tx * 32 + px - 1, gy = cu.ty * 16 + py - 1;
        const bool ok = r < UD_INROWS && px < 34 && gx >= 0 && gx < W && gy >= 0 && gy < H;
        const ET* p = ok ? simg + (unsigned)((gy * W + gx) * 32 + q8) : zero_page;
        ud_dma16(lb + k * 4096, p);
      }
    }
  };
  // fragment read offsets (bytes inside a buffer): rows rg8 * 8 + s (s = 0..9), pixel xh * 16 + li + dx
  int xoff[2][3];
  {
    const int rowbase = rg8 * 8 * UD_PW + xh * 16 + li;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) xoff[sp][dx] = UB_OFF(rowbase + sp * UD_PW + dx, lk) * 2;
  }
  // the two logit stores of a tile: lane group lk holds rows lk and lk + 4 of its 8 after the reduce-scatter
  const unsigned soff0 = (unsigned)(((rg8 * 8 + lk) * W + xh * 16 + li) * 4), soff1 = soff0 + (unsigned)(4 * W * 4);

  // ---- prologue: group 1 stages item 0 (its "phase -2"), group 0 item 1; every wave walks the cursor
  int ax0 = 0, ay0 = 0, aimg = 0;      // item p - 1 (the epilogue of the OFF group)
  int bx0, by0, bimg; bool blive;      // item p
  int cx0, cy0, cimg; bool clive;      // item p + 1
  bool alive = false;
  bx0 = cu.tx * 32; by0 = cu.ty * 16; bimg = cu.img; blive = true;
  using K0 = std::integral_constant<int, 0>;
  using K1 = std::integral_constant<int, PP_K1>;
  using KN = std::integral_constant<int, PP_NPIECE>;
  if (grp == 1) stage(0, K0{}, KN{});
  advance(tid == 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();      // (a ticket id is read a barrier after it was written)
  cx0 = cu.tx * 32; cy0 = cu.ty * 16; cimg = cu.img; clive = cu.live;
  if (grp == 0 && cu.live) stage(1, K0{}, KN{});
  if (cu.live) advance(tid == 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  f32x4 acc[8][2];
#ifdef PP_STAMP
  unsigned long long st_[PP_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0};
  PP_T(tl0_);
#endif
  bool had_stores = false;      // this wave's most recent OFF phase ended with its two logit stores ...
  bool had_off = false;         // ... and began with its PP_K1 pieces
  int bf = 0;                          // buffer of item p
  // one phase of a wave; ON: multiply item p; OFF: stage item p + 2, finish item p - 1.  Returns false when the workgroup is done.
  auto phase = [&](const bool on) __attribute__((always_inline)) -> bool {
    if (!blive && !alive) return false;
    const int bf2 = bf == 0 ? 2 : bf - 1;      // (p + 2) % 3
    PP_T(ta_);
    if (on) {
#if defined(PP_ABL) && (PP_ABL & 1)      // diagnostic (wrong results): no fragment reads, no MFMAs
      if (false) {
#else
      if (blive) {
#endif
        const unsigned char* sb = smem + bf * PP_BUFB;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int m = 0; m < 8; ++m) acc[m][n] = bv[n];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          v8 xq[10];
#pragma unroll
          for (int s = 0; s < 10; ++s) xq[s] = *(const v8*)(sb + xoff[s & 1][dx] + (s & ~1) * UD_PW * 64);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
              for (int n = 0; n < 2; ++n) acc[m][n] = E16<ET>::mfma(wreg[dy * 3 + dx][n], xq[m + dy], acc[m][n]);
        }
      }
      // this wave's pieces of item p + 1 (issued one phase ago in its OFF phase, in front of its stores) have landed
#ifdef PP_STAMP
      asm volatile("" :: "v"(acc[7][1]), "v"(acc[0][0]));
#endif
      PP_T(tb_);
      pp_wait_vm2<0, 2>(false, had_stores);
      if (PP_K1 < PP_NPIECE && cu.live) stage(bf2, K1{}, KN{});      // the ON group's share of item p + 2, behind its MFMAs
      PP_T(tc_);
      PP_ADD(0, ta_, tb_); PP_ADD(1, tb_, tc_);
    } else {
      if (PP_OFFPRIO) __builtin_amdgcn_s_setprio(PP_OFFPRIO);
      had_off = cu.live;
      if (cu.live) stage(bf2, K0{}, K1{});
      PP_T(tb_);
      PP_ADD(3, ta_, tb_);
      had_stores = false;
#if defined(PP_ABL) && (PP_ABL & 4)      // diagnostic (wrong results): no epilogue
      asm volatile("" :: "v"(acc[0][0]), "v"(acc[7][1]));
      if (false) {
#else
      if (alive) {
#endif
        // logit = head_b + sum over the 32 channels of relu(conv) * head_w: 8 in the lane (the fma chain of k_conv3_dma16), then
        // (lane groups 0 + 1) + (lane groups 2 + 3) as a reduce-scatter
        // (the eight rows' chains side by side: hipcc keeps the source order, and one chain alone is 16 dependent instructions)
        float t[8];
#pragma unroll
        for (int m = 0; m < 8; ++m) t[m] = 0.0f;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int m = 0; m < 8; ++m) t[m] = __builtin_fmaf(pp_relu(acc[m][n][r]), hw[4 * n + r], t[m]);
        float u[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) u[j] = pp_swap16_add(t[2 * j], t[2 * j + 1]);      // lane groups with bit 0 clear: row 2 j, set: row 2 j + 1
        const float w0 = pp_swap32_add(u[0], u[1]);      // row 2 (lk >> 1) + (lk & 1) = lk
        const float w1 = pp_swap32_add(u[2], u[3]);      // row 4 + lk
        float* lo = logits + ((size_t)aimg * H + ay0) * W + ax0;
        ud_store4((float*)((char*)lo + soff0), hb + w0);
        ud_store4((float*)((char*)lo + soff1), hb + w1);
        had_stores = true;
      }
      // the pieces this wave issued behind its MFMAs one phase ago (item p + 1) have landed: younger are this phase's pieces and stores
      if (PP_K1 < PP_NPIECE) pp_wait_vm2<PP_K1, 2>(had_off, had_stores);
      if (PP_OFFPRIO) __builtin_amdgcn_s_setprio(0);
      PP_T(tc_);
      PP_ADD(4, tb_, tc_);
    }
    PP_T(td_);
    // every wave: the cursor's item becomes item p + 2
    ax0 = bx0; ay0 = by0; aimg = bimg; alive = blive;
    bx0 = cx0; by0 = cy0; bimg = cimg; blive = clive;
    cx0 = cu.tx * 32; cy0 = cu.ty * 16; cimg = cu.img; clive = cu.live;
    if (cu.live) advance(on && ltid == 0);
    bf = bf == 2 ? 0 : bf + 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PP_T(te_);
    PP_ADD(on ? 2 : 5, td_, te_);
#ifdef PP_STAMP
    st_[6] += 1;
#endif
    return true;
  };
  // (the roles run separate loops: in one shared body hipcc keeps one role's registers alive through the other's branch)
  if (PP_G1PRIO && grp == 1) __builtin_amdgcn_s_setprio(PP_G1PRIO);
  if (grp == 0) {
#pragma unroll 1
    for (;;) { if (!phase(true)) break; if (!phase(false)) break; }
  } else {
#pragma unroll 1
    for (;;) { if (!phase(false)) break; if (!phase(true)) break; }
  }
#ifdef PP_STAMP
  {
    PP_T(tl1_);
    st_[7] = tl1_ - tl0_;
    if (lane == 0 && blockIdx.x < 256)
      for (int i = 0; i < PP_NSTAMP; ++i) pp_stamp[(blockIdx.x * 8 + wave) * PP_NSTAMP + i] = st_[i];
  }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------
// enc0: image -> [enc0a: 1 -> 32, 3x3, ReLU] -> LDS -> [enc0b: 32 -> 32, 3x3, ReLU] -> skip0 (+ 2x2 max pool -> level 1), ping-pong.
//   ON  (group p & 1):  enc0b on the halo tile h(p) of tile p: 144 MFMAs per wave, 30 pixel + 18 weight fragment reads
//   OFF (other group):  the patch loads of tile p + 2 are issued; enc0a of tile p + 1 on the matrix cores (patch(p + 1) -> h(p + 1),
//                       ~10 groups of 16 halo pixels per wave); the epilogue of tile p - 1 (rounded, ReLU'd, stored; pooled); the
//                       patch of tile p + 2 is scaled, split and written to LDS
// Two halo buffers, two patch slots, one s_barrier per phase.  No LDS-DMA here (the input is 4 bytes per pixel): the kernel's
// vector-memory operations are the patch loads and the epilogue's stores.
//
// First conv on the matrix cores: a patch value v is split into its ET high part and ET low part (v - hi); K layout of the MFMA:
// lane group lk = patch row ty (0..2; lk = 3 has zero weights), element j = column offset tx for the high parts (j = 0..2) and
// 4 + tx for the low parts, so ONE MFMA per 16 pixels x 16 channels does what k_enc0_fused16 needed two for.  The 4 + 4 consecutive
// patch values a lane needs are two aligned ds_read_b64 from copy (q & 3) of the high / low patch (4 shifted copies each, where
// k_enc0_fused16 kept 8 + 8 for its ds_read_b128).  Image precision ~2^-17, ET-rounded weights, f32 accumulate from the bias.
#define E0_HB (UD_INROWS * 64)              // 41 472: a halo buffer
#define E0_CS 1472                          // bytes between two shifted copies of a patch part (8 of front padding + 2 x 728, rounded up)
#define E0_PSLOT (8 * E0_CS)                // 11 776: 4 copies of the high part, 4 of the low part
#define E0_POFF (2 * E0_HB)                 // 82 944
#define E0_WOFF (E0_POFF + 2 * E0_PSLOT)    // 106 496: enc0b weights [9 taps][32 rows] of 64 B (in LDS: the OFF phase needs the registers)
#define E0_SMEM (E0_WOFF + 9 * 32 * 64)     // 124 928
#ifndef E0_OFFPRIO
#define E0_OFFPRIO 1                        // s_setprio of a wave in its OFF phase: the OFF side (~500 instructions) is this kernel's critical path (measured: 0.398 -> 0.383 ms)
#endif
#define E0_NGRP 41                          // groups of 16 halo rows (648 = 40.5 x 16)

// groups T0 .. T1 - 1 of a wave's share of the first conv (k_enc0_pp): all fragment reads of the batch, then its MFMAs, then its
// conversions and writes -- one group after the other is one LDS latency + one MFMA latency per group
template <typename ET, int T0, int T1>
__device__ __forceinline__ void e0_first_conv_batch(const unsigned char* fp, unsigned char* op, const typename E16<ET>::v8 (&wA)[2], const f32x4 (&b0v)[2],
                                                     bool interior, int p0 /*16 wv + li*/, int x0, int y0, int H, int W, bool low8 /*li < 8*/) {
  using v8 = typename E16<ET>::v8;
  using v2 = typename E16<ET>::v2;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  constexpr int N = T1 - T0;
  u32x2 fh[N], fl[N];
#pragma unroll
  for (int i = 0; i < N; ++i) { fh[i] = *(const u32x2*)(fp + 128 * (T0 + i)); fl[i] = *(const u32x2*)(fp + 128 * (T0 + i) + 4 * E0_CS); }
  f32x4 a[N][2];
#pragma unroll
  for (int i = 0; i < N; ++i) {
    const u32x4 fq = {fh[i][0], fh[i][1], fl[i][0], fl[i][1]};
#pragma unroll
    for (int n = 0; n < 2; ++n) a[i][n] = E16<ET>::mfma(wA[n], __builtin_bit_cast(v8, fq), b0v[n]);
  }
#pragma unroll
  for (int i = 0; i < N; ++i) {
    u32x4 o;
    const s16x2 z = {0, 0};      // ReLU on the rounded values: a negative ET is a negative int16
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      const f32x2 a01 = {a[i][n][0], a[i][n][1]}, a23 = {a[i][n][2], a[i][n][3]};
      o[2 * n] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(a01, v2)), z));
      o[2 * n + 1] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(a23, v2)), z));
    }
    if (!interior) {      // outside the image: the second conv's zero padding
      const int p = p0 + 64 * (T0 + i), py = p / UD_PW, px = p - py * UD_PW;
      const int gy = y0 - 1 + py, gx = x0 - 1 + px;
      if (!(gy >= 0 && gy < H && gx >= 0 && gx < W)) o = u32x4{0u, 0u, 0u, 0u};
    }
    if (T0 + i < 10 || low8) *(u32x4*)(op + 4096 * (T0 + i)) = o;
  }
}

template <int EK, bool RAW /*the input is the unscaled f64 image + its bounds*/>
__global__ void __launch_bounds__(PP_THREADS)
k_enc0_pp(const float* __restrict__ image, const float* __restrict__ w0 /*[9][32] f32*/, const float* __restrict__ b0 /*[32]*/,
          const u16* __restrict__ wgt_ /*enc0b, packed [9][1][32][32]*/, const float* __restrict__ bias /*[32]*/,
          u16* __restrict__ skip_, u16* __restrict__ pooled_, int H, int W, int nimg,
          const double* __restrict__ raw /*nullable: the UNSCALED image [nimg][H][W] f64 and ...*/,
          const unsigned long long* __restrict__ mm_enc /*... its minimum / complemented maximum per image, encoded (k_anp_rows)*/,
          unsigned* __restrict__ ticket, const int* __restrict__ tk_tab, int ntk) {
  using ET = typename EKT<EK>::type;
  using v8 = typename E16<ET>::v8;
  using v2 = typename E16<ET>::v2;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  const ET* wgt = (const ET*)wgt_;
  ET* skip = (ET*)skip_;
  ET* pooled = (ET*)pooled_;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[E0_SMEM];
  __shared__ int s_q[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2), wv = __builtin_amdgcn_readfirstlane(wave & 3);
  const int xh = wv & 1, rg8 = wv >> 1;
  const int ltid = tid & (PP_GTHREADS - 1);
  const int tiles_x = W / 32, tiles_y = H / 16;
  if (tid == 0) { s_q[0] = ud_take_ticket(ticket); s_q[1] = ud_take_ticket(ticket); }

  // ---- once per workgroup: enc0b weights -> LDS (row dealing and slot swizzle of k_conv3_dma16, NN = 2), biases, first-conv weights
  for (int e = tid; e < 9 * 32 * 4; e += PP_THREADS) {
    const int row = e >> 2, q = e & 3;
    const int tap = row >> 5, j = row & 31;
    const int ch = 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);
    *(u32x4*)(smem + E0_WOFF + e * 16) = *(const u32x4*)(wgt + (size_t)(tap * 32 + ch) * 32 + ((q ^ ((row >> 1) & 2)) << 3));
  }
  f32x4 bv[2], b0v[2];
  v8 wA[2];
#pragma unroll
  for (int n = 0; n < 2; ++n) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { bv[n][r] = bias[8 * lk + 4 * n + r]; b0v[n][r] = b0[8 * lk + 4 * n + r]; }
#pragma unroll
    for (int j = 0; j < 8; ++j) {      // MFMA row li of tile n = channel 8 (li >> 2) + 4 n + (li & 3): lane group lk of the result owns channels 8 lk .. 8 lk + 7
      const float wv0 = w0[(min(lk, 2) * 3 + min(j & 3, 2)) * 32 + 8 * (li >> 2) + 4 * n + (li & 3)];
      wA[n][j] = (lk < 3 && (j & 3) < 3) ? (ET)wv0 : (ET)0.0f;
    }
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    pp_settle(bv[n]); pp_settle(b0v[n]);
    u32x4 t = __builtin_bit_cast(u32x4, wA[n]);
    pp_settle(t);
    wA[n] = __builtin_bit_cast(v8, t);
  }
  __syncthreads();
  const int t0 = __builtin_amdgcn_readfirstlane(s_q[0]);
  __syncthreads();
  // the patch slots start as zeros: a patch writes elements 0 .. 719 of every copy; the last rows' fragments also read the few
  // elements behind them (against zero weights: they must be finite)
  for (int e = tid; e < 2 * E0_PSLOT / 16; e += PP_THREADS) *(u32x4*)(smem + E0_POFF + e * 16) = u32x4{0u, 0u, 0u, 0u};

  PpCursor cu;
  if (t0 >= ntk) return;
  {
    cu.w = tk_tab[t0]; cu.wend = tk_tab[t0 + 1]; cu.qk = 1; cu.live = true;
    int w = cu.w;
    cu.tx = w % tiles_x; w /= tiles_x; cu.ty = w % tiles_y; cu.img = w / tiles_y;
  }
  auto advance = [&](const bool fetcher) __attribute__((always_inline)) {
    if (cu.w + 1 < cu.wend) { ++cu.w; if (++cu.tx == tiles_x) { cu.tx = 0; if (++cu.ty == tiles_y) { cu.ty = 0; ++cu.img; } } return; }
    const int nt = __builtin_amdgcn_readfirstlane(s_q[cu.qk]);
    if (nt < ntk) {
      if (fetcher) s_q[cu.qk ^ 1] = ud_take_ticket(ticket);
      cu.qk ^= 1;
      cu.w = tk_tab[nt]; cu.wend = tk_tab[nt + 1];
      int w = cu.w;
      cu.tx = w % tiles_x; w /= tiles_x; cu.ty = w % tiles_y; cu.img = w / tiles_y;
    } else cu.live = false;
  };

  // ---- the image patch of a tile: rows y0 - 2 .. y0 + 17, columns x0 - 2 .. x0 + 33 (zero outside the image): 720 values
  // e[row * 36 + col], three per lane of the loading group (e = ltid + 256 k).  From the unscaled image the values stay doubles while
  // the loads are in flight; patch_store applies X * scale_ + min_ (k_anp_scale's expression, so the same float) when it splits them.
  // Loads are unconditional (coordinates clamped into the image, the value zeroed when it is split): a load inside a branch makes
  // hipcc wait for it at the branch's join -- the whole memory latency, in front of everything else the phase has to do.
  struct Patch { float f[3]; double d[3]; unsigned ok; };
  double sc = 1.0, mn = 0.0;
  int sc_img = -1;
  int pe_row[3], pe_col[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) { const int e = ltid + 256 * k; pe_row[k] = e / UD_PW; pe_col[k] = e - pe_row[k] * UD_PW; }
  auto patch_load = [&](Patch& r) __attribute__((always_inline)) {      // the cursor's tile
    const int x0 = cu.tx * 32, y0 = cu.ty * 16;
    r.ok = 0u;
    if (RAW && cu.img != sc_img) {      // (per image, not per tile: the reciprocal is a division)
      const unsigned long long elo = mm_enc[2 * cu.img], ehi = ~mm_enc[2 * cu.img + 1];      // order-preserving encoding (k_slices.h: enc_f64 / dec_f64)
      const double lo = __longlong_as_double((long long)((elo & 0x8000000000000000ull) ? (elo & 0x7FFFFFFFFFFFFFFFull) : ~elo));
      const double hi = __longlong_as_double((long long)((ehi & 0x8000000000000000ull) ? (ehi & 0x7FFFFFFFFFFFFFFFull) : ~ehi));
      double rng = hi - lo;
      if (rng == 0.0) rng = 1.0;
      sc = 1.0 / rng; mn = 0.0 - lo * sc;
      sc_img = cu.img;
    }
    const size_t ibase = (size_t)cu.img * H * W;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int gy = y0 - 2 + pe_row[k], gx = x0 - 2 + pe_col[k];
      const bool in = pe_row[k] < 20 && gy >= 0 && gy < H && gx >= 0 && gx < W;
      const unsigned off = (unsigned)(min(max(gy, 0), H - 1) * W + min(max(gx, 0), W - 1));
      if (RAW) r.d[k] = raw[ibase + off];
      else r.f[k] = image[ibase + off];
      r.ok |= in ? 1u << k : 0u;
    }
  };
  // split into high and low parts; copy s of a part holds e[i + s] at index i (i = -s .. 727 - s; 8 bytes of padding in front of index 0)
  auto patch_store = [&](int slot, const Patch& r) __attribute__((always_inline)) {
    unsigned char* base = smem + E0_POFF + slot * E0_PSLOT + 8;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int e = ltid + 256 * k;
      if (k < 2 || e < 720) {
        const float val = (r.ok >> k & 1u) ? (RAW ? (float)(r.d[k] * sc + mn) : r.f[k]) : 0.0f;
        const ET hi = (ET)val;
        const ET lo = (ET)(val - (float)hi);
        unsigned char* pe = base + 2 * e;
#pragma unroll
        for (int sft = 0; sft < 4; ++sft) {
          *(ET*)(pe + sft * (E0_CS - 2)) = hi;                       // copy sft, index e - sft
          *(ET*)(pe + sft * (E0_CS - 2) + 4 * E0_CS) = lo;
        }
      }
    }
  };
  // ---- first conv: group g (16 halo rows p = 16 g + li) of a tile; wave wv of the OFF group takes g = wv + 4 t.
  // Fragment of lane (li, lk): patch values e[q .. q + 3], q = p + 36 min(lk, 2) (lk = 3 multiplies zero weights: any finite values),
  // from copy q & 3 = li & 3 at index q - (q & 3): byte address = lane constant + 32 g.  Output: halo row p, 16-byte slot lk.
  const int qlane = li + UD_PW * min(lk, 2);
  const int fr_lane = E0_POFF + 8 + (li & 3) * E0_CS + 2 * (qlane - (li & 3));      // + slot * E0_PSLOT + 32 g (+ 4 E0_CS: the low part)
  const int out_lane = UB_OFF(li, lk) * 2;                                            // + buffer + 1024 g  (bit 2 of p = bit 2 of li)
  auto enc0a = [&](int slot, int hb, int x0, int y0) __attribute__((always_inline)) {      // the tile at (x0, y0): patch slot -> halo buffer hb
    const bool interior = x0 > 0 && x0 + 32 < W && y0 > 0 && y0 + 16 < H;
    const unsigned char* fp = smem + fr_lane + slot * E0_PSLOT + 32 * wv;
    unsigned char* op = smem + hb * E0_HB + out_lane + 1024 * wv;
    const int p0 = 16 * wv + li;
    e0_first_conv_batch<ET, 0, 5>(fp, op, wA, b0v, interior, p0, x0, y0, H, W, li < 8);
    e0_first_conv_batch<ET, 5, 10>(fp, op, wA, b0v, interior, p0, x0, y0, H, W, li < 8);
    if (wv == 0) e0_first_conv_batch<ET, 10, 11>(fp, op, wA, b0v, interior, p0, x0, y0, H, W, li < 8);      // group 40 (rows 640 .. 655: the first eight exist)
  };
  // fragment read offsets of the second conv (bytes inside a halo buffer): rows rg8 * 8 + s (s = 0..9), pixel xh * 16 + li + dx
  int xoff[2][3];
  {
    const int rowbase = rg8 * 8 * UD_PW + xh * 16 + li;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) xoff[sp][dx] = UB_OFF(rowbase + sp * UD_PW + dx, lk) * 2;
  }
  // output offsets of this lane inside a tile (elements): skip0 row m of its 8, the pooled row pair
  const unsigned so_lane = (unsigned)(((rg8 * 8) * W + xh * 16 + li) * 32 + 8 * lk);
  const unsigned po_lane = (unsigned)(((rg8 * 4) * (W / 2) + (xh * 16 + li) / 2) * 32 + 8 * lk);

  // ---- prologue: patches of items 0 and 1 -> LDS, first conv of item 0 -> halo buffer 0 (both groups share the work of these)
  int ax0 = 0, ay0 = 0, aimg = 0;      // item p - 1
  int bx0, by0, bimg; bool blive;      // item p
  int cx0, cy0, cimg; bool clive;      // item p + 1
  bool alive = false;
  Patch pr;
  bx0 = cu.tx * 32; by0 = cu.ty * 16; bimg = cu.img; blive = true;
  if (grp == 0) patch_load(pr);
  advance(tid == 0);
  if (grp == 0) patch_store(0, pr);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  cx0 = cu.tx * 32; cy0 = cu.ty * 16; cimg = cu.img; clive = cu.live;
  if (grp == 1) enc0a(0, 0, bx0, by0);
  if (grp == 0 && cu.live) { patch_load(pr); patch_store(1, pr); }
  if (cu.live) advance(tid == 0);
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  f32x4 acc[8][2];
#ifdef PP_STAMP      // 0 ON multiply, 1 OFF first conv, 2 ON barrier, 3 OFF patch loads issued, 4 OFF patch store + epilogue, 5 OFF barrier, 6 phases, 7 loop
  unsigned long long st_[PP_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0};
  PP_T(tl0_);
#endif
  int bf = 0;                          // halo buffer / patch slot of item p: p & 1
  auto phase = [&](const bool on) __attribute__((always_inline)) -> bool {
    if (!blive && !alive) return false;
    PP_T(ta_);
    if (on) {
      if (blive) {
        const unsigned char* sb = smem + bf * E0_HB;
        const unsigned char* wbp = smem + E0_WOFF + UB_OFF(li, lk) * 2;
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int m = 0; m < 8; ++m) acc[m][n] = bv[n];
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
          v8 xq[10];
#pragma unroll
          for (int s = 0; s < 10; ++s) xq[s] = *(const v8*)(sb + xoff[s & 1][dx] + (s & ~1) * UD_PW * 64);
#pragma unroll
          for (int dy = 0; dy < 3; ++dy) {
            v8 wf[2];
#pragma unroll
            for (int n = 0; n < 2; ++n) wf[n] = *(const v8*)(wbp + ((dy * 3 + dx) * 32 + n * 16) * 64);
#pragma unroll
            for (int m = 0; m < 8; ++m)
#pragma unroll
              for (int n = 0; n < 2; ++n) acc[m][n] = E16<ET>::mfma(wf[n], xq[m + dy], acc[m][n]);
          }
        }
      }
#ifdef PP_STAMP
      asm volatile("" :: "v"(acc[7][1]), "v"(acc[0][0]));
#endif
      PP_T(tb_);
      PP_ADD(0, ta_, tb_);
    } else {
      // item p + 2 (the cursor): its patch loads are in flight during the rest of the phase
      const bool stage2 = cu.live;
      if (E0_OFFPRIO) __builtin_amdgcn_s_setprio(E0_OFFPRIO);
      if (stage2) patch_load(pr);
      PP_T(tb_);
      if (clive) enc0a(bf ^ 1, bf ^ 1, cx0, cy0);      // item p + 1: patch slot and halo buffer (p + 1) & 1
      PP_T(tc_);
      PP_ADD(3, ta_, tb_); PP_ADD(1, tb_, tc_);
      // (in front of the epilogue: the compiler's wait for the patch loads then does not include the epilogue's stores)
      if (stage2) patch_store(bf, pr);      // slot (p + 2) & 1 = p & 1: its last reader was enc0a of item p, one barrier ago
      if (alive) {
        // epilogue of item p - 1: rounded, ReLU'd on the rounded values, 16 bytes (8 consecutive channels) per pixel and lane
        ET* out = skip + ((size_t)aimg * H * W + (size_t)ay0 * W + ax0) * 32 + so_lane;
        u32x4 o[8];
        const s16x2 z = {0, 0};
#pragma unroll
        for (int m = 0; m < 8; ++m) {
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const f32x2 a01 = {acc[m][n][0], acc[m][n][1]}, a23 = {acc[m][n][2], acc[m][n][3]};
            o[m][2 * n] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(a01, v2)), z));
            o[m][2 * n + 1] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(a23, v2)), z));
          }
          ud_store16(out + (size_t)m * W * 32, o[m]);
        }
        // 2x2 max pool on the rounded values (non-negative ETs order like int16): rows pair inside the lane, columns with lane li ^ 1
        ET* po = pooled + ((size_t)aimg * (H / 2) * (W / 2) + (size_t)(ay0 / 2) * (W / 2) + ax0 / 2) * 32 + po_lane;
        unsigned pv[16];
#pragma unroll
        for (int mp = 0; mp < 4; ++mp)
#pragma unroll
          for (int i = 0; i < 4; ++i) pv[4 * mp + i] = pp_pkmax(o[2 * mp][i], o[2 * mp + 1][i]);
        pp_pkmax_lane1_n(pv);
#pragma unroll
        for (int mp = 0; mp < 4; ++mp) {
          const u32x4 v = {pv[4 * mp], pv[4 * mp + 1], pv[4 * mp + 2], pv[4 * mp + 3]};
          if ((li & 1) == 0) ud_store16(po + (size_t)mp * (W / 2) * 32, v);
        }
      }
      if (E0_OFFPRIO) __builtin_amdgcn_s_setprio(0);
      PP_T(tf_);
      PP_ADD(4, tc_, tf_);
    }
    PP_T(td_);
    ax0 = bx0; ay0 = by0; aimg = bimg; alive = blive;
    bx0 = cx0; by0 = cy0; bimg = cimg; blive = clive;
    cx0 = cu.tx * 32; cy0 = cu.ty * 16; cimg = cu.img; clive = cu.live;
    if (cu.live) advance(on && ltid == 0);
    bf ^= 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PP_T(te_);
    PP_ADD(on ? 2 : 5, td_, te_);
#ifdef PP_STAMP
    st_[6] += 1;
#endif
    return true;
  };
  if (PP_G1PRIO && grp == 1) __builtin_amdgcn_s_setprio(PP_G1PRIO);
  if (grp == 0) {
#pragma unroll 1
    for (;;) { if (!phase(true)) break; if (!phase(false)) break; }
  } else {
#pragma unroll 1
    for (;;) { if (!phase(false)) break; if (!phase(true)) break; }
  }
#ifdef PP_STAMP
  {
    PP_T(tl1_);
    st_[7] = tl1_ - tl0_;
    if (lane == 0 && blockIdx.x < 256)
      for (int i = 0; i < PP_NSTAMP; ++i) pp_stamp[(256 * 8 + blockIdx.x * 8 + wave) * PP_NSTAMP + i] = st_[i];
  }
#endif
}


// ---------------------------------------------------------------------------------------------------------------------------
// dec0a with its up-convolution:  dec0a( concat( skip0 [32 ch, H x W],  up0( low [64 ch, H/2 x W/2] ) ) ) -> 32 ch, H x W, ping-pong.
// Tiles are 32 x 8 pixels (two sets of every buffer fit beside the conv's weights: 2 x (skip halo 23 040 + up halo 23 040 + low tile
// 13 824) + 36 864 = 156 672 bytes), walked down the image columns (ty fastest: the two halo rows a tile shares with the tile above
// were fetched one phase ago).
//   ON  (group p & 1):  the conv of tile p, both 32-channel chunks: skip(p) then up(p); 144 MFMAs per wave (4 rows x 16 pixels x 32 couts)
//   OFF (other group):  LDS-DMA of skip(p + 1) and of low(p + 2); up-conv of tile p + 1 on the matrix cores (low(p + 1) -> up(p + 1): a wave
//                       owns one of the four output phases, its four weight fragments stay in registers); epilogue of tile p - 1;
//                       counted wait: the pieces have landed (the epilogue's four stores may still be in flight)
// Same arithmetic in the same order as k_dec0a_up16 (and so as k_upconv16 + k_conv3_dma16): the same tensor bit for bit.
#define DA_ROWS (10 * UD_PW)                 // 360 halo rows of 64 B: 10 x 34 pixels at pitch 36
#define DA_HB (DA_ROWS * 64)                 // 23 040
#define DA_LROWS 108                         // 6 x 18 low-resolution pixels per 32-channel chunk
#define DA_LB (2 * DA_LROWS * 64)            // 13 824
#define DA_UP (2 * DA_HB)                    // 46 080
#define DA_LOW (4 * DA_HB)                   // 92 160
#define DA_W (DA_LOW + 2 * DA_LB)            // 119 808: dec0a weights [2 chunks][9 taps][32 rows] of 64 B
#define DA_SMEM (DA_W + 2 * 9 * 32 * 64)     // 156 672
#define DA_NSP 6                             // LDS-DMA pieces per staging wave: skip halo (1 440 slots of 16 B) ...
#define DA_NLP 4                             // ... and low tile (864 slots)
#ifndef DA_ONPRIO
#define DA_ONPRIO 0                          // s_setprio of a wave while it multiplies
#endif

template <int EK>
__global__ void __launch_bounds__(PP_THREADS)
k_dec0a_up_pp(const u16* __restrict__ skip_ /*[img][H W][32]*/, const u16* __restrict__ low_ /*[img][2][H/2 W/2][32]*/,
              const u16* __restrict__ wgt_ /*dec0a packed [9][2][32][32]*/, const float* __restrict__ bias,
              const u16* __restrict__ wup_ /*up0 packed [4][2][32][32]*/, const float* __restrict__ upb,
              u16* __restrict__ dst_ /*[img][H W][32]*/, int H, int W, int nimg, const u16* __restrict__ zero_page_,
              unsigned* __restrict__ ticket, const int* __restrict__ tk_tab, int ntk) {
  using ET = typename EKT<EK>::type;
  using v8 = typename E16<ET>::v8;
  using v2 = typename E16<ET>::v2;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef short s16x2 __attribute__((ext_vector_type(2)));
  const ET* skip = (const ET*)skip_;
  const ET* low = (const ET*)low_;
  const ET* wgt = (const ET*)wgt_;
  const ET* wup = (const ET*)wup_;
  ET* dst = (ET*)dst_;
  const ET* zero_page = (const ET*)zero_page_;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[DA_SMEM];
  __shared__ int s_q[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2), wv = __builtin_amdgcn_readfirstlane(wave & 3);
  const int xh = wv & 1, rg4 = wv >> 1;
  const int ltid = tid & (PP_GTHREADS - 1);
  const int H2 = H >> 1, W2 = W >> 1;
  const int tiles_x = W / 32, tiles_y = H / 8;
  if (tid == 0) { s_q[0] = ud_take_ticket(ticket); s_q[1] = ud_take_ticket(ticket); }

  // ---- once per workgroup: the conv's weights -> LDS; this wave's up-conv fragments and the biases -> registers
  // dec0a: LDS row (chunk, tap, 16 n + i) <- packed row (tap, chunk, channel 8 (i >> 2) + 4 n + (i & 3)); slot swizzle on the source
  for (int e = tid; e < 2 * 9 * 32 * 4; e += PP_THREADS) {
    const int row = e >> 2, q = e & 3;
    const int cc = row / 288, rem = row - cc * 288, tap = rem >> 5, j = rem & 31;
    const int ch = 8 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);
    *(u32x4*)(smem + DA_W + e * 16) = *(const u32x4*)(wgt + (size_t)((tap * 2 + cc) * 32 + ch) * 32 + ((q ^ ((row >> 1) & 2)) << 3));
  }
  // up0: phase wv, chunk kc, cout tile n: MFMA row li = channel 8 (li >> 2) + 4 n + (li & 3), k = 8 lk .. 8 lk + 7
  v8 uw[2][2];
  f32x4 bv[2], ub[2];
#pragma unroll
  for (int kc = 0; kc < 2; ++kc)
#pragma unroll
    for (int n = 0; n < 2; ++n) {
      u32x4 t = *(const u32x4*)(wup + (size_t)((wv * 2 + kc) * 32 + 8 * (li >> 2) + 4 * n + (li & 3)) * 32 + 8 * lk);
      pp_settle(t);
      uw[kc][n] = __builtin_bit_cast(v8, t);
    }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
#pragma unroll
    for (int r = 0; r < 4; ++r) { bv[n][r] = bias[8 * lk + 4 * n + r]; ub[n][r] = upb[8 * lk + 4 * n + r]; }
    pp_settle(bv[n]); pp_settle(ub[n]);
  }
  __syncthreads();
  const int t0 = __builtin_amdgcn_readfirstlane(s_q[0]);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __syncthreads();

  PpCursor cu;
  if (t0 >= ntk) return;
  auto decode = [&](int w) __attribute__((always_inline)) { cu.ty = w % tiles_y; w /= tiles_y; cu.tx = w % tiles_x; cu.img = w / tiles_x; };
  cu.w = tk_tab[t0]; cu.wend = tk_tab[t0 + 1]; cu.qk = 1; cu.live = true;
  decode(cu.w);
  auto advance = [&](const bool fetcher) __attribute__((always_inline)) {
    if (cu.w + 1 < cu.wend) { ++cu.w; if (++cu.ty == tiles_y) { cu.ty = 0; if (++cu.tx == tiles_x) { cu.tx = 0; ++cu.img; } } return; }
    const int nt = __builtin_amdgcn_readfirstlane(s_q[cu.qk]);
    if (nt < ntk) {
      if (fetcher) s_q[cu.qk ^ 1] = ud_take_ticket(ticket);
      cu.qk ^= 1;
      cu.w = tk_tab[nt]; cu.wend = tk_tab[nt + 1];
      decode(cu.w);
    } else cu.live = false;
  };

  // ---- staging plans of a group (lane constants).  Skip halo: slot e = ltid + 256 k -> row e >> 2 = py * 36 + px; low tile: slot e ->
  // chunk e / 432, row (e % 432) >> 2 = a * 18 + b.  The 16-byte slot of a row is XOR-swizzled on the source address (bit 2 of the row).
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(ud_lptr)smem);
  const unsigned wv1024 = (unsigned)(wv * 1024);
  int svoff[DA_NSP], lvoff[DA_NLP];
  const int sq8 = ((ltid & 3) ^ (((ltid >> 2) >> 1) & 2)) * 8;
#pragma unroll
  for (int k = 0; k < DA_NSP; ++k) {
    const int r = (ltid >> 2) + 64 * k, py = r / UD_PW, px = r - py * UD_PW;
    svoff[k] = (r < DA_ROWS && px < 34) ? ((py * W + px) * 32 + sq8) * 2 : ((W + 1) * 32 + sq8) * 2;
  }
#pragma unroll
  for (int k = 0; k < DA_NLP; ++k) {
    const int e = ltid + 256 * k, c = e >= 4 * DA_LROWS ? 1 : 0, rem = e - c * 4 * DA_LROWS, lr = rem >> 2;
    const int a = lr / 18, b = lr - a * 18;
    lvoff[k] = (c * (H2 * W2) + a * W2 + b) * 64 + (((rem & 3) ^ ((lr >> 1) & 2)) << 4);
  }
  const bool s5 = (ltid >> 2) + 64 * 5 < DA_ROWS, l3 = ltid + 256 * 3 < 2 * 4 * DA_LROWS;      // the lanes of the last, partial pieces
  auto stage_skip = [&](int bf) __attribute__((always_inline)) {      // the cursor's tile -> skip buffer bf
    const ET* simg = skip + (size_t)cu.img * H * W * 32;
    const unsigned lb = lds0 + (unsigned)(bf * DA_HB) + wv1024;
    if (cu.tx > 0 && cu.tx + 1 < tiles_x && cu.ty > 0 && cu.ty + 1 < tiles_y) {      // interior: every halo pixel is inside the image
      const ET* base = simg + ((size_t)(cu.ty * 8 - 1) * W + (cu.tx * 32 - 1)) * 32;
#pragma unroll
      for (int k = 0; k < DA_NSP; ++k)
        if (k < DA_NSP - 1 || s5) ud_dma16_s(lb + k * 4096, (unsigned)svoff[k], base);
    } else {
#pragma unroll
      for (int k = 0; k < DA_NSP; ++k) {
        const int r = (ltid >> 2) + 64 * k, py = r / UD_PW, px = r - py * UD_PW;
        const int gx = cu.tx * 32 + px - 1, gy = cu.ty * 8 + py - 1;
        const bool ok = px < 34 && gx >= 0 && gx < W && gy >= 0 && gy < H;
        const ET* p = ok ? simg + (unsigned)((gy * W + gx) * 32 + sq8) : zero_page;
        if (k < DA_NSP - 1 || s5) ud_dma16(lb + k * 4096, p);
      }
    }
  };
  auto stage_low = [&](int bf) __attribute__((always_inline)) {      // the cursor's tile -> low buffer bf
    const ET* limg = low + (size_t)cu.img * H2 * W2 * 64;
    const unsigned lb = lds0 + (unsigned)(DA_LOW + bf * DA_LB) + wv1024;
    if (cu.tx > 0 && cu.tx + 1 < tiles_x && cu.ty > 0 && cu.ty + 1 < tiles_y) {
      const ET* base = limg + ((size_t)(cu.ty * 4 - 1) * W2 + (cu.tx * 16 - 1)) * 32;
#pragma unroll
      for (int k = 0; k < DA_NLP; ++k)
        if (k < DA_NLP - 1 || l3) ud_dma16_s(lb + k * 4096, (unsigned)lvoff[k], base);
    } else {
#pragma unroll
      for (int k = 0; k < DA_NLP; ++k) {
        const int e = ltid + 256 * k, c = e >= 4 * DA_LROWS ? 1 : 0, rem = e - c * 4 * DA_LROWS, lr = rem >> 2;
        const int a = lr / 18, b = lr - a * 18;
        const int ly = cu.ty * 4 - 1 + a, lx = cu.tx * 16 - 1 + b;
        const bool ok = ly >= 0 && ly < H2 && lx >= 0 && lx < W2;
        const ET* p = ok ? limg + (size_t)c * H2 * W2 * 32 + (unsigned)((ly * W2 + lx) * 32 + (((rem & 3) ^ ((lr >> 1) & 2)) << 3)) : zero_page;
        if (k < DA_NLP - 1 || l3) ud_dma16(lb + k * 4096, p);
      }
    }
  };
  // ---- the up-conv half of a halo tile: phase (pdy, pdx) = (wv >> 1, wv & 1) of this wave: 5 x 17 pixels, six groups of 16
  const int pdy = wv >> 1, pdx = wv & 1;
  int u_low[6], u_out[6];      // byte offsets: the low pixel's row in a chunk of the low tile, the halo row in the up buffer (-1: no pixel)
#pragma unroll
  for (int g = 0; g < 6; ++g) {
    const int t = 16 * g + li;
    const bool valid = t < 85;
    const int a = valid ? t / 17 : 0, bc = valid ? t - a * 17 : 0;
    const int lowrow = (a + (pdy == 0 ? 1 : 0)) * 18 + bc + (pdx == 0 ? 1 : 0);
    const int py = 2 * a + (pdy == 0 ? 1 : 0), px = 2 * bc + (pdx == 0 ? 1 : 0);
    u_low[g] = UB_OFF(lowrow, lk) * 2;
    u_out[g] = valid ? UB_OFF(py * UD_PW + px, lk) * 2 : -1;
  }
  auto upconv = [&](int lb, int ub_, int x0, int y0) __attribute__((always_inline)) {      // low buffer lb -> up buffer ub_, the tile at (x0, y0)
    const bool interior = x0 > 0 && x0 + 32 < W && y0 > 0 && y0 + 8 < H;
    const unsigned char* lp = smem + DA_LOW + lb * DA_LB;
    unsigned char* op = smem + DA_UP + ub_ * DA_HB;
#pragma unroll
    for (int g0 = 0; g0 < 6; g0 += 3) {      // (two batches of three groups: reads, MFMAs, conversions and writes of a batch together)
      v8 xf[3][2];
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int kc = 0; kc < 2; ++kc) xf[g][kc] = *(const v8*)(lp + kc * (DA_LROWS * 64) + u_low[g0 + g]);
      f32x4 ua[3][2];
#pragma unroll
      for (int g = 0; g < 3; ++g)
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          ua[g][n] = ub[n];
#pragma unroll
          for (int kc = 0; kc < 2; ++kc) ua[g][n] = E16<ET>::mfma(uw[kc][n], xf[g][kc], ua[g][n]);
        }
#pragma unroll
      for (int g = 0; g < 3; ++g) {
        u32x4 o;
#pragma unroll
        for (int n = 0; n < 2; ++n) {
          const f32x2 a01 = {ua[g][n][0], ua[g][n][1]}, a23 = {ua[g][n][2], ua[g][n][3]};
          o[2 * n] = __builtin_bit_cast(unsigned, __builtin_convertvector(a01, v2));
          o[2 * n + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(a23, v2));
        }
        if (!interior) {      // outside the image: the conv's zero padding
          const int t = 16 * (g0 + g) + li, a = t / 17, bc = t - a * 17;
          const int gy = y0 - 1 + 2 * a + (pdy == 0 ? 1 : 0), gx = x0 - 1 + 2 * bc + (pdx == 0 ? 1 : 0);
          if (!(gy >= 0 && gy < H && gx >= 0 && gx < W)) o = u32x4{0u, 0u, 0u, 0u};
        }
        if (u_out[g0 + g] >= 0) *(u32x4*)(op + u_out[g0 + g]) = o;
      }
    }
  };
  // fragment read offsets of the conv (bytes inside a halo buffer): rows rg4 * 4 + s (s = 0..5), pixel xh * 16 + li + dx
  int xoff[2][3];
  {
    const int rowbase = rg4 * 4 * UD_PW + xh * 16 + li;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) xoff[sp][dx] = UB_OFF(rowbase + sp * UD_PW + dx, lk) * 2;
  }
  const int woff = UB_OFF(li, lk) * 2;
  const unsigned so_lane = (unsigned)(((rg4 * 4) * W + xh * 16 + li) * 32 + 8 * lk);

  // ---- prologue: low(0), skip(0), low(1) land; up(0) is computed
  int ax0 = 0, ay0 = 0, aimg = 0;      // item p - 1
  int bx0, by0, bimg; bool blive;      // item p
  int cx0, cy0, cimg; bool clive;      // item p + 1
  bool alive = false;
  bx0 = cu.tx * 32; by0 = cu.ty * 8; bimg = cu.img; blive = true;
  if (grp == 0) stage_low(0); else stage_skip(0);
  advance(tid == 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
  cx0 = cu.tx * 32; cy0 = cu.ty * 8; cimg = cu.img; clive = cu.live;
  if (grp == 0 && cu.live) stage_low(1);
  if (grp == 1) upconv(0, 0, bx0, by0);
  if (cu.live) advance(tid == 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");

  f32x4 acc[4][2];
#ifdef PP_STAMP      // 0 ON multiply, 1 OFF up-conv, 2 ON barrier, 3 OFF pieces issued, 4 OFF epilogue + vm wait, 5 OFF barrier, 6 phases, 7 loop
  unsigned long long st_[PP_NSTAMP] = {0, 0, 0, 0, 0, 0, 0, 0};
  PP_T(tl0_);
#endif
  int bf = 0;                          // buffers of item p: p & 1
  auto phase = [&](const bool on) __attribute__((always_inline)) -> bool {
    if (!blive && !alive) return false;
    PP_T(ta_);
    if (on) {
      if (blive) {
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int m = 0; m < 4; ++m) acc[m][n] = bv[n];
        // Six groups (chunk, dx) of three taps.  The pixel rows of the next group and the (up chunk's) weight fragments of the next tap
        // are requested before the MFMAs of this tap: hipcc otherwise reads a fragment right in front of its first use, one exposed
        // LDS latency per four MFMAs (measured: 4 000 - 4 800 cycles for the 144 MFMAs of a phase).
        if (DA_ONPRIO) __builtin_amdgcn_s_setprio(DA_ONPRIO);
        const unsigned char* sb0 = smem + bf * DA_HB;
        const unsigned char* sb1 = smem + DA_UP + bf * DA_HB;
        const unsigned char* wbp = smem + DA_W + woff;
        v8 xq[2][6], wf[3][2];
        auto tap_of = [](int t) { const int g = t / 3, dy = t % 3; return (g / 3) * 9 + dy * 3 + g % 3; };      // t = 3 (3 chunk + dx) + dy -> packed tap row block
#pragma unroll
        for (int s = 0; s < 6; ++s) xq[0][s] = *(const v8*)(sb0 + xoff[s & 1][0] + (s & ~1) * UD_PW * 64);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int n = 0; n < 2; ++n) wf[t][n] = *(const v8*)(wbp + (tap_of(t) * 32 + n * 16) * 64);
#pragma unroll
        for (int t = 0; t < 18; ++t) {
          const int g = t / 3, dy = t % 3;
          if (dy == 0 && g + 1 < 6) {      // the next group's six pixel rows
            const unsigned char* sbn = (g + 1) / 3 == 0 ? sb0 : sb1;
#pragma unroll
            for (int s = 0; s < 6; ++s) xq[(g + 1) & 1][s] = *(const v8*)(sbn + xoff[s & 1][(g + 1) % 3] + (s & ~1) * UD_PW * 64);
          }
          if (t + 2 < 18) {      // the weight fragments of the tap after the next
#pragma unroll
            for (int n = 0; n < 2; ++n) wf[(t + 2) % 3][n] = *(const v8*)(wbp + (tap_of(t + 2) * 32 + n * 16) * 64);
          }
          __builtin_amdgcn_sched_barrier(0);      // (requests in front of this tap's MFMAs, MFMAs in front of the next requests)
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < 2; ++n) acc[m][n] = E16<ET>::mfma(wf[t % 3][n], xq[g & 1][m + dy], acc[m][n]);
          __builtin_amdgcn_sched_barrier(0);
        }
        if (DA_ONPRIO) __builtin_amdgcn_s_setprio(0);
      }
#ifdef PP_STAMP
      asm volatile("" :: "v"(acc[3][1]), "v"(acc[0][0]));
#endif
      PP_T(tb_);
      PP_ADD(0, ta_, tb_);
    } else {
      // item p + 1 (c): its skip tile; item p + 2 (the cursor): its low tile -- both land during this phase
      bool had_stores = false;
      if (clive) {
        const PpCursor keep = cu;
        cu.tx = cx0 >> 5; cu.ty = cy0 >> 3; cu.img = cimg;
        stage_skip(bf ^ 1);
        cu = keep;
      }
      if (cu.live) stage_low(bf);
      PP_T(tb_);
      if (clive) upconv(bf ^ 1, bf ^ 1, cx0, cy0);
      PP_T(tc_);
      PP_ADD(3, ta_, tb_); PP_ADD(1, tb_, tc_);
      if (alive) {
        ET* out = dst + ((size_t)aimg * H * W + (size_t)ay0 * W + ax0) * 32 + so_lane;
        const s16x2 z = {0, 0};
#pragma unroll
        for (int m = 0; m < 4; ++m) {
          u32x4 o;
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const f32x2 a01 = {acc[m][n][0], acc[m][n][1]}, a23 = {acc[m][n][2], acc[m][n][3]};
            o[2 * n] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(a01, v2)), z));
            o[2 * n + 1] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, __builtin_convertvector(a23, v2)), z));
          }
          ud_store16(out + (size_t)m * W * 32, o);
        }
        had_stores = true;
      }
      // the pieces have landed (they are older than the stores)
      if (had_stores) pp_wait_vm<4>(); else pp_wait_vm<0>();
      PP_T(tf_);
      PP_ADD(4, tc_, tf_);
    }
    PP_T(td_);
    ax0 = bx0; ay0 = by0; aimg = bimg; alive = blive;
    bx0 = cx0; by0 = cy0; bimg = cimg; blive = clive;
    cx0 = cu.tx * 32; cy0 = cu.ty * 8; cimg = cu.img; clive = cu.live;
    if (cu.live) advance(on && ltid == 0);
    bf ^= 1;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    PP_T(te_);
    PP_ADD(on ? 2 : 5, td_, te_);
#ifdef PP_STAMP
    st_[6] += 1;
#endif
    return true;
  };
  if (PP_G1PRIO && grp == 1) __builtin_amdgcn_s_setprio(PP_G1PRIO);
  if (grp == 0) {
#pragma unroll 1
    for (;;) { if (!phase(true)) break; if (!phase(false)) break; }
  } else {
#pragma unroll 1
    for (;;) { if (!phase(false)) break; if (!phase(true)) break; }
  }
#ifdef PP_STAMP
  {
    PP_T(tl1_);
    st_[7] = tl1_ - tl0_;
    if (lane == 0 && blockIdx.x < 256)
      for (int i = 0; i < PP_NSTAMP; ++i) pp_stamp[(2 * 256 * 8 + blockIdx.x * 8 + wave) * PP_NSTAMP + i] = st_[i];
  }
#endif
}

}  // namespace sh
