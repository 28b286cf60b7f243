// k_unet_bf16_dma.h -- 3x3 conv, bf16, for the layers with >= 64 output channels: persistent workgroups with
// LDS-DMA double buffering (the ~900 TFLOP/s ceiling of the two-barriers-per-chunk kernel in k_unet_bf16.h is a property
// of that structure: every chunk waits for its own staging; here the next chunk streams into the other LDS buffer while
// this one is multiplied).
//
// Workgroup = 512 lanes = 8 waves, one per CU (155 KB of LDS).  Work item = (image, 32x16-pixel tile, 64-cout group); a
// workgroup walks a contiguous range of items (cout groups of one tile are neighbours: their input tile stays in L2), the
// (item, 32-channel chunk) steps are flattened and pipelined: while the MFMAs of step i read LDS buffer i & 1, the
// `global_load_lds_dwordx4` pieces of step i + 1 -- one issued per tap, between the MFMA groups -- fill buffer (i + 1) & 1.
// One raw s_barrier per step; the DMA is retired by a counted s_waitcnt that leaves the epilogue stores of the previous
// item in flight (those are inline-asm stores so that their number is exact).
// LDS image per buffer: 18 x 36 halo-pixel rows (pitch 36: the swizzle bit of a fragment row then depends only on row
// parity, dx and the lane -> 7 address registers + immediates for all 72 fragment reads of a step), then 576 weight rows
// ([tap][64 couts]); 32 channels = 64 B per row, XOR slot swizzle as in k_unet_bf16.h, applied on the DMA's SOURCE address
// (the DMA writes lane-linear) and on the fragment read.  Out-of-image halo pixels read a 64-byte page of zeros.
#pragma once
#include "k_unet_bf16.h"

namespace sh {

#define UD_THREADS 512
#define UD_ROWS (UD_INROWS + 576)           // 1224
#define UD_BUF (UD_ROWS * 64)               // 78336
#define UD_SLOTS (UD_ROWS * 4)              // 4896
#define UD_BIAS_OFF (2 * UD_BUF)
#define UD_SMEM (2 * UD_BUF + 2048)         // 158720

template <int N> __device__ inline void ud_wait_vm() {      // s_waitcnt takes an immediate
  if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
  else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
  else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
  else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
  else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Yielding the reserve (sh_ctx: persistent_grid / lane_busy).  Beside another lane a persistent launch leaves SHOULDER_CU_RESERVE CUs
// to that lane's geometry chain -- but the chain is in flight for only part of a pass (5.2 of 7.8 ms on the two-lane headline).
// With a YieldArg the launch covers the whole chip and its LAST `nres` workgroups ask, whenever they would take a work ticket,
// whether any OTHER lane of the device has its busy word up (one word per lane, written on the lane's stream around its chain):
// if so they take no (further) ticket and leave their CU.  Only those workgroups ever yield, so every ticket is always taken.
#define SH_YIELD_SLOTS 16
struct YieldArg { const int* flags; int self; int nres; };
__device__ inline int ud_take_ticket(unsigned* ticket, int ntk, const YieldArg& y) {      // (called by one lane of the workgroup)
  if (y.flags != nullptr && (int)blockIdx.x >= (int)gridDim.x - y.nres) {
    int busy = 0;
#pragma unroll
    for (int j = 0; j < SH_YIELD_SLOTS; ++j) busy |= (j != y.self) ? __hip_atomic_load(y.flags + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
    if (busy) return ntk;      // "no ticket left" for this workgroup
  }
  return (int)atomicAdd(ticket, 1u);
}

// FUSE: 0, UF_POOL (2x2 max pool written beside the output) or UF_HEAD (1x1 head: only the logits leave the kernel).
// NN: 16-cout tiles per item: 4 (64-cout groups) or 2 (the 32-channel level: 288 weight rows, 8 DMA pieces per step).
// Output channels of an item are dealt to the accumulator tiles so that a lane ends up with CONSECUTIVE channels: weight
// row j = 16 n + i of the LDS image (tile n, MFMA row i) holds channel 4 NN (i >> 2) + 4 n + (i & 3) of the item's group, so
// lane group lk owns channels 4 NN lk .. 4 NN lk + 4 NN - 1 of a pixel and the epilogue stores 16 bytes per instruction
// (half the store instructions of the 4-channel form: the epilogue is store-ISSUE bound, cdna_hip_programming.md T21).  The
// permutation is applied where the LDS-DMA computes its source rows; the packed weights in HBM keep channel order.
// SCHED: order of the 9 taps inside a step.  0: row-major taps, 4 pixel + NN weight fragments read per tap (72 reads at
// NN = 4).  1: column (dx) major -- the 6 pixel-row fragments of a dx serve its three dy taps (rows m + dy), so a step reads
// 18 pixel + 9 NN weight fragments (54 at NN = 4): a quarter less LDS traffic per MFMA (what the chip spends per MFMA
// decides the clock it holds, cdna_hip_programming.md 5.4 rule 28).
// WRES: weights resident.  A layer with ONE cout group whose packed weights fit beside two input buffers (nchunk * 16 NN <= 128
// rows per tap set: 32->64, 64->64, 64->32, 32->32 -- the 64- and 32-channel levels, where a weight set is shared by 8 192 /
// 32 768 tiles) loads them into LDS once per workgroup; a step then stages only its 41 KB input tile: 6 DMA pieces instead of
// 10 / 8 (the ablation builds price the weight pieces at 10-14 % of such a layer).
// WRES = 2 (32 -> 32 channels: one chunk, NN = 2): the 18 weight fragments of a lane additionally stay in REGISTERS (72 VGPRs) for the
// whole launch.  A 32-cout item reads one LDS fragment per two MFMAs, i.e. 256 B per cycle and CU at full matrix rate -- twice what
// the LDS delivers; without the weight reads a step needs 18 fragment reads for its 72 MFMAs instead of 36.
template <int EK, int FUSE, int NN = 4, int SCHED = 1, int WRES = 0>
__global__ void __launch_bounds__(UD_THREADS)
k_conv3_dma16(const u16* __restrict__ src0_, const u16* __restrict__ src1_, int C0, int C1,
              const u16* __restrict__ wgt_, const float* __restrict__ bias, u16* __restrict__ dst_,
              int H, int W, int Cout, int relu, int nimg, const u16* __restrict__ zero_page_, u16* __restrict__ pooled_,
              const float* __restrict__ head_w, const float* __restrict__ head_b, float* __restrict__ logits,
              unsigned* __restrict__ ticket /*zero at launch; nullptr: fixed equal shares*/, const int* __restrict__ tk_tab /*[ntk + 1] item bounds*/, int ntk,
              YieldArg yl) {
  using ET = typename EKT<EK>::type;
  const ET* src0 = (const ET*)src0_;
  const ET* src1 = (const ET*)src1_;
  const ET* wgt = (const ET*)wgt_;
  ET* dst = (ET*)dst_;
  const ET* zero_page = (const ET*)zero_page_;
  ET* pooled = (ET*)pooled_;

  using v8 = typename E16<ET>::v8;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[UD_SMEM];
  static_assert(NN == 4 || NN == 2, "64- or 32-cout items");
  static_assert(!(FUSE & UF_HEAD) || NN == 2, "the head reads all 32 channels of a pixel from one item");
  constexpr int WR = 16 * NN;                        // weight rows per tap
  constexpr int WSH = NN == 4 ? 6 : 5;               // log2(WR)
  constexpr int SLOTS = WRES ? UD_INROWS * 4 : (UD_INROWS + 9 * WR) * 4;    // 16-byte slots of a step: 4896 / 3744; 2592 with resident weights
  constexpr int NPIECE = (SLOTS + UD_THREADS - 1) / UD_THREADS;      // 10 / 8; 6
  constexpr int BUFB = WRES ? UD_INROWS * 64 : UD_BUF;               // bytes of a staging buffer
  constexpr int WRES_OFF = 2 * UD_INROWS * 64;                       // resident weights: [chunk][tap][WR rows] of 64 B behind the two input buffers
  constexpr int NSTORE = (FUSE & UF_HEAD) ? 4 : 2 * NN + ((FUSE & UF_POOL) ? NN : 0);      // dwordx4 stores per wave and item
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int xh = wave & 1, rg = wave >> 1;
  const int Cin = C0 + C1, nchunk = Cin >> 5;
  const int tiles_x = W / 32, tiles_y = H / 16, ngroups = Cout / WR;
  const int total = nimg * tiles_x * tiles_y * ngroups;
  // Work distribution.  Fixed: workgroup g walks items [g per, (g + 1) per).  Tickets (the default beside another lane): runs
  // of items of DECREASING length (tk_tab, built on the host: every ticket is half of what would be a fair share of the
  // remaining items) are handed out by a global counter, so a workgroup that starts late -- its CU was still held by a
  // kernel of the engine's other lane -- takes less and the layer ends when the chip runs out of work; long first tickets
  // keep the counter traffic at ~5 atomics per workgroup, short last ones the tail.  The id of the ticket after the
  // current one is always already in LDS (fetched one ticket ahead by lane 0 of wave 0): no step waits for the counter.
  __shared__ int s_q[2];
  const bool dyn = ticket != nullptr;
  if (dyn && tid == 0) { s_q[0] = ud_take_ticket(ticket, ntk, yl); s_q[1] = ud_take_ticket(ticket, ntk, yl); }
#ifdef SH_DMA_PRIO_HALF
  if (threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);      // experiment: static priority for the younger half (MI355X_MICROARCH.md, two waves per SIMD, item 4)
#endif
#ifdef SH_DMA_PRIO
  __builtin_amdgcn_s_setprio(SH_DMA_PRIO);      // experiment: waves of another lane's kernels that share the SIMD lose the issue arbitration
#endif

  float* s_bias = (float*)(smem + UD_BIAS_OFF);
  for (int i = tid; i < Cout; i += UD_THREADS) s_bias[i] = bias[i];
  if (FUSE & UF_HEAD) for (int i = tid; i < 32; i += UD_THREADS) s_bias[256 + i] = head_w[i];
  if constexpr (WRES != 0) {
    // LDS row (chunk, tap, 16 n + i) <- packed row (tap, chunk, channel of that MFMA row); the 16-byte slot swizzle on the source
    const int nrows = (Cin >> 5) * 9 * WR;
    for (int e = tid; e < nrows * 4; e += UD_THREADS) {
      const int row = e >> 2, q = e & 3;
      const int cc = row / (9 * WR), rem = row - cc * 9 * WR, tap = rem >> WSH, j = rem & (WR - 1);
      const int ch = 4 * NN * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);
      *(u32x4*)(smem + WRES_OFF + e * 16) = *(const u32x4*)(wgt + (size_t)((tap * (Cin >> 5) + cc) * Cout + ch) * 32 + ((q ^ ((row >> 1) & 2)) << 3));
    }
  }
  __syncthreads();       // every ordinary load is retired before the first LDS-DMA is issued
  static_assert(WRES != 2 || (NN == 2 && SCHED == 1), "register-resident weights: the 32 -> 32 layers");
  v8 wreg[WRES == 2 ? 9 : 1][2];
  if constexpr (WRES == 2) {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int n = 0; n < 2; ++n) wreg[tap][n] = *(const v8*)(smem + WRES_OFF + (tap * WR + n * 16) * 64 + UB_OFF(lane & 15, lane >> 4) * 2);
  }
  int qk = 1;                                                  // s_q slot of the ticket after the current one
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

  // staging plan: slot e_k = tid + 512 k -> row r_k = (tid >> 2) + 128 k; k = 0..4 halo rows, k = 5 mixed, k = 6..9 weight rows.
  // The swizzle bit (bit 2 of the row) is the same for every k, weight rows advance by two taps per k.
  const int r0 = tid >> 2;
  const int q8 = ((tid & 3) ^ ((r0 >> 1) & 2)) * 8;
  const int rw5 = r0 + 640 - UD_INROWS;
  const int wstep = (128 / WR) * nchunk * Cout * 32;      // 128 rows per piece = 2 (4) taps
  const int wj5 = rw5 & (WR - 1);                                                     // LDS weight row inside its tap: 16 n + i
  const int wch5 = 4 * NN * ((wj5 & 15) >> 2) + 4 * (wj5 >> 4) + (wj5 & 3);            // the channel it holds
  const int wrel5 = ((rw5 >> WSH) * nchunk * Cout + wch5) * 32 + q8;
  const bool in5 = rw5 < 0;
  const bool wlast = tid + 512 * (NPIECE - 1) < SLOTS;

  // fragment read offsets (bytes inside a buffer)
  int xoff[2][3], woff;
  {
    const int rowbase = rg * 4 * UD_PW + xh * 16 + li;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) xoff[sp][dx] = UB_OFF(rowbase + sp * UD_PW + dx, lk) * 2;
    woff = UB_OFF(UD_INROWS + li, lk) * 2;
  }

  int i_g, i_tx, i_ty, i_img;      // item being staged
  auto decode = [&](int w) {
    i_g = w % ngroups; w /= ngroups;
    i_tx = w % tiles_x; w /= tiles_x;
    i_ty = w % tiles_y; i_img = w / tiles_y;
  };
  decode(w_begin);
  int pixoff[6];
  auto item_lane_setup = [&]() {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      const int r = r0 + 128 * k;
      const int py = r / UD_PW, px = r - py * UD_PW;
      const int gx = i_tx * 32 + px - 1, gy = i_ty * 16 + py - 1;
      const bool ok = px < 34 && gx >= 0 && gx < W && gy >= 0 && gy < H;
      pixoff[k] = ok ? gy * W + gx : -1;
    }
  };
  const ET* n_simg; const ET* n_wbase; int n_Cs, n_cb; unsigned char* n_lbase;      // step being staged (wave-uniform)
  auto describe = [&](int cc, int buf) {
    const int c0 = cc * 32;
    const bool first = c0 < C0;
    n_Cs = first ? C0 : C1;
    n_cb = ((first ? c0 : c0 - C0) >> 5) * (H * W);      // pixel offset of this chunk's 32-channel plane (channel-blocked activations)
    n_simg = (first ? src0 : src1) + (size_t)i_img * H * W * n_Cs;
    n_wbase = wgt + ((size_t)cc * Cout + i_g * WR) * 32;
    n_lbase = smem + buf * BUFB + wave * 1024;
  };
  auto piece = [&](int k) {      // k is a compile-time constant at every call site
#ifdef SH_ABL_NODMA
    if (k >= SH_ABL_NODMA) return;      // ablation build (wrong results): what the steady state costs without (some of) its DMA pieces
#endif
    if (k < 5) {
      const ET* p = pixoff[k] >= 0 ? n_simg + (unsigned)((n_cb + pixoff[k]) * 32 + q8) : zero_page;
      __builtin_amdgcn_global_load_lds((ud_gptr)p, (ud_lptr)(n_lbase + k * 8192), 16, 0, 0);
    } else if (k == 5) {
      const ET* pi = pixoff[5] >= 0 ? n_simg + (unsigned)((n_cb + pixoff[5]) * 32 + q8) : zero_page;
      if constexpr (WRES != 0) {      // the last 8 halo rows: the first 32 lanes of wave 0
        if (in5) __builtin_amdgcn_global_load_lds((ud_gptr)pi, (ud_lptr)(n_lbase + 5 * 8192), 16, 0, 0);
      } else {
        const ET* p = in5 ? pi : n_wbase + wrel5;
        __builtin_amdgcn_global_load_lds((ud_gptr)p, (ud_lptr)(n_lbase + 5 * 8192), 16, 0, 0);
      }
    } else if (k < NPIECE - 1) {
      __builtin_amdgcn_global_load_lds((ud_gptr)(n_wbase + (wrel5 + (k - 5) * wstep)), (ud_lptr)(n_lbase + k * 8192), 16, 0, 0);
    } else if (k == NPIECE - 1) {
      if (wlast) __builtin_amdgcn_global_load_lds((ud_gptr)(n_wbase + (wrel5 + (NPIECE - 6) * wstep)), (ud_lptr)(n_lbase + (NPIECE - 1) * 8192), 16, 0, 0);
    }
  };

  item_lane_setup();
  describe(0, 0);
#pragma unroll
  for (int k = 0; k < NPIECE; ++k) piece(k);
  int buf = 0;
  bool stores_in_flight = false;
  for (int w = w_begin;;) {
    bool more = true;
    const int c_x0 = i_tx * 32, c_y0 = i_ty * 16, c_img = i_img, c_n0 = i_g * WR;
    f32x4 acc[4][NN];
#pragma unroll
    for (int n = 0; n < NN; ++n) {
      const f32x4 bv = *(const f32x4*)(s_bias + c_n0 + 4 * NN * lk + 4 * n);
#pragma unroll
      for (int m = 0; m < 4; ++m) acc[m][n] = bv;
    }
    for (int cc = 0; cc < nchunk; ++cc) {
      // this wave's DMA pieces of the step are older than the NSTORE epilogue stores of the previous item
      if (stores_in_flight) ud_wait_vm<NSTORE>();
      else ud_wait_vm<0>();
      stores_in_flight = false;
      __builtin_amdgcn_s_barrier();      // every wave's pieces have landed; every wave is done reading the other buffer
      bool has_next = true;
      if (cc + 1 < nchunk) {
        describe(cc + 1, buf ^ 1);
      } else if (w + 1 < w_end) {
        ++w;
        if (++i_g == ngroups) { i_g = 0; if (++i_tx == tiles_x) { i_tx = 0; if (++i_ty == tiles_y) { i_ty = 0; ++i_img; } } }
        item_lane_setup();
        describe(0, buf ^ 1);
      } else if (dyn) {      // next ticket: its id was written before this step's barrier; the slot it frees is refilled for the one after
        const int nt = __builtin_amdgcn_readfirstlane(s_q[qk]);
        if (nt < ntk) {
          if (tid == 0) s_q[qk ^ 1] = ud_take_ticket(ticket, ntk, yl);
          qk ^= 1;
          w = tk_tab[nt]; w_end = tk_tab[nt + 1];
          decode(w);
          item_lane_setup();
          describe(0, buf ^ 1);
        } else { has_next = false; more = false; }
      } else { has_next = false; more = false; }
#ifdef SH_DMA_UNCOND
      // experiment: the pieces are issued unconditionally (no branch around each of them in the step body); without a next
      // step they re-stage the current step's sources into the other buffer, which nobody reads
      if (!has_next) n_lbase = smem + (buf ^ 1) * BUFB + wave * 1024;
#define UD_HAS_NEXT true
#else
#define UD_HAS_NEXT has_next
#endif
      const unsigned char* sb = smem + buf * BUFB;
      const unsigned char* xb[2][3];
#pragma unroll
      for (int sp = 0; sp < 2; ++sp)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) xb[sp][dx] = sb + xoff[sp][dx];
      const unsigned char* wbp = WRES ? smem + WRES_OFF + cc * 9 * WR * 64 + (woff - UD_INROWS * 64) : sb + woff;
      if constexpr (SCHED == 0) {
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int dy = tap / 3, dx = tap % 3;
        v8 xf[4], wf[NN];
#pragma unroll
        for (int m = 0; m < 4; ++m) { const int s = m + dy; xf[m] = *(const v8*)(xb[s & 1][dx] + (s & ~1) * UD_PW * 64); }
#pragma unroll
        for (int n = 0; n < NN; ++n) wf[n] = *(const v8*)(wbp + (tap * WR + n * 16) * 64);
        if (UD_HAS_NEXT) { if (tap < NPIECE) piece(tap); if (tap == 8 && NPIECE == 10) piece(9); }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int n = 0; n < NN; ++n) acc[m][n] = E16<ET>::mfma(wf[n], xf[m], acc[m][n]);
      }
      } else {
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        v8 xq[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) xq[s] = *(const v8*)(xb[s & 1][dx] + (s & ~1) * UD_PW * 64);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
          const int tap = dy * 3 + dx, slot = dx * 3 + dy;
          v8 wf[NN];
#pragma unroll
          for (int n = 0; n < NN; ++n) {
            if constexpr (WRES == 2) wf[n] = wreg[tap][n & 1];
            else wf[n] = *(const v8*)(wbp + (tap * WR + n * 16) * 64);
          }
#ifdef SH_DMA_EARLY
          // experiment: two pieces per tap in the first taps of the step, so the last piece has half a step to land
          if (UD_HAS_NEXT) { if (2 * slot < NPIECE) piece(2 * slot); if (2 * slot + 1 < NPIECE) piece(2 * slot + 1); }
#else
          if (UD_HAS_NEXT) { if (slot < NPIECE) piece(slot); if (slot == 8 && NPIECE == 10) piece(9); }
#endif
#pragma unroll
          for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int n = 0; n < NN; ++n) acc[m][n] = E16<ET>::mfma(wf[n], xq[m + dy], acc[m][n]);
        }
      }
      }
      buf ^= 1;
    }
    if (FUSE & UF_HEAD) {      // logit = head_b + sum over the 32 channels of relu(conv) * head_w: 8 in the lane, the rest in lanes li + 16 k
      float* lo = logits + (size_t)c_img * H * W;
      const float hb = head_b[0];
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float sacc = 0.0f;      // same operation order as the UF_HEAD epilogue of k_conv_mfma_bf16
#pragma unroll
        for (int n = 0; n < NN; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc = __builtin_fmaf(fmaxf(acc[m][n][r], 0.0f), s_bias[256 + 8 * lk + 4 * n + r], sacc);
        sacc += __shfl_xor(sacc, 16);
        sacc += __shfl_xor(sacc, 32);
        // all four lanes of a pixel store the same value: the store count per wave stays exact
        ud_store4(lo + (size_t)(c_y0 + rg * 4 + m) * W + c_x0 + xh * 16 + li, hb + sacc);
      }
    } else {
    ET* out = dst + (size_t)c_img * H * W * Cout;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int gy = c_y0 + rg * 4 + m, gx = c_x0 + xh * 16 + li;
#pragma unroll
      for (int h = 0; h < NN / 2; ++h) {
        v8 o;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
          float v = acc[m][2 * h + (r >> 2)][r & 3];
          if (relu) v = fmaxf(v, 0.0f);
          o[r] = (ET)v;
        }
        ud_store16(out + act_off((size_t)H * W, (size_t)gy * W + gx, c_n0 + 4 * NN * lk + 8 * h), o);
      }
    }
    }
    if (FUSE & UF_POOL) {      // 2x2 max pool of this wave's 4 rows x 16 pixels (rows pair inside the lane, columns with lane li ^ 1)
      ET* po = pooled + (size_t)c_img * (H / 2) * (W / 2) * Cout;
#pragma unroll
      for (int mp = 0; mp < 2; ++mp)
#pragma unroll
        for (int h = 0; h < NN / 2; ++h) {
          v8 o;
#pragma unroll
          for (int r = 0; r < 8; ++r) {
            float v = fmaxf(acc[2 * mp][2 * h + (r >> 2)][r & 3], acc[2 * mp + 1][2 * h + (r >> 2)][r & 3]);
            v = fmaxf(v, __shfl_xor(v, 1));
            if (relu) v = fmaxf(v, 0.0f);
            o[r] = (ET)v;
          }
          // odd lanes store too (same value, the pixel of their even neighbour): the store count per wave stays exact
          ud_store16(po + act_off((size_t)(H / 2) * (W / 2), (size_t)((c_y0 + rg * 4) / 2 + mp) * (W / 2) + (c_x0 + xh * 16 + li) / 2, c_n0 + 4 * NN * lk + 8 * h), o);
        }
    }
    (void)NSTORE;
    stores_in_flight = true;
    if (!more) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace sh
