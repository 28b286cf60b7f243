// k_unet16_ldr.h -- 3x3 conv, 16-bit, for the layers with >= 64 output channels: persistent workgroups (one per CU, 512 lanes,
// 155 KB of LDS) that walk (image, 32x16-pixel tile, 64-cout group) items handed out in work tickets, with the two jobs of a
// workgroup given to different waves.
//
// A workgroup is 4 COMPUTE waves + 4 LOADER waves (one of each per SIMD):
//   compute wave (cw = 0..3): 8 rows x 16 pixels x 64 couts of the 32 x 16 tile = 128 accumulator registers, 288 MFMAs per
//     32-channel step and nothing else but the fragment reads -- 66 `ds_read_b128` per step (per column offset dx: 10 pixel-row
//     fragments that serve its three dy taps + 12 weight fragments); no vector-memory instruction except the epilogue stores,
//     no vmcnt wait, no address arithmetic per step;
//   loader wave (lw = 0..3): decodes the next step, issues ALL its LDS-DMA pieces (20, or 11 with resident weights) right
//     after the step barrier, waits for them (vmcnt(0)) and joins the next barrier: a step's data has a whole step to land,
//     and the wave is parked -- it takes no issue slot from the compute wave of its SIMD -- for most of it.
// One raw s_barrier per step for the whole workgroup.  LDS image per buffer (two of them): 18 x 36 halo-pixel rows (pitch 36:
// the swizzle bit of a fragment row then depends only on row parity, dx and the lane), then 576 weight rows ([tap][64 couts]);
// 32 channels = 64 B per row, XOR slot swizzle applied on the DMA's SOURCE address (the DMA writes lane-linear) and on the
// fragment read.  Out-of-image halo pixels read a 64-byte page of zeros.  Output channels of an item are dealt to the
// accumulator tiles so that a lane ends up with CONSECUTIVE channels (weight row 16 n + i of the LDS image holds channel
// 16 (i >> 2) + 4 n + (i & 3) of the group): the epilogue stores 16 bytes per instruction.
// The pieces are inline assembly (k_unet16_base.h): through the builtin hipcc drains every vector-memory operation of a wave in
// front of its first LDS read -- for a compute wave the stores of the item it has just finished.
// Work tickets: runs of items of DECREASING length (tk_tab, built on the host) are handed out by a global counter, so a workgroup
// that starts late -- its CU was still held by a kernel of the engine's other lane -- takes less and the layer ends when the chip
// runs out of work; the id of the ticket after the current one is always already in LDS (fetched one ticket ahead).
#pragma once
#include "k_unet_bf16.h"

namespace sh {

#define UD_THREADS 512
#define UD_ROWS (UD_INROWS + 576)           // 1224 LDS rows of 64 B per buffer: 648 halo rows + [9 taps][64 couts] weight rows
#define UD_BUF (UD_ROWS * 64)               // 78336
#define UD_SLOTS (UD_ROWS * 4)              // 4896
#define UD_BIAS_OFF (2 * UD_BUF)
#define UD_SMEM (2 * UD_BUF + 2048)         // 158720
#define UL_NCW 4                          // compute waves; waves UL_NCW .. 7 load
#define UL_LTHREADS 256                   // lanes of the loader half

// The second source as an UP-CONVOLUTION computed in place (UPL = chunks of the low-resolution input, 0 = off): the decoder's
// conv reads concat(skip, up(low)); instead of fetching the "up" chunks of a halo tile from a tensor that a launch of its own wrote
// (0.54 GB out, 0.54 GB back in at level 1), the four LOADER waves -- which hold the registers of a compute wave and use a fifth of
// them -- compute them: wave lw owns output phase (dy, dx) = (lw >> 1, lw & 1), 9 x 17 of the 18 x 34 halo pixels, ten groups of 16;
// pixel fragments (18 x 10 low-resolution pixels x UPL chunks) and weight fragments come straight from L2 into registers, 8 UPL MFMAs
// per group on the matrix pipe the compute wave of the SIMD is using, the rounded result goes to the halo rows the DMA would have
// filled (zeros outside the image).  Same arithmetic in the same order as k_upconv16g (bias in the accumulator, chunks in order), so
// the conv sees the same 16-bit values.  For layers with ONE cout group only (level 1): every group of a tile would compute the chunk again.
struct UpSrc { const u16* low; const u16* w /*packed [4][UPL][C1][32]*/; const float* b; };

// FUSE: 0 or UF_POOL.  WRES: 0 = weights staged with every step; 1 = one cout group whose packed weights fit behind the two
// input buffers (nchunk <= 2): loaded once per workgroup.
template <int EK, int FUSE, int WRES, int UPL = 0>
__global__ void __launch_bounds__(UD_THREADS)
k_conv3_ldr16(const u16* __restrict__ src0_, const u16* __restrict__ src1_, int C0, int C1,
              const u16* __restrict__ wgt_, const float* __restrict__ bias, u16* __restrict__ dst_,
              int H, int W, int Cout, int relu, int nimg, const u16* __restrict__ zero_page_, u16* __restrict__ pooled_,
              unsigned* __restrict__ ticket /*zero at launch*/, const int* __restrict__ tk_tab /*[ntk + 1] item bounds*/, int ntk, const UpSrc up) {
  using ET = typename EKT<EK>::type;
  const ET* src0 = (const ET*)src0_;
  const ET* src1 = (const ET*)src1_;
  const ET* wgt = (const ET*)wgt_;
  ET* dst = (ET*)dst_;
  const ET* zero_page = (const ET*)zero_page_;
  ET* pooled = (ET*)pooled_;
  using v8 = typename E16<ET>::v8;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[UD_SMEM];
  __shared__ int s_q[2];
  constexpr int WR = 64;                                               // weight rows per tap
  constexpr int SLOTS = WRES ? UD_INROWS * 4 : UD_SLOTS;               // 16-byte slots of a step: 2592 / 4896
  constexpr int NPIECE = (SLOTS + UL_LTHREADS - 1) / UL_LTHREADS;      // 11 / 20 pieces per loader wave and step
  constexpr int NHALO = (UD_INROWS * 4 + UL_LTHREADS - 1) / UL_LTHREADS;      // 11: pieces 0..10 carry halo rows (10: the last 8 rows)
  constexpr int BUFB = WRES ? UD_INROWS * 64 : UD_BUF;
  constexpr int WRES_OFF = 2 * UD_INROWS * 64;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const bool loader = wave >= UL_NCW;
  const int li = lane & 15, lk = lane >> 4;
  const int Cin = C0 + C1, nchunk = Cin >> 5;
  const int tiles_x = W / 32, tiles_y = H / 16, ngroups = Cout / WR;
  if (tid == UL_LTHREADS) { s_q[0] = ud_take_ticket(ticket); s_q[1] = ud_take_ticket(ticket); }

  float* s_bias = (float*)(smem + UD_BIAS_OFF);
  for (int i = tid; i < Cout; i += UD_THREADS) s_bias[i] = bias[i];
  if constexpr (WRES != 0) {
    // LDS row (chunk, tap, 16 n + i) <- packed row (tap, chunk, channel 16 (i >> 2) + 4 n + (i & 3)); the 16-byte slot swizzle on the source
    const int nrows = nchunk * 9 * WR;
    for (int e = tid; e < nrows * 4; e += UD_THREADS) {
      const int row = e >> 2, q = e & 3;
      const int cc = row / (9 * WR), rem = row - cc * 9 * WR, tap = rem >> 6, j = rem & (WR - 1);
      const int ch = 16 * ((j & 15) >> 2) + 4 * (j >> 4) + (j & 3);
      *(u32x4*)(smem + WRES_OFF + e * 16) = *(const u32x4*)(wgt + (size_t)((tap * nchunk + cc) * Cout + ch) * 32 + ((q ^ ((row >> 1) & 2)) << 3));
    }
  }
  __syncthreads();       // every ordinary load is retired before the first LDS-DMA is issued
  int qk = 1;
  int w_begin, w_end;
  {
    const int t0 = __builtin_amdgcn_readfirstlane(s_q[0]);
    if (t0 >= ntk) return;
    w_begin = tk_tab[t0]; w_end = tk_tab[t0 + 1];
  }

  // ---- loader lanes: staging plan.  Slot e_k = ltid + 256 k -> LDS row r_k = (ltid >> 2) + 64 k; rows < 648 are halo pixels
  // (row = py * 36 + px), rows 648 + 64 tap + j weight row j of tap `tap`.  The swizzle bit (bit 2 of the row) is the same
  // for every k.
  const int ltid = tid - UL_LTHREADS, lw = wave - UL_NCW;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(ud_lptr)smem);
  const int r0 = (ltid >> 2) & 63;
  const int q8 = ((ltid & 3) ^ ((r0 >> 1) & 2)) * 8;
  // (halo pixel of piece k: row r0 + 64 k = py * 36 + px, px >= 34: padding column -- recomputed per item: two registers per piece
  //  kept for the whole launch pushed the pooled instantiation into scratch)
  // weight rows: lanes r0 >= 8 hold row j = r0 - 8 of tap k - 10 in piece k, lanes r0 < 8 row j = r0 + 56 of tap k - 11
  const bool wlow = r0 < 8;                                          // (these lanes carry the last 8 halo rows in piece 10)
  const int wj = wlow ? r0 + 56 : r0 - 8;
  const int wch = 16 * ((wj & 15) >> 2) + 4 * (wj >> 4) + (wj & 3);
  const int wtap_stride = nchunk * Cout * 32;                        // elements between two taps of the packed weights
  const int wlane = (wlow ? -11 : -10) * wtap_stride + wch * 32 + q8;

  int i_g, i_tx, i_ty, i_img;      // item being staged
  auto decode = [&](int w) {      // cout groups of one tile are neighbours: their input tile stays in L2
    i_g = w % ngroups; w /= ngroups;
    i_tx = w % tiles_x; w /= tiles_x;
    i_ty = w % tiles_y; i_img = w / tiles_y;
  };
  decode(w_begin);
  i_g = __builtin_amdgcn_readfirstlane(i_g); i_tx = __builtin_amdgcn_readfirstlane(i_tx);
  i_ty = __builtin_amdgcn_readfirstlane(i_ty); i_img = __builtin_amdgcn_readfirstlane(i_img);
  // The pixel a halo piece fetches: eleven offsets per lane, set up once per item (pixoff).  In the instantiation with the fused pool
  // and streamed weights (LAZY) they are recomputed where the piece is issued instead: there the eleven registers -- and what hipcc
  // hoists around them -- pushed the kernel into scratch, with a vmcnt(0) behind every reload between two LDS-DMA pieces (enc2b 0.26 ->
  // 0.48 ms).  Everywhere else the recomputation costs more than it frees (66 vector instructions per step beside the compute wave
  // of the SIMD: the 3x3 layers 5-12 % slower).
  constexpr bool LAZY = (FUSE & UF_POOL) != 0 && WRES == 0;
  auto halo_pixel = [&](int r0v, int k) -> int {      // image pixel index of halo row r0 + 64 k of the item being staged, -1: zero padding
    const int r = r0v + 64 * k, hpy = r / UD_PW, hpx = r - hpy * UD_PW;
    const int gx = i_tx * 32 + hpx - 1, gy = i_ty * 16 + hpy - 1;
    return (hpx < 34 && gx >= 0 && gx < W && gy >= 0 && gy < H) ? gy * W + gx : -1;
  };
  int pixoff[LAZY ? 1 : NHALO];
  auto item_lane_setup = [&]() {
    if constexpr (!LAZY) {
#pragma unroll
      for (int k = 0; k < NHALO; ++k) pixoff[k] = halo_pixel(r0, k);
    }
  };
  auto stage = [&](int cc, int buf) {      // all pieces of step (current item, chunk cc) -> buffer buf; loader waves only
    const int c0 = cc * 32;
    const bool first = c0 < C0;
    const int Cs = first ? C0 : C1;
    const int cb = ((first ? c0 : c0 - C0) >> 5) * (H * W);      // pixel offset of this chunk's 32-channel plane (channel-blocked activations)
    const ET* simg = (first ? src0 : src1) + (size_t)i_img * H * W * Cs;
    const ET* wbase = wgt + ((size_t)cc * Cout + i_g * WR) * 32;
    const unsigned lbase = lds0 + (unsigned)(buf * BUFB + lw * 1024);      // (LDS byte address: the pieces are inline assembly, k_unet16_base.h)
    int r0v = r0;
    if constexpr (LAZY) asm volatile("" : "+v"(r0v));      // (opaque per call: nothing of the recomputation is hoisted out of the loop)
    const bool upstep = UPL != 0 && !first;      // the halo rows of this step are computed (up_chunk below), only its weights are fetched
#pragma unroll
    for (int k = 0; k < NPIECE; ++k) {
      if (k < NHALO - 1) {
        if (!upstep) {
          const int po = LAZY ? halo_pixel(r0v, k) : pixoff[LAZY ? 0 : k];
          const ET* p = po >= 0 ? simg + (unsigned)((cb + po) * 32 + q8) : zero_page;
          ud_dma16(lbase + k * 4096, p);
        }
      } else if (k == NHALO - 1) {      // rows 640..703: 8 halo rows, then the first 56 weight rows
        const int po = LAZY ? halo_pixel(r0v, k) : pixoff[LAZY ? 0 : k];
        const ET* pi = po >= 0 ? simg + (unsigned)((cb + po) * 32 + q8) : zero_page;
        if constexpr (WRES != 0) {
          if (wlow && !upstep) ud_dma16(lbase + k * 4096, pi);
        } else {
          const ET* p = wlow ? pi : wbase + (wlane + k * wtap_stride);
          if (!(wlow && upstep)) ud_dma16(lbase + k * 4096, p);
        }
      } else if (k < NPIECE - 1) {
        ud_dma16(lbase + k * 4096, (wbase + (wlane + k * wtap_stride)));
      } else {                          // the last piece: rows 1216..1223 only (tap 8, rows 56..63)
        if (wlow) ud_dma16(lbase + k * 4096, (wbase + (wlane + k * wtap_stride)));
      }
    }
    if constexpr (UPL != 0) {
      if (upstep) {
        // ---- the halo rows of up chunk u = 32 channels of up(low): this wave's phase, ten groups of 16 pixels
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        using v2 = typename E16<ET>::v2;
        const int u = (c0 - C0) >> 5;
        const int H2 = H >> 1, W2 = W >> 1;
        const int pdy = lw >> 1, pdx = lw & 1, oy = pdy == 0 ? 1 : 0, ox = pdx == 0 ? 1 : 0;
        const ET* limg = (const ET*)up.low + (size_t)i_img * H2 * W2 * (UPL * 32);
        const ET* uwp = (const ET*)up.w + ((size_t)(lw * UPL) * C1 + u * 32 + 8 * (li >> 2) + (li & 3)) * 32 + 8 * lk;
        v8 uw[UPL][2];
#pragma unroll
        for (int kc = 0; kc < UPL; ++kc)
#pragma unroll
          for (int n = 0; n < 2; ++n) uw[kc][n] = *(const v8*)(uwp + ((size_t)kc * C1 + 4 * n) * 32);
        f32x4 ubv[2];
#pragma unroll
        for (int n = 0; n < 2; ++n) ubv[n] = *(const f32x4*)(up.b + u * 32 + 8 * lk + 4 * n);
        unsigned char* hb = smem + buf * BUFB;
        const int ly0 = i_ty * 8 - 1 + oy, lx0 = i_tx * 16 - 1 + ox;
        // a rolling window of five groups' pixel fragments: the loads of group g + 5 go out behind the MFMAs of group g, so a step
        // pays one L2 latency, not one per batch (all ten at once do not fit the loader's registers)
        constexpr int WIN = 5;
        v8 xf[WIN][UPL];
        bool inside[WIN];
        auto request = [&](int g) __attribute__((always_inline)) {
          const int t = 16 * g + li, a = t / 17, bc = t - a * 17;
          const int ly = ly0 + a, lx = lx0 + bc;
          inside[g % WIN] = ly >= 0 && ly < H2 && lx >= 0 && lx < W2;
          const unsigned po = (unsigned)(min(max(ly, 0), H2 - 1) * W2 + min(max(lx, 0), W2 - 1));      // (unconditional, clamped: a load in a branch is waited for at its join)
#pragma unroll
          for (int kc = 0; kc < UPL; ++kc) xf[g % WIN][kc] = *(const v8*)(limg + ((size_t)kc * H2 * W2 + po) * 32 + 8 * lk);
        };
#pragma unroll
        for (int g = 0; g < WIN; ++g) request(g);
#pragma unroll
        for (int g = 0; g < 10; ++g) {
          f32x4 ua[2];
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            ua[n] = ubv[n];
#pragma unroll
            for (int kc = 0; kc < UPL; ++kc) ua[n] = E16<ET>::mfma(uw[kc][n], xf[g % WIN][kc], ua[n]);
          }
          const bool in_g = inside[g % WIN];
          if (g + WIN < 10) request(g + WIN);
          u32x4 o;
#pragma unroll
          for (int n = 0; n < 2; ++n) {
            const f32x2 a01 = {ua[n][0], ua[n][1]}, a23 = {ua[n][2], ua[n][3]};
            o[2 * n] = __builtin_bit_cast(unsigned, __builtin_convertvector(a01, v2));
            o[2 * n + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(a23, v2));
          }
          if (!in_g) o = u32x4{0u, 0u, 0u, 0u};      // outside the image: the conv's zero padding
          const int t = 16 * g + li, a = t / 17, bc = t - a * 17;
          if (t < 153) *(u32x4*)(hb + UB_OFF((2 * a + oy) * UD_PW + 2 * bc + ox, lk) * 2) = o;
        }
      }
    }
  };

  // ---- compute lanes: fragment read offsets (bytes inside a buffer)
  const int xh = wave & 1, rg8 = (wave >> 1) & 1;
  int xoff[2][3], woff;
  {
    const int rowbase = rg8 * 8 * UD_PW + xh * 16 + li;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) xoff[sp][dx] = UB_OFF(rowbase + sp * UD_PW + dx, lk) * 2;
    woff = UB_OFF(UD_INROWS + li, lk) * 2;
  }

  // ---- the walk: the step after (this item, chunk cc).  Every wave keeps it (the compute waves need the item's coordinates for their
  // epilogue); the same in every lane: say so, or hipcc keeps it -- and the 64-bit source addresses it feeds -- in vector registers,
  // which pushed the pooled instantiation into scratch, with a vmcnt(0) behind every reload between two LDS-DMA pieces
  int w = w_begin;
  auto walk = [&](int cc, int& n_cc, bool& has_next, bool& new_item, bool& more) __attribute__((always_inline)) {
    n_cc = cc + 1;
    has_next = true; new_item = false;
    if (n_cc < nchunk) {
    } else if (w + 1 < w_end) {
      ++w;
      if (++i_g == ngroups) { i_g = 0; if (++i_tx == tiles_x) { i_tx = 0; if (++i_ty == tiles_y) { i_ty = 0; ++i_img; } } }
      n_cc = 0; new_item = true;
    } else {      // next ticket: its id was written before this step's barrier; the slot it frees is refilled for the one after
      const int nt = __builtin_amdgcn_readfirstlane(s_q[qk]);
      if (nt < ntk) {
        if (tid == UL_LTHREADS) s_q[qk ^ 1] = ud_take_ticket(ticket);
        qk ^= 1;
        w = tk_tab[nt]; w_end = tk_tab[nt + 1];
        decode(w);
        n_cc = 0; new_item = true;
      } else { has_next = false; more = false; }
    }
    i_g = __builtin_amdgcn_readfirstlane(i_g); i_tx = __builtin_amdgcn_readfirstlane(i_tx);
    i_ty = __builtin_amdgcn_readfirstlane(i_ty); i_img = __builtin_amdgcn_readfirstlane(i_img);
    w = __builtin_amdgcn_readfirstlane(w); w_end = __builtin_amdgcn_readfirstlane(w_end);
    n_cc = __builtin_amdgcn_readfirstlane(n_cc);
  };
  // ---- a compute wave's step: per column offset the 10 pixel rows, per tap the 4 weight fragments (compiler-scheduled)
  auto multiply = [&](f32x4 (&acc)[8][4], int cc, int buf) __attribute__((always_inline)) {
    const unsigned char* sb = smem + buf * BUFB;
    const unsigned char* wbp = WRES ? smem + WRES_OFF + cc * 9 * WR * 64 + (woff - UD_INROWS * 64) : sb + woff;
#pragma unroll
    for (int dx = 0; dx < 3; ++dx) {
      v8 xq[10];
#pragma unroll
      for (int s = 0; s < 10; ++s) xq[s] = *(const v8*)(sb + xoff[s & 1][dx] + (s & ~1) * UD_PW * 64);
#pragma unroll
      for (int dy = 0; dy < 3; ++dy) {
        const int tap = dy * 3 + dx;
        v8 wf[4];
#pragma unroll
        for (int n = 0; n < 4; ++n) wf[n] = *(const v8*)(wbp + (tap * WR + n * 16) * 64);
#pragma unroll
        for (int m = 0; m < 8; ++m)
#pragma unroll
          for (int n = 0; n < 4; ++n) acc[m][n] = E16<ET>::mfma(wf[n], xq[m + dy], acc[m][n]);
      }
    }
  };
  // ---- a compute wave's epilogue of the item at (c_img, c_y0, c_x0), cout group c_n0
  auto epilogue = [&](const f32x4 (&acc)[8][4], int c_x0, int c_y0, int c_img, int c_n0) __attribute__((always_inline)) {
    // channel-blocked output: channel c_n0 + 16 lk + 8 h of pixel (gy, gx) is element ((c >> 5) HW + pix) 32 + (c & 31); one
    // uniform 64-bit base per item and one 32-bit lane offset, the rest of every address is a constant
    const unsigned HW = (unsigned)(H * W);
    ET* ob = dst + ((size_t)c_img * Cout + c_n0) * HW;
    const unsigned lo = (((unsigned)(lk >> 1) * HW + (unsigned)((c_y0 + rg8 * 8) * W + c_x0 + xh * 16 + li)) << 5) + 16u * (lk & 1);
    ET* pb = pooled;
    unsigned plo = 0;
    if (FUSE & UF_POOL) {
      pb = pooled + ((size_t)c_img * Cout + c_n0) * (HW >> 2);
      plo = (((unsigned)(lk >> 1) * (HW >> 2) + (unsigned)(((c_y0 + rg8 * 8) >> 1) * (W >> 1) + ((c_x0 + xh * 16 + li) >> 1))) << 5) + 16u * (lk & 1);
    }
#pragma unroll
    for (int mp = 0; mp < 4; ++mp)
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        // Rounded in pairs (one v_cvt_pk per two values), then ReLU on the rounded 16-bit values, two per instruction: a negative
        // bf16 / f16 is a negative int16 (sign bit), so max(bits, 0) as packed int16 is max(x, +0.0) -- rounding is monotonic and keeps
        // the sign, -0.0 becomes +0.0 either way: the same bits as fmaxf on the f32 accumulators followed by the conversion (NaN aside,
        // which no layer produces from finite input).  Without ReLU the max is against the smallest int16: the identity.  (fmaxf is two
        // v_max_f32 per value -- it quiets NaNs first -- and a `relu` flag tested per value a select: 380 instead of 128 instructions.)
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        using v2 = typename E16<ET>::v2;
        const unsigned zsel = relu ? 0u : 0x80008000u;
        u32x4 ua, ub;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const f32x4& ra = acc[2 * mp][2 * h + (i >> 1)];
          const f32x4& rb = acc[2 * mp + 1][2 * h + (i >> 1)];
          const f32x2 pa = {ra[2 * (i & 1)], ra[2 * (i & 1) + 1]}, pb2 = {rb[2 * (i & 1)], rb[2 * (i & 1) + 1]};
          ua[i] = pp_pkmax(__builtin_bit_cast(unsigned, __builtin_convertvector(pa, v2)), zsel);
          ub[i] = pp_pkmax(__builtin_bit_cast(unsigned, __builtin_convertvector(pb2, v2)), zsel);
        }
        const v8 oa = __builtin_bit_cast(v8, ua), ob8 = __builtin_bit_cast(v8, ub);
        *(v8*)(ob + (lo + (unsigned)((2 * mp) * W * 32 + 8 * h))) = oa;
        *(v8*)(ob + (lo + (unsigned)((2 * mp + 1) * W * 32 + 8 * h))) = ob8;
        if (FUSE & UF_POOL) {
          // 2x2 max on the rounded, ReLU'd values (non-negative: they order like int16; the host fuses the pool behind a ReLU only):
          // rows inside the lane, columns with lane ^ 1
          u32x4 pv;
#pragma unroll
          for (int i = 0; i < 4; ++i) pv[i] = pp_pkmax_lane1(pp_pkmax(ua[i], ub[i]));      // (one dword at a time: the batched form's temporaries spilled here)
          const v8 op = __builtin_bit_cast(v8, pv);
          if (!(li & 1)) *(v8*)(pb + (plo + (unsigned)(mp * (W >> 1) * 32 + 8 * h))) = op;
        }
      }
  };

  if constexpr (UPL != 0) {
    // The two jobs as two loops (the same walk, the same barriers): with one loop for both, the compute wave's 128 accumulators are
    // live in the loader's branch as well, and the fragments of the up-convolution do not fit beside them.
    if (loader) {
      item_lane_setup();
      stage(0, 0);
      int buf = 0;
      for (;;) {
        bool more = true;
        for (int cc = 0; cc < nchunk; ++cc) {
          asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's pieces of the step have landed, what it computed of them is written
          __builtin_amdgcn_s_barrier();
          int n_cc; bool has_next, new_item;
          walk(cc, n_cc, has_next, new_item, more);
          if (has_next) {
            if (new_item) item_lane_setup();
            stage(n_cc, buf ^ 1);
          }
          buf ^= 1;
        }
        if (!more) break;
      }
    } else {
      int buf = 0;
      for (;;) {
        bool more = true;
        const int c_x0 = i_tx * 32, c_y0 = i_ty * 16, c_img = i_img, c_n0 = i_g * WR;
        f32x4 acc[8][4];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
          const f32x4 bv = *(const f32x4*)(s_bias + c_n0 + 16 * lk + 4 * n);
#pragma unroll
          for (int m = 0; m < 8; ++m) acc[m][n] = bv;
        }
        for (int cc = 0; cc < nchunk; ++cc) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          __builtin_amdgcn_s_barrier();
          int n_cc; bool has_next, new_item;
          walk(cc, n_cc, has_next, new_item, more);
          multiply(acc, cc, buf);
          buf ^= 1;
        }
        epilogue(acc, c_x0, c_y0, c_img, c_n0);
        if (!more) break;
      }
    }
    return;
  }

  if (loader) {
    item_lane_setup();
    stage(0, 0);
  }
  int buf = 0;
  for (;;) {
    bool more = true;
    const int c_x0 = i_tx * 32, c_y0 = i_ty * 16, c_img = i_img, c_n0 = i_g * WR;
    f32x4 acc[8][4];
    if (!loader) {
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const f32x4 bv = *(const f32x4*)(s_bias + c_n0 + 16 * lk + 4 * n);
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m][n] = bv;
      }
    }
    for (int cc = 0; cc < nchunk; ++cc) {
      if (loader) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of the step have landed
      else asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();      // every loader's pieces have landed; every compute wave is done reading the other buffer
      int n_cc; bool has_next, new_item;
      walk(cc, n_cc, has_next, new_item, more);
      if (loader) {
        if (has_next) {
          if (new_item) item_lane_setup();
          stage(n_cc, buf ^ 1);
        }
      } else multiply(acc, cc, buf);
      buf ^= 1;
    }
    if (!loader) epilogue(acc, c_x0, c_y0, c_img, c_n0);
    if (!more) break;
  }
}


}  // namespace sh
