// k_unet16_occ.h -- dec0b (32 -> 32 channels, 3x3) with the 1x1 head in its epilogue, as SMALL workgroups that share a CU.
//
// k_conv3_dma16<.., UF_HEAD, 2, 1, 2> runs this layer as one 8-wave workgroup per CU on 32 x 16 tiles: every wave multiplies
// (72 MFMAs), then every wave runs its ~50-instruction head epilogue with two ds_bpermute round trips, then every wave waits at
// the step barrier -- in lockstep (matrix pipe busy 0.26, 0.47 ms).  This is the other way to cover the phases: independent 4-wave
// workgroups on 32 x 8 tiles, two per CU, whose epilogues and LDS-DMA waits the hardware interleaves with each other's MFMAs.
// MEASURED (round 3): tools/probes/occ_probe.hip, the same structure run alone in a loop, needs 0.36 ms per layer -- but inside the
// network, where the layer follows dec0a_up at the clock the chip holds there, this kernel needs 0.50 ms against the 8-wave
// kernel's 0.48 (tickets of 4 / 16 / 64 tiles, dealt round robin, border pixels clamped instead of zero-paged: 0.48-0.51 all).
// The layer is bound by feeding 41 KB of halo per 576 MFMAs either way.  NOT the default: SHOULDER_DEC0B_OCC=1 selects it.
//   workgroup = 4 waves (2 x 2: 4 rows x 16 pixels x 32 couts each = 32 accumulators), tile 32 x 8 pixels;
//   LDS: two halo buffers of 10 x 36 pixels x 64 B (23 040 B each); the 18 weight fragments of a lane stay in registers (72 VGPRs),
//   read once from the packed weights; every wave issues its 6 LDS-DMA pieces of the NEXT tile right after the barrier;
//   work: tickets of SH_OCC_TK consecutive tiles from a global counter, the id of the next ticket fetched one ticket ahead.
// Same arithmetic as the 8-wave kernel, operation for operation (bias in the accumulator, taps dx-major, head sum in the lane then
// over the four lane groups): bit-identical logits (tests/test_gpu_unet_bf16.py::test_small_workgroup_dec0b_bit_identical).
#pragma once
#include "k_unet_bf16_dma.h"

namespace sh {

#define SH_OCC_THREADS 256
#define SH_OCC_TR 8                                   // tile rows
#define SH_OCC_INROWS ((SH_OCC_TR + 2) * UD_PW)       // 360 halo pixels
#define SH_OCC_BUF (SH_OCC_INROWS * 64)               // 23 040
#define SH_OCC_TK 16                                  // tiles per ticket

template <int EK>
__global__ void __launch_bounds__(SH_OCC_THREADS)
k_dec0b_head_occ(const u16* __restrict__ src_ /*[img][H W][32]*/, const u16* __restrict__ wgt_ /*packed [9][1][32][32]*/, const float* __restrict__ bias,
                 const float* __restrict__ head_w, const float* __restrict__ head_b, float* __restrict__ logits, int H, int W, int nimg,
                 const u16* __restrict__ zero_page_, unsigned* __restrict__ ticket /*zero at launch*/) {
  using ET = typename EKT<EK>::type;
  using v8 = typename E16<ET>::v8;
  const ET* src = (const ET*)src_;
  const ET* wgt = (const ET*)wgt_;
  const ET* zero_page = (const ET*)zero_page_;
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * SH_OCC_BUF];
  __shared__ int s_q[2];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int li = lane & 15, lk = lane >> 4;
  const int xh = wave & 1, rg = wave >> 1;
  const int tiles_x = W / 32, tiles_y = H / SH_OCC_TR;
  const int total = nimg * tiles_x * tiles_y;
  const int ntk = (total + SH_OCC_TK - 1) / SH_OCC_TK;
  if (tid == 0) { s_q[0] = (int)atomicAdd(ticket, 1u); s_q[1] = (int)atomicAdd(ticket, 1u); }

  // weight fragments: MFMA row i of tile n holds output channel 8 (i >> 2) + 4 n + (i & 3) (the dealing of k_conv3_dma16 at NN = 2)
  v8 wreg[9][2];
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int n = 0; n < 2; ++n) wreg[tap][n] = *(const v8*)(wgt + ((size_t)tap * 32 + 8 * (li >> 2) + 4 * n + (li & 3)) * 32 + 8 * lk);
  f32x4 bv[2];
  float hw[8];
#pragma unroll
  for (int n = 0; n < 2; ++n)
#pragma unroll
    for (int r = 0; r < 4; ++r) { bv[n][r] = bias[8 * lk + 4 * n + r]; hw[4 * n + r] = head_w[8 * lk + 4 * n + r]; }
  const float hb = head_b[0];

  // fragment read offsets (bytes inside a halo buffer)
  int xoff[2][3];
  {
    const int rowbase = rg * 4 * UD_PW + xh * 16 + li;
#pragma unroll
    for (int sp = 0; sp < 2; ++sp)
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) xoff[sp][dx] = UB_OFF(rowbase + sp * UD_PW + dx, lk) * 2;
  }
  // staging plan: piece k of this lane = LDS row r0 + 64 k (halo pixel (r / 36, r % 36)), 16-byte slot tid & 3 (swizzled on the source)
  const int r0 = tid >> 2;
  const int q8 = ((tid & 3) ^ ((r0 >> 1) & 2)) * 8;
  int hpy[6], hpx[6];
#pragma unroll
  for (int k = 0; k < 6; ++k) { const int r = r0 + 64 * k; hpy[k] = r / UD_PW; hpx[k] = r - hpy[k] * UD_PW; }
  auto stage = [&](int t, int buf) {      // all pieces of tile t -> halo buffer buf
    const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y);
    const ET* simg = src + (size_t)img * H * W * 32;
    unsigned char* lbase = smem + buf * SH_OCC_BUF + wave * 1024;
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (r0 + 64 * k < SH_OCC_INROWS) {
        const int gx = tx * 32 + hpx[k] - 1, gy = ty * SH_OCC_TR + hpy[k] - 1;
        const bool ok = hpx[k] < 34 && gx >= 0 && gx < W && gy >= 0 && gy < H;
        const ET* p = ok ? simg + ((size_t)(gy * W + gx) * 32 + q8) : zero_page;
        __builtin_amdgcn_global_load_lds((ud_gptr)p, (ud_lptr)(lbase + k * 4096), 16, 0, 0);
      }
    }
  };

  __syncthreads();
  int qk = 1;
  int tk = __builtin_amdgcn_readfirstlane(s_q[0]);
  if (tk >= ntk) return;
  int t = tk * SH_OCC_TK, t_end = min(total, t + SH_OCC_TK);
  stage(t, 0);
  int buf = 0;
  for (;;) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of tile t have landed (and its stores of the tile before)
    __builtin_amdgcn_s_barrier();                         // every wave's pieces have landed; every wave is done reading the other buffer
    // the tile after this one
    int nt = t + 1, nt_end = t_end;
    bool more = true;
    if (nt >= t_end) {
      const int ntk_id = __builtin_amdgcn_readfirstlane(s_q[qk]);      // written before this barrier
      if (ntk_id < ntk) {
        if (tid == 0) s_q[qk ^ 1] = (int)atomicAdd(ticket, 1u);      // (read after the NEXT ticket's first barrier)
        qk ^= 1;
        nt = ntk_id * SH_OCC_TK; nt_end = min(total, nt + SH_OCC_TK);
      } else more = false;
    }
    if (more) stage(nt, buf ^ 1);
    const unsigned char* sb = smem + buf * SH_OCC_BUF;
    f32x4 acc[4][2];
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int n = 0; n < 2; ++n) acc[m][n] = bv[n];
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
    {      // logit = head_b + sum over the 32 channels of relu(conv) * head_w: 8 in the lane, the rest in lanes li + 16 k
      const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, img = t / (tiles_x * tiles_y);
      float* lo = logits + (size_t)img * H * W;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        float sacc = 0.0f;      // same operation order as the UF_HEAD epilogue of k_conv3_dma16
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int r = 0; r < 4; ++r) sacc = __builtin_fmaf(fmaxf(acc[m][n][r], 0.0f), hw[4 * n + r], sacc);
        sacc += __shfl_xor(sacc, 16);
        sacc += __shfl_xor(sacc, 32);
        if (lk == 0) lo[(size_t)(ty * SH_OCC_TR + rg * 4 + m) * W + tx * 32 + xh * 16 + li] = hb + sacc;
      }
    }
    if (!more) break;
    t = nt; t_end = nt_end;
    buf ^= 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

}  // namespace sh
