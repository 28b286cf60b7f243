// k_stl.h -- binary STL ingest on the device (`trimesh.load_mesh(stl, process=True)` at reference
// src/shoulder/humerus/mesh.py:22-27 with the canonical rule of oracle/stl.py and shoulder_amd/stl.py: vertices with
// equal bit patterns (after -0.0 -> +0.0) merge, merged vertices are numbered by first appearance in the file,
// triangles that use a vertex twice are dropped, kept triangles stay in file order).
//   k_stl_corners   one lane per triangle corner: 50-byte records (2-byte aligned) -> float32 xyz
//   k_stl_hash      open-addressing table per mesh keyed by the 96-bit pattern; entry = (owner corner, first corner)
//   stl_rank_launch six passes over blocks of 256 corners / triangles: "this corner is its vertex's first appearance" counted per
//                   block, block counts scanned per mesh, ranks -> vertex ids; per-triangle keep flags likewise -> positions; counts V, F
//   k_stl_emit      verts / faces at their batch offsets
// Integer work end to end: the result is bit-identical to the host routine (tests/test_gpu_stl.py).
#pragma once
#include "sh_common.h"

namespace sh {


#define SH_STL_SCAN_THREADS 1024      // (k_clip.h's one-workgroup scans)

__device__ inline unsigned stl_hash3(unsigned x, unsigned y, unsigned z) {
  unsigned long long k = ((unsigned long long)x << 32 | y) * 0x9E3779B97F4A7C15ull ^ ((unsigned long long)z * 0xC2B2AE3D27D4EB4Full);
  k ^= k >> 29; k *= 0xBF58476D1CE4E5B9ull; k ^= k >> 32;
  return (unsigned)k;
}

__global__ void k_stl_corners(const unsigned char* __restrict__ raw, const long long* __restrict__ file_off, const long long* __restrict__ coff,
                              float* __restrict__ corners, int* __restrict__ nonfinite /* [B]: set when a coordinate is NaN / inf */) {
  const int b = blockIdx.y;
  const long long c0 = coff[b], n = coff[b + 1] - c0;
  const unsigned short* base = (const unsigned short*)(raw + file_off[b] + 84);      // file starts are 4-byte aligned, 84 + 50 t + 12 is even
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const long long t = i / 3;
    const int c = (int)(i - 3 * t);
    const unsigned short* p = base + (25 * t + 6 + 6 * c);      // byte 50 t + 12 + 12 c
    float* o = corners + 3 * (c0 + i);
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const unsigned u = (unsigned)p[2 * k] | ((unsigned)p[2 * k + 1] << 16);
      o[k] = __uint_as_float(u) + 0.0f;      // -0.0 -> +0.0: equal values share one bit pattern
      if ((u & 0x7f800000u) == 0x7f800000u) atomicOr(&nonfinite[b], 1);
    }
  }
}

__global__ void k_stl_table_init(int2* __restrict__ table, size_t n) {
  for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) table[i] = make_int2(-1, 0x7fffffff);
}

__global__ void k_stl_hash(const float* __restrict__ corners, const long long* __restrict__ coff, int2* __restrict__ table, int tsize,
                           int* __restrict__ slot_of) {
  const int b = blockIdx.y;
  const long long c0 = coff[b], n = coff[b + 1] - c0;
  const unsigned* K = (const unsigned*)(corners + 3 * c0);
  int2* T = table + (size_t)b * tsize;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const unsigned x = K[3 * i], y = K[3 * i + 1], z = K[3 * i + 2];
    unsigned h = stl_hash3(x, y, z) & (unsigned)(tsize - 1);
    for (;;) {
      const int o = atomicCAS(&T[h].x, -1, (int)i);
      if (o == -1 || (K[3 * (size_t)o] == x && K[3 * (size_t)o + 1] == y && K[3 * (size_t)o + 2] == z)) {
        atomicMin(&T[h].y, (int)i);
        slot_of[c0 + i] = (int)h;
        break;
      }
      h = (h + 1) & (unsigned)(tsize - 1);
    }
  }
}

// exclusive scan of one value per thread over the workgroup (SH_STL_SCAN_THREADS lanes); returns the total through *total (k_clip.h)
__device__ inline int stl_block_scan(int v, int* s_wave /*[16]*/, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int incl = v;
  for (int off = 1; off < 64; off <<= 1) { int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  int base = 0, tot = 0;
  for (int w = 0; w < SH_STL_SCAN_THREADS / 64; ++w) { const int s = s_wave[w]; if (w < wave) base += s; tot += s; }
  __syncthreads();
  *total = tot;
  return base + incl - v;
}

// ---- ranks: vertex ids by first appearance, kept triangles in file order ------------------------------------------------------
// (round 4: was ONE workgroup per mesh walking 96 consecutive corners per lane -- 1.3 ms for a 64-batch on 64 of the 256 CUs, the
// longest kernel of the staged STL path.  Now every pass runs over blocks of SH_STL_BLK corners / triangles with coalesced reads;
// per block a count, per mesh a scan of its block counts, per block the ranks.  Same integers: tests/test_gpu_stl.py.)
#define SH_STL_BLK 256

// exclusive rank of `flag` inside the workgroup (SH_STL_BLK lanes = 4 waves) and the workgroup's total
__device__ inline int stl_blk_rank(bool flag, int* s_w /*[4]*/, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const unsigned long long m = __ballot(flag);
  const int before = __popcll(m & ((1ull << lane) - 1ull));
  if (lane == 0) s_w[wave] = __popcll(m);
  __syncthreads();
  int base = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < SH_STL_BLK / 64; ++w) { const int c = s_w[w]; if (w < wave) base += c; tot += c; }
  __syncthreads();
  *total = tot;
  return base + before;
}

// pass 1 (grid: corner blocks x B): bcnt[b][blk] = first appearances among the block's corners
__global__ void __launch_bounds__(SH_STL_BLK)
k_stl_first_count(const long long* __restrict__ coff, const int2* __restrict__ table, int tsize, const int* __restrict__ slot_of, int* __restrict__ bcnt, int nblk) {
  __shared__ int s_w[SH_STL_BLK / 64];
  const int b = blockIdx.y, blk = blockIdx.x;
  const long long c0 = coff[b];
  const int n = (int)(coff[b + 1] - c0), i = blk * SH_STL_BLK + threadIdx.x;
  if (blk * SH_STL_BLK >= n) return;
  const bool first = i < n && table[(size_t)b * tsize + slot_of[c0 + i]].y == i;
  int tot;
  (void)stl_blk_rank(first, s_w, &tot);
  if (threadIdx.x == 0) bcnt[(size_t)b * nblk + blk] = tot;
}

// passes 2 and 5 (grid: B, one wave each): exclusive scan of a mesh's block counts in place; the total -> counts[2 b + which]
__global__ void k_stl_scan_blocks(const long long* __restrict__ coff, int* __restrict__ bcnt, int nblk, int div /*1: corner blocks, 3: triangle blocks*/,
                                  int* __restrict__ counts, int which) {
  const int b = blockIdx.x, lane = threadIdx.x;
  const int n = (int)((coff[b + 1] - coff[b]) / div), nb = (n + SH_STL_BLK - 1) / SH_STL_BLK;
  int* a = bcnt + (size_t)b * nblk;
  int carry = 0;
  for (int k0 = 0; k0 < nb; k0 += 64) {
    const int k = k0 + lane;
    const int v = k < nb ? a[k] : 0;
    int incl = v;
    for (int off = 1; off < 64; off <<= 1) { const int o = __shfl_up(incl, off); if (lane >= off) incl += o; }
    if (k < nb) a[k] = carry + incl - v;
    carry += __shfl(incl, 63);
  }
  if (lane == 0) counts[2 * b + which] = carry;
}

// pass 3 (grid: corner blocks x B): vid of the first appearances
__global__ void __launch_bounds__(SH_STL_BLK)
k_stl_vid_first(const long long* __restrict__ coff, const int2* __restrict__ table, int tsize, const int* __restrict__ slot_of, const int* __restrict__ bcnt, int nblk,
                int* __restrict__ vid) {
  __shared__ int s_w[SH_STL_BLK / 64];
  const int b = blockIdx.y, blk = blockIdx.x;
  const long long c0 = coff[b];
  const int n = (int)(coff[b + 1] - c0), i = blk * SH_STL_BLK + threadIdx.x;
  if (blk * SH_STL_BLK >= n) return;
  const bool first = i < n && table[(size_t)b * tsize + slot_of[c0 + i]].y == i;
  int tot;
  const int r = stl_blk_rank(first, s_w, &tot);
  if (first) vid[c0 + i] = bcnt[(size_t)b * nblk + blk] + r;
}

// pass 4 (grid: triangle blocks x B): the other corners copy their vertex's id; tcnt[b][blk] = kept triangles of the block
__global__ void __launch_bounds__(SH_STL_BLK)
k_stl_tri_count(const long long* __restrict__ coff, const int2* __restrict__ table, int tsize, const int* __restrict__ slot_of, int* __restrict__ vid,
                int* __restrict__ tcnt, int nblk) {
  __shared__ int s_w[SH_STL_BLK / 64];
  const int b = blockIdx.y, blk = blockIdx.x;
  const long long c0 = coff[b];
  const int ntri = (int)((coff[b + 1] - c0) / 3), t = blk * SH_STL_BLK + threadIdx.x;
  if (blk * SH_STL_BLK >= ntri) return;
  bool keep = false;
  if (t < ntri) {
    int v[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int i = 3 * t + k, f = table[(size_t)b * tsize + slot_of[c0 + i]].y;
      v[k] = vid[c0 + f];                      // (f == i: its own id, written by pass 3; other blocks write only non-first corners)
      if (f != i) vid[c0 + i] = v[k];
    }
    keep = v[0] != v[1] && v[1] != v[2] && v[0] != v[2];
  }
  int tot;
  (void)stl_blk_rank(keep, s_w, &tot);
  if (threadIdx.x == 0) tcnt[(size_t)b * nblk + blk] = tot;
}

// pass 6 (grid: triangle blocks x B): fpos[t0 + t] = position of triangle t among the kept ones or -1
__global__ void __launch_bounds__(SH_STL_BLK)
k_stl_fpos(const long long* __restrict__ coff, const int* __restrict__ vid, const int* __restrict__ tcnt, int nblk, int* __restrict__ fpos) {
  __shared__ int s_w[SH_STL_BLK / 64];
  const int b = blockIdx.y, blk = blockIdx.x;
  const long long c0 = coff[b];
  const int ntri = (int)((coff[b + 1] - c0) / 3), t = blk * SH_STL_BLK + threadIdx.x;
  if (blk * SH_STL_BLK >= ntri) return;
  bool keep = false;
  if (t < ntri) {
    const int v0 = vid[c0 + 3 * t], v1 = vid[c0 + 3 * t + 1], v2 = vid[c0 + 3 * t + 2];
    keep = v0 != v1 && v1 != v2 && v0 != v2;
  }
  int tot;
  const int r = stl_blk_rank(keep, s_w, &tot);
  if (t < ntri) fpos[c0 / 3 + t] = keep ? tcnt[(size_t)b * nblk + blk] + r : -1;
}

// the six passes on stream `st`; bsum: B x 2 x nblk ints of scratch (nblk = corner blocks of the largest file)
inline void stl_rank_launch(hipStream_t st, int B, long long maxc, const long long* coff, const int2* table, int tsize, const int* slot_of, int* vid, int* fpos, int* counts,
                            int* bsum) {
  const int nblk = (int)((maxc + SH_STL_BLK - 1) / SH_STL_BLK), nblk_t = (int)((maxc / 3 + SH_STL_BLK - 1) / SH_STL_BLK);
  int* bcnt = bsum;
  int* tcnt = bsum + (size_t)B * nblk;
  hipLaunchKernelGGL(k_stl_first_count, dim3(nblk, B), dim3(SH_STL_BLK), 0, st, coff, table, tsize, slot_of, bcnt, nblk);
  hipLaunchKernelGGL(k_stl_scan_blocks, dim3(B), dim3(64), 0, st, coff, bcnt, nblk, 1, counts, 0);
  hipLaunchKernelGGL(k_stl_vid_first, dim3(nblk, B), dim3(SH_STL_BLK), 0, st, coff, table, tsize, slot_of, (const int*)bcnt, nblk, vid);
  hipLaunchKernelGGL(k_stl_tri_count, dim3(nblk_t, B), dim3(SH_STL_BLK), 0, st, coff, table, tsize, slot_of, vid, tcnt, nblk);
  hipLaunchKernelGGL(k_stl_scan_blocks, dim3(B), dim3(64), 0, st, coff, tcnt, nblk, 3, counts, 1);
  hipLaunchKernelGGL(k_stl_fpos, dim3(nblk_t, B), dim3(SH_STL_BLK), 0, st, coff, (const int*)vid, (const int*)tcnt, nblk, fpos);
}
inline size_t stl_rank_scratch_ints(int B, long long maxc) { return (size_t)B * 2 * (size_t)((maxc + SH_STL_BLK - 1) / SH_STL_BLK); }

__global__ void k_stl_emit(const float* __restrict__ corners, const long long* __restrict__ coff, const int2* __restrict__ table, int tsize,
                           const int* __restrict__ slot_of, const int* __restrict__ vid, const int* __restrict__ fpos,
                           const long long* __restrict__ voff, const long long* __restrict__ foff, float* __restrict__ verts, int* __restrict__ faces) {
  const int b = blockIdx.y;
  const long long c0 = coff[b], n = coff[b + 1] - c0;
  const int2* T = table + (size_t)b * tsize;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const int v = vid[c0 + i];
    if (T[slot_of[c0 + i]].y == (int)i) {
      float* o = verts + 3 * (voff[b] + v);
      const float* p = corners + 3 * (c0 + i);
      o[0] = p[0]; o[1] = p[1]; o[2] = p[2];
    }
    const long long t = i / 3;
    const int fp = fpos[c0 / 3 + t];
    if (fp >= 0) faces[3 * (foff[b] + fp) + (int)(i - 3 * t)] = v;
  }
}

// What sh_upload_meshes checks on the host, for a batch that is staged asynchronously (sh_stage_meshes): every face index inside
// its mesh, every coordinate finite.  flag |= 1 / 2.
__global__ void k_validate_meshes(const float* __restrict__ verts, const int* __restrict__ faces, const long long* __restrict__ voff,
                                  const long long* __restrict__ foff, int* __restrict__ flag) {
  const int b = blockIdx.y;
  const long long v0 = voff[b], nv = voff[b + 1] - v0, f0 = foff[b], nf = foff[b + 1] - f0;
  int bad = 0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < 3 * nf; i += (long long)gridDim.x * blockDim.x) {
    const int id = faces[3 * f0 + i];
    if (id < 0 || id >= nv) bad |= 1;
  }
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < 3 * nv; i += (long long)gridDim.x * blockDim.x) {
    const float x = verts[3 * v0 + i];
    if (!(fabsf(x) <= 3.402823466e+38f)) bad |= 2;
  }
  if (bad) atomicOr(flag, bad);
}

}  // namespace sh
