// k_stl.h -- binary STL ingest on the device (`trimesh.load_mesh(stl, process=True)` at reference
// src/shoulder/humerus/mesh.py:22-27 with the canonical rule of oracle/stl.py and shoulder_amd/stl.py: vertices with
// equal bit patterns (after -0.0 -> +0.0) merge, merged vertices are numbered by first appearance in the file,
// triangles that use a vertex twice are dropped, kept triangles stay in file order).
//   k_stl_corners   one lane per triangle corner: 50-byte records (2-byte aligned) -> float32 xyz
//   k_stl_hash      open-addressing table per mesh keyed by the 96-bit pattern; entry = (owner corner, first corner)
//   k_stl_rank      one workgroup per mesh: flag "this corner is its vertex's first appearance", block scan -> vertex ids;
//                   per-triangle keep flags, block scan -> positions; counts V, F
//   k_stl_emit      verts / faces at their batch offsets
// Integer work end to end: the result is bit-identical to the host routine (tests/test_gpu_stl.py).
#pragma once
#include "sh_common.h"

namespace sh {

#define SH_STL_SCAN_THREADS 1024

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

// exclusive scan of one value per thread over the workgroup (SH_STL_SCAN_THREADS lanes); returns the total through *total
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

// vid[c0 + i] = vertex id of corner i; fpos[t0 + t] = position of triangle t among the kept ones or -1; counts[b] = (V, F)
__global__ void __launch_bounds__(SH_STL_SCAN_THREADS)
k_stl_rank(const long long* __restrict__ coff, const int2* __restrict__ table, int tsize, const int* __restrict__ slot_of,
           int* __restrict__ vid, int* __restrict__ fpos, int* __restrict__ counts) {
  __shared__ int s_wave[SH_STL_SCAN_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const long long c0 = coff[b];
  const int n = (int)(coff[b + 1] - c0), ntri = n / 3;
  const int2* T = table + (size_t)b * tsize;
  const int* S = slot_of + c0;
  // vertex ids: rank of the first-appearance corner among all first-appearance corners
  const int per = (n + SH_STL_SCAN_THREADS - 1) / SH_STL_SCAN_THREADS;
  const int a = min(n, tid * per), e = min(n, a + per);
  int cnt = 0;
  for (int i = a; i < e; ++i) cnt += T[S[i]].y == i ? 1 : 0;
  int V;
  int pos = stl_block_scan(cnt, s_wave, &V);
  for (int i = a; i < e; ++i) if (T[S[i]].y == i) vid[c0 + i] = pos++;      // first appearances carry the id ...
  __syncthreads();
  __threadfence_block();
  for (int i = a; i < e; ++i) { const int f = T[S[i]].y; if (f != i) vid[c0 + i] = vid[c0 + f]; }      // ... the others copy it
  __syncthreads();
  // kept triangles, in file order
  const int pert = (ntri + SH_STL_SCAN_THREADS - 1) / SH_STL_SCAN_THREADS;
  const int ta = min(ntri, tid * pert), te = min(ntri, ta + pert);
  int kc = 0;
  for (int t = ta; t < te; ++t) {
    const int v0 = vid[c0 + 3 * t], v1 = vid[c0 + 3 * t + 1], v2 = vid[c0 + 3 * t + 2];
    kc += (v0 != v1 && v1 != v2 && v0 != v2) ? 1 : 0;
  }
  int F;
  int fp = stl_block_scan(kc, s_wave, &F);
  int* FP = fpos + c0 / 3;
  for (int t = ta; t < te; ++t) {
    const int v0 = vid[c0 + 3 * t], v1 = vid[c0 + 3 * t + 1], v2 = vid[c0 + 3 * t + 2];
    FP[t] = (v0 != v1 && v1 != v2 && v0 != v2) ? fp++ : -1;
  }
  if (tid == 0) { counts[2 * b] = V; counts[2 * b + 1] = F; }
}

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
