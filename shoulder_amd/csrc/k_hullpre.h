// k_hullpre.h -- device-side point prefilter for the host quickhull ("GPU extreme-point prefilter + host
// quickhull"; the hull itself is `Trimesh.apply_obb()` -> qhull in the reference, src/shoulder/humerus/mesh.py:82).
// A vertex strictly inside the polytope spanned by a few extreme VERTICES of the mesh lies strictly inside the convex
// hull, so it is no hull vertex and the hull of the remaining points is the same hull.  61 % of a humerus's vertices go
// this way: the read-back shrinks by that much and the host quickhull gets 2.5-3.5x faster.
//   k_hullpre_extremes   per humerus the vertex farthest along each of 26 directions ({-1,0,1}^3 \ 0; ties -> lowest index)
//   k_hullpre_polytope   supporting planes of those <= 26 points by brute force: a triple is a face iff every other
//                        extreme point lies on or behind its plane (<= 2 600 triples x 26 tests per humerus)
//   k_hullpre_filter     keep a vertex unless it is more than `margin` behind EVERY plane: a counting pass, the batch-wide
//                        offsets (k_hullpre_offsets), then a writing pass that compacts the survivors of ALL humeri into one
//                        array, each in file order (block scan) -> one read-back, a deterministic point list per hull
// A humerus whose polytope cannot be built (flat input, > SH_HP_MAXPL planes) keeps all its vertices.
#pragma once
#include "sh_common.h"
#include "k_stl.h"

namespace sh {

#define SH_HP_NDIR 26
#define SH_HP_MAXPL 128
// Safety of the test: a vertex is dropped only if it lies behind EVERY accepted plane, so accepting more planes can only keep
// more vertices; what must not happen is that a true face of the polytope is rejected (the test region would then reach
// outside the polytope).  A triple is therefore accepted as soon as no extreme point is more than TOL in front of it, with
// TOL far above the evaluation error of a badly conditioned (sliver) triple (~1e-7 at 300 mm), and the margin far above TOL.
#define SH_HP_TOL 1e-6        // |distance| below this = on the plane
#define SH_HP_MARGIN 1e-4     // a vertex closer than this to the polytope's surface is kept

__device__ inline void hp_dir(int k, int* d) {      // k in [0, 26): the 27 sign triples without (0,0,0)
  const int t = k >= 13 ? k + 1 : k;
  d[0] = t % 3 - 1; d[1] = (t / 3) % 3 - 1; d[2] = t / 9 - 1;
}

__global__ void __launch_bounds__(256)
k_hullpre_extremes(const float* __restrict__ verts, const long long* __restrict__ voff, int* __restrict__ ext /* B x 26 */) {
  __shared__ double s_v[4][SH_HP_NDIR];
  __shared__ int s_i[4][SH_HP_NDIR];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long v0 = voff[b];
  const int nv = (int)(voff[b + 1] - v0);
  const float* P = verts + 3 * v0;
  double best[SH_HP_NDIR];
  int bi[SH_HP_NDIR];
#pragma unroll
  for (int k = 0; k < SH_HP_NDIR; ++k) { best[k] = -1e300; bi[k] = 0x7fffffff; }
  for (int i = tid; i < nv; i += 256) {
    const double x = P[3 * i], y = P[3 * i + 1], z = P[3 * i + 2];
#pragma unroll
    for (int k = 0; k < SH_HP_NDIR; ++k) {
      int d[3];
      hp_dir(k, d);
      const double v = (d[0] * x + d[1] * y) + d[2] * z;
      if (v > best[k]) { best[k] = v; bi[k] = i; }      // ascending i per lane: the first maximum stays
    }
  }
#pragma unroll
  for (int k = 0; k < SH_HP_NDIR; ++k) {
    for (int off = 32; off > 0; off >>= 1) {
      const double ov = __shfl_down(best[k], off);
      const int oi = __shfl_down(bi[k], off);
      if (ov > best[k] || (ov == best[k] && oi < bi[k])) { best[k] = ov; bi[k] = oi; }
    }
    if (lane == 0) { s_v[wave][k] = best[k]; s_i[wave][k] = bi[k]; }
  }
  __syncthreads();
  if (tid < SH_HP_NDIR) {
    double v = s_v[0][tid];
    int i = s_i[0][tid];
    for (int w = 1; w < 4; ++w)
      if (s_v[w][tid] > v || (s_v[w][tid] == v && s_i[w][tid] < i)) { v = s_v[w][tid]; i = s_i[w][tid]; }
    ext[b * SH_HP_NDIR + tid] = nv > 0 ? i : -1;
  }
}

__global__ void __launch_bounds__(256)
k_hullpre_polytope(const float* __restrict__ verts, const long long* __restrict__ voff, const int* __restrict__ ext,
                   double* __restrict__ planes /* B x MAXPL x 4 */, int* __restrict__ nplanes /* B; -1 = no filter */) {
  __shared__ double E[SH_HP_NDIR][3];
  __shared__ int id[SH_HP_NDIR];
  __shared__ int cnt;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < SH_HP_NDIR) {
    const int i = ext[b * SH_HP_NDIR + tid];
    id[tid] = i;
    const float* p = verts + 3 * (voff[b] + (i < 0 ? 0 : i));
    E[tid][0] = p[0]; E[tid][1] = p[1]; E[tid][2] = p[2];
  }
  if (tid == 0) cnt = 0;
  __syncthreads();
  if (id[0] < 0) { if (tid == 0) nplanes[b] = -1; return; }
  constexpr int NT = SH_HP_NDIR * (SH_HP_NDIR - 1) * (SH_HP_NDIR - 2) / 6;      // 2600
  double* PL = planes + (size_t)b * SH_HP_MAXPL * 4;
  for (int t = tid; t < NT; t += 256) {
    // t -> (i < j < k)
    int i = 0, r = t;
    for (;; ++i) { const int c = (SH_HP_NDIR - 1 - i) * (SH_HP_NDIR - 2 - i) / 2; if (r < c) break; r -= c; }
    int j = i + 1;
    for (;; ++j) { const int c = SH_HP_NDIR - 1 - j; if (r < c) break; r -= c; }
    const int k = j + 1 + r;
    if (id[i] == id[j] || id[j] == id[k] || id[i] == id[k]) continue;
    const double ux = E[j][0] - E[i][0], uy = E[j][1] - E[i][1], uz = E[j][2] - E[i][2];
    const double vx = E[k][0] - E[i][0], vy = E[k][1] - E[i][1], vz = E[k][2] - E[i][2];
    double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
    const double nn = sqrt((nx * nx + ny * ny) + nz * nz);
    const double lu = sqrt((ux * ux + uy * uy) + uz * uz), lv = sqrt((vx * vx + vy * vy) + vz * vz);
    if (!(nn > 1e-9 * lu * lv)) continue;      // (nearly) collinear
    nx /= nn; ny /= nn; nz /= nn;
    const double d0 = (nx * E[i][0] + ny * E[i][1]) + nz * E[i][2];
    double smax = 0.0, smin = 0.0;
    for (int q = 0; q < SH_HP_NDIR; ++q) {
      const double s = ((nx * E[q][0] + ny * E[q][1]) + nz * E[q][2]) - d0;
      smax = fmax(smax, s); smin = fmin(smin, s);
    }
    double sg = 0.0;
    if (smax <= SH_HP_TOL && smin < -SH_HP_TOL) sg = 1.0;            // everything on or behind: outward normal n
    else if (smin >= -SH_HP_TOL && smax > SH_HP_TOL) sg = -1.0;      // everything on or in front: outward normal -n
    if (sg == 0.0) continue;
    const int slot = atomicAdd(&cnt, 1);
    if (slot < SH_HP_MAXPL) { PL[4 * slot] = sg * nx; PL[4 * slot + 1] = sg * ny; PL[4 * slot + 2] = sg * nz; PL[4 * slot + 3] = sg * d0; }
  }
  __syncthreads();
  if (tid == 0) nplanes[b] = (cnt >= 4 && cnt <= SH_HP_MAXPL) ? cnt : -1;
}

// keep test of one vertex against the planes staged in LDS
__device__ inline bool hp_keep(const float* P, int i, const double* PL, int np) {
  if (np < 0) return true;
  const double x = P[3 * i], y = P[3 * i + 1], z = P[3 * i + 2];
  for (int q = 0; q < np; ++q)
    if (((PL[4 * q] * x + PL[4 * q + 1] * y) + PL[4 * q + 2] * z) - PL[4 * q + 3] >= -SH_HP_MARGIN) return true;
  return false;
}

// WRITE = false: count the survivors of humerus b -> nkept[b].  WRITE = true: write them, in file order, at koff[b] of one
// array shared by the batch (k_hullpre_offsets in between), so that ONE copy brings every humerus's points to the host.
template <bool WRITE>
__global__ void __launch_bounds__(SH_STL_SCAN_THREADS)
k_hullpre_filter(const float* __restrict__ verts, const long long* __restrict__ voff, const double* __restrict__ planes,
                 const int* __restrict__ nplanes, const long long* __restrict__ koff, float* __restrict__ kept, int* __restrict__ nkept) {
  __shared__ double PL[SH_HP_MAXPL * 4];
  __shared__ int s_wave[SH_STL_SCAN_THREADS / 64];
  const int b = blockIdx.x, tid = threadIdx.x;
  const long long v0 = voff[b];
  const int nv = (int)(voff[b + 1] - v0);
  const int np = nplanes[b];
  for (int i = tid; i < 4 * max(np, 0); i += SH_STL_SCAN_THREADS) PL[i] = planes[(size_t)b * SH_HP_MAXPL * 4 + i];
  __syncthreads();
  const float* P = verts + 3 * v0;
  const int per = (nv + SH_STL_SCAN_THREADS - 1) / SH_STL_SCAN_THREADS;
  const int a = min(nv, tid * per), e = min(nv, a + per);
  int c = 0;
  for (int i = a; i < e; ++i) c += hp_keep(P, i, PL, np) ? 1 : 0;
  int total;
  int pos = stl_block_scan(c, s_wave, &total);
  if (!WRITE) { if (tid == 0) nkept[b] = total; return; }
  float* K = kept + 3 * koff[b];
  for (int i = a; i < e; ++i)
    if (hp_keep(P, i, PL, np)) { K[3 * pos] = P[3 * i]; K[3 * pos + 1] = P[3 * i + 1]; K[3 * pos + 2] = P[3 * i + 2]; ++pos; }
}

// koff[b] = survivors of the humeri before b; koff[B] = all of them (one workgroup; B is a batch size, not a mesh size)
__global__ void k_hullpre_offsets(const int* __restrict__ nkept, long long* __restrict__ koff, int B) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    long long acc = 0;
    for (int b = 0; b < B; ++b) { koff[b] = acc; acc += nkept[b]; }
    koff[B] = acc;
  }
}

}  // namespace sh
