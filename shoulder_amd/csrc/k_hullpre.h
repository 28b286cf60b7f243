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

// SH_HP_PARTS workgroups per humerus (a contiguous range of its vertices each) in the passes over the vertices: with one workgroup
// per humerus a batch of 64 used 64 of the 256 CUs and the three passes took 0.29 + 0.28 + 0.19 ms (round 3: 0.83 ms of the CUs
// the UNet leaves free, per step).  Partial extremes are merged by k_hullpre_polytope (ranges ascend, `>` keeps the first maximum:
// the same vertex as one pass in index order); survivors are counted and written per range, every range at its own offset of the
// batch's array (k_hullpre_offsets), in file order inside it -- the same array as before, byte for byte.
#define SH_HP_PARTS 8

__global__ void __launch_bounds__(256)
k_hullpre_extremes(const float* __restrict__ verts, const long long* __restrict__ voff, double* __restrict__ pval /* B x PARTS x 26 */,
                   int* __restrict__ pidx /* B x PARTS x 26; 0x7fffffff = empty range */) {
  __shared__ double s_v[4][SH_HP_NDIR];
  __shared__ int s_i[4][SH_HP_NDIR];
  const int b = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long v0 = voff[b];
  const int nv = (int)(voff[b + 1] - v0);
  const int per = (nv + SH_HP_PARTS - 1) / SH_HP_PARTS;
  const int a = min(nv, part * per), e = min(nv, a + per);
  const float* P = verts + 3 * v0;
  double best[SH_HP_NDIR];
  int bi[SH_HP_NDIR];
#pragma unroll
  for (int k = 0; k < SH_HP_NDIR; ++k) { best[k] = -1e300; bi[k] = 0x7fffffff; }
  for (int i = a + tid; i < e; i += 256) {
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
    pval[((size_t)b * SH_HP_PARTS + part) * SH_HP_NDIR + tid] = v;
    pidx[((size_t)b * SH_HP_PARTS + part) * SH_HP_NDIR + tid] = i;
  }
}

__global__ void __launch_bounds__(256)
k_hullpre_polytope(const float* __restrict__ verts, const long long* __restrict__ voff, const double* __restrict__ pval, const int* __restrict__ pidx,
                   int* __restrict__ ext /* B x 26: the extreme vertices (out) */, double* __restrict__ planes /* B x MAXPL x 4 */, int* __restrict__ nplanes /* B; -1 = no filter */) {
  __shared__ double E[SH_HP_NDIR][3];
  __shared__ int id[SH_HP_NDIR];
  __shared__ int cnt;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < SH_HP_NDIR) {
    double v = -1e300;
    int i = 0x7fffffff;
    for (int part = 0; part < SH_HP_PARTS; ++part) {      // ranges in ascending vertex order
      const double pv = pval[((size_t)b * SH_HP_PARTS + part) * SH_HP_NDIR + tid];
      const int pi = pidx[((size_t)b * SH_HP_PARTS + part) * SH_HP_NDIR + tid];
      if (pi != 0x7fffffff && (pv > v || (pv == v && pi < i))) { v = pv; i = pi; }
    }
    if (i == 0x7fffffff) i = -1;
    ext[b * SH_HP_NDIR + tid] = i;
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

// WRITE = false: count the survivors of range `part` of humerus b -> pcnt[b][part].  WRITE = true: write them, in file order, at
// poff[b][part] of one array shared by the batch (k_hullpre_offsets in between), so that ONE copy brings every humerus's points
// to the host.  A wave takes 64 consecutive vertices at a time (coalesced reads; ballot + popcount give the positions).
template <bool WRITE>
__global__ void __launch_bounds__(256)
k_hullpre_filter(const float* __restrict__ verts, const long long* __restrict__ voff, const double* __restrict__ planes,
                 const int* __restrict__ nplanes, const long long* __restrict__ poff, float* __restrict__ kept, int* __restrict__ pcnt) {
  __shared__ double PL[SH_HP_MAXPL * 4];
  __shared__ int s_wave[2][4];
  const int b = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long v0 = voff[b];
  const int nv = (int)(voff[b + 1] - v0);
  const int np = nplanes[b];
  for (int i = tid; i < 4 * max(np, 0); i += 256) PL[i] = planes[(size_t)b * SH_HP_MAXPL * 4 + i];
  __syncthreads();
  const float* P = verts + 3 * v0;
  const int per = (nv + SH_HP_PARTS - 1) / SH_HP_PARTS;
  const int a = min(nv, part * per), e = min(nv, a + per);
  float* K = WRITE ? kept + 3 * poff[(size_t)b * SH_HP_PARTS + part] : (float*)nullptr;
  int run = 0, it = 0;
  for (int base = a; base < e; base += 256, it ^= 1) {
    const int i = base + tid;
    const bool keep = i < e && hp_keep(P, i, PL, np);
    const unsigned long long bal = __ballot(keep);
    if (lane == 0) s_wave[it][wave] = __popcll(bal);
    __syncthreads();      // (s_wave alternates between two slots: one barrier per round is enough)
    int woff = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { const int cw = s_wave[it][w]; if (w < wave) woff += cw; total += cw; }
    if (WRITE && keep) {
      const int pos = run + woff + __popcll(bal & (lane == 0 ? 0ull : (~0ull >> (64 - lane))));
      K[3 * pos] = P[3 * i]; K[3 * pos + 1] = P[3 * i + 1]; K[3 * pos + 2] = P[3 * i + 2];
    }
    run += total;
  }
  if (!WRITE && tid == 0) pcnt[(size_t)b * SH_HP_PARTS + part] = run;
}

// koff[b] = survivors of the humeri before b (koff[B] = all of them), poff[b][part] = those before range `part` of humerus b,
// nkept[b] = survivors of humerus b.  One wave: a lane per humerus, 64 humeri per round, a wave prefix sum between them.
__global__ void __launch_bounds__(64)
k_hullpre_offsets(const int* __restrict__ pcnt, long long* __restrict__ koff, long long* __restrict__ poff, int* __restrict__ nkept, int B) {
  const int lane = threadIdx.x;
  long long carry = 0;
  for (int base = 0; base < B; base += 64) {
    const int b = base + lane;
    int c[SH_HP_PARTS];
    long long mine = 0;
#pragma unroll
    for (int p = 0; p < SH_HP_PARTS; ++p) { c[p] = b < B ? pcnt[(size_t)b * SH_HP_PARTS + p] : 0; mine += c[p]; }
    long long incl = mine;
    for (int off = 1; off < 64; off <<= 1) { const long long t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    long long acc = carry + incl - mine;
    if (b < B) {
      koff[b] = acc; nkept[b] = (int)mine;
#pragma unroll
      for (int p = 0; p < SH_HP_PARTS; ++p) { poff[(size_t)b * SH_HP_PARTS + p] = acc; acc += c[p]; }
    }
    carry += __shfl(incl, 63);
  }
  if (lane == 0) koff[B] = carry;
}

}  // namespace sh
