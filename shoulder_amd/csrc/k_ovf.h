// k_ovf.h -- sections with more crossing segments than the fixed slot range holds (reference: `section_multiplane`,
// slice.py:26-28, has no limit; a 519 k-triangle humerus crosses a plane ~1 300 times).
//
// The fast path keeps its layout -- SH_MAXSEG = 1 024 segment slots and ring points per (humerus, plane), one pass, LDS joins in
// two tiers (k_slices.h).  A plane whose crossing count exceeds the slots (k_slice_emit counts every crossing, it just does
// not store past the range) becomes an OVERFLOW plane:
//   k_ovf_plan       per plane: count > SH_MAXSEG -> ranges of the context's pools for its segments, its closed ring and its
//                    join workspace (device-side bump allocation, no host round trip), and an entry in the set's list;
//   k_slice_emit_ovf the section again, for the listed planes only, into the segment pool (count -> allocate -> emit);
//   k_slice_link_huge    the join of k_slices.h with every array in global memory: same hash join, same pointer jumping for
//                    the canonical start and the ranks, same loop order, same lane-strided area sums -- one workgroup of 1 024
//                    lanes per listed plane; centroid / area / loop count / ring_n go where the fast path puts them, the ring
//                    into the ring pool;
//   k_resample_polar_huge, k_te_rows_huge, te_final's ring: the consumers of a ring, for listed planes, from the pool.
// All of them return at once when the set's list is empty (the normal case: ~2 us per launch).  Pools are sized by the host
// (`ovf.*` buffers); a run that needs more records how much (ctr[3..5]), flags the humerus SH_ERR_CAPACITY_DEV, and
// sh_collect grows the pools and runs the batch again.
#pragma once
#include "k_slices.h"

namespace sh {

struct OvfPools {
  Seg* segs; double* ring; unsigned char* work;
  unsigned long long seg_cap, ring_cap, work_cap;      // in segments / ring points / bytes
  unsigned long long* ctr;                             // [0] segments used (per set) [1] ring points used (per run) [2] work bytes used (per set)
                                                       // [3..5] high-water marks of what the run would have needed
};

// per-set plan arrays, one entry per (humerus, plane): ranges in the pools (-1: not an overflow plane)
struct OvfSet { long long* soff; long long* roff; long long* woff; int* fill; int* list; int* nlist /*[0] listed planes, [1] planes with many loops*/; int* list2; };

__host__ __device__ inline unsigned ovf_hash_size(int n) { unsigned h = 64; while (h < 2u * (unsigned)n) h <<= 1; return h; }
// join workspace of a plane with n segments (bytes, 16-aligned): skey, labA, labB (u64) | rx, ry (f64) | table (int, hash size) |
// nxt, jmpA, jmpB, offA, offB, pos (int); the resampler and the rectangle kernel reuse the same range
__host__ __device__ inline unsigned long long ovf_work_bytes(int n) {
  unsigned long long b = (unsigned long long)n * 8ull * 5ull + (unsigned long long)ovf_hash_size(n) * 4ull + (unsigned long long)n * 4ull * 6ull + 64ull;
  const unsigned long long te = (unsigned long long)(n + 1) * 8ull * 7ull + 64ull;      // k_te_rows_huge: xy (2), hx, hy, dq (2 int = 1), hull (2 int = 1)
  if (te > b) b = te;
  return (b + 15ull) & ~15ull;
}

__global__ void k_ovf_plan(int N, int nplanes, const int* __restrict__ seg_count, OvfPools P, OvfSet S, int* __restrict__ err) {
  const int pl = blockIdx.x * blockDim.x + threadIdx.x;
  if (pl >= nplanes) return;
  const int cnt = seg_count[pl];
  S.soff[pl] = -1; S.roff[pl] = -1; S.woff[pl] = -1; S.fill[pl] = 0;
  if (cnt <= SH_MAXSEG) return;
  const unsigned long long wb = ovf_work_bytes(cnt);
  const unsigned long long s = atomicAdd(&P.ctr[0], (unsigned long long)cnt);
  const unsigned long long r = atomicAdd(&P.ctr[1], (unsigned long long)cnt + 1ull);
  const unsigned long long w = atomicAdd(&P.ctr[2], wb);
  atomicMax(&P.ctr[3], s + cnt); atomicMax(&P.ctr[4], r + cnt + 1ull); atomicMax(&P.ctr[5], w + wb);
  if (s + cnt > P.seg_cap || r + cnt + 1ull > P.ring_cap || w + wb > P.work_cap) { atomicExch(&err[pl / N], SH_ERR_CAPACITY_DEV); return; }
  S.soff[pl] = (long long)s; S.roff[pl] = (long long)r; S.woff[pl] = (long long)w;
  S.list[atomicAdd(S.nlist, 1)] = pl;
}

// the section of k_slice_emit (same sign rule, same crossing-point arithmetic) for the listed planes only
__global__ void __launch_bounds__(256)
k_slice_emit_ovf(const double* __restrict__ vobb, const int* __restrict__ faces, const long long* __restrict__ voff, const long long* __restrict__ foff,
                 const double* __restrict__ zeff, int N, OvfPools P, OvfSet S) {
  if (*S.nlist == 0) return;
  const int b = blockIdx.y;
  const long long f0 = foff[b], nf = foff[b + 1] - f0;
  const double* vb = vobb + 3 * voff[b];
  const double* zp = zeff + (size_t)b * N;
  const double z_first = zp[0], z_last = zp[N - 1];
  const double inv_step = (double)(N - 1) / (z_last - z_first);
  for (long long fi = blockIdx.x * (long long)blockDim.x + threadIdx.x; fi < nf; fi += (long long)gridDim.x * blockDim.x) {
    const int* f = faces + 3 * (f0 + fi);
    const int id[3] = {f[0], f[1], f[2]};
    double Z[3];
    for (int k = 0; k < 3; ++k) Z[k] = vb[3 * (size_t)id[k] + 2];
    const double fzmin = fmin(Z[0], fmin(Z[1], Z[2])), fzmax = fmax(Z[0], fmax(Z[1], Z[2]));
    const double ka = (fzmin - z_first) * inv_step, kb = (fzmax - z_first) * inv_step;
    const double klo = fmin(ka, kb), khi = fmax(ka, kb);
    if (khi < -1.0 || klo > (double)N) continue;
    int lo = (int)floor(fmax(klo, 0.0)) - 1, hi = (int)ceil(fmin(khi, (double)(N - 1))) + 1;
    lo = lo < 0 ? 0 : lo;
    hi = hi > N - 1 ? N - 1 : hi;
    if (N == 1) { lo = 0; hi = 0; }
    for (int k = lo; k <= hi; ++k) {
      const long long so = S.soff[(size_t)b * N + k];
      if (so < 0) continue;
      const double z = zp[k];
      double d[3];
      int s[3];
      for (int j = 0; j < 3; ++j) { d[j] = Z[j] - z; s[j] = d[j] < -SH_SECTION_TOL ? -1 : 1; }
      if (s[0] == s[1] && s[1] == s[2]) continue;
      const int slot = atomicAdd(&S.fill[(size_t)b * N + k], 1);
      int up = 0, dn = 0;
      for (int j = 0; j < 3; ++j) {
        const int jn = (j + 1) % 3;
        if (s[j] == -1 && s[jn] == 1) up = j;
        if (s[j] == 1 && s[jn] == -1) dn = j;
      }
      Seg sg;
      {  // start = crossing on the edge walked downwards (+ -> -)
        const int a = dn, c = (dn + 1) % 3;
        const int l = id[a] < id[c] ? a : c, h = id[a] < id[c] ? c : a;
        sg.s_lo = (uint32_t)id[l]; sg.s_hi = (uint32_t)id[h];      // (its point: seg_start_point, in the join)
      }
      {
        const int a = up, c = (up + 1) % 3;
        const int l = id[a] < id[c] ? a : c, h = id[a] < id[c] ? c : a;
        sg.e_lo = (uint32_t)id[l]; sg.e_hi = (uint32_t)id[h];
      }
      P.segs[so + slot] = sg;
    }
  }
}

// Planes the LDS joins handed over because they have more than SH_MAXLOOPS loops (k_slices.h ManyLoops): pool ranges like
// k_ovf_plan's, their segments copied from the fixed slots into the segment pool, an entry in the set's list -- k_slice_link_huge
// joins them next, their consumers read the ring from the pool.
__global__ void __launch_bounds__(256)
k_ovf_plan_loops(int N, const int* __restrict__ seg_count, const Seg* __restrict__ segs /*fixed slots*/, OvfPools P, OvfSet S, int* __restrict__ err) {
  __shared__ long long so, ro, wo;
  const int n2 = S.nlist[1];
  for (int i = blockIdx.x; i < n2; i += gridDim.x) {
    const int pl = S.list2[i];
    const int cnt = seg_count[pl];
    if (threadIdx.x == 0) {
      const unsigned long long wb = ovf_work_bytes(cnt);
      const unsigned long long s = atomicAdd(&P.ctr[0], (unsigned long long)cnt);
      const unsigned long long r = atomicAdd(&P.ctr[1], (unsigned long long)cnt + 1ull);
      const unsigned long long w = atomicAdd(&P.ctr[2], wb);
      atomicMax(&P.ctr[3], s + cnt); atomicMax(&P.ctr[4], r + cnt + 1ull); atomicMax(&P.ctr[5], w + wb);
      if (s + cnt > P.seg_cap || r + cnt + 1ull > P.ring_cap || w + wb > P.work_cap) { atomicExch(&err[pl / N], SH_ERR_CAPACITY_DEV); so = -1; }
      else { so = (long long)s; ro = (long long)r; wo = (long long)w; }
    }
    __syncthreads();
    if (so >= 0) {
      const Seg* src = segs + (size_t)pl * SH_MAXSEG;
      for (int j = threadIdx.x; j < cnt; j += 256) P.segs[so + j] = src[j];
      if (threadIdx.x == 0) { S.soff[pl] = so; S.roff[pl] = ro; S.woff[pl] = wo; S.fill[pl] = cnt; S.list[atomicAdd(&S.nlist[0], 1)] = pl; }
    }
    __syncthreads();
  }
}

#define SH_HUGE_THREADS 1024
#define SH_MAXLOOPS_G 1024      // loops per plane in the overflow tier's join (the LDS tiers: SH_MAXLOOPS = 32, more go here)

// slice_link_plane (k_slices.h) with its arrays in the plane's workspace; every step in the same order with the same arithmetic
__device__ inline void slice_link_plane_g(const int pl, int N, const int* __restrict__ seg_count, const Seg* __restrict__ sp, const double* __restrict__ vb /*verts_obb of the plane's mesh*/,
             const double zpl /*the plane's height*/, unsigned char* __restrict__ wk,
             double* __restrict__ centroids, double* __restrict__ areas, int* __restrict__ nloops, int* __restrict__ ring_n, double* __restrict__ ring_out /*nullable*/,
             int select, int* __restrict__ err, double* __restrict__ areas_total) {
  constexpr int T = SH_HUGE_THREADS;
  __shared__ unsigned long long l_key[SH_MAXLOOPS_G];
  __shared__ int l_start[SH_MAXLOOPS_G], l_len[SH_MAXLOOPS_G], l_off[SH_MAXLOOPS_G];
  __shared__ double l_area[SH_MAXLOOPS_G], l_sel[SH_MAXLOOPS_G];
  __shared__ int n_loops, bad;
  __shared__ double bbw[T / 64][4];
  const int b = pl / N, tid = threadIdx.x;
  const int n = seg_count[pl];
  const unsigned HASH = ovf_hash_size(n);
  unsigned long long* skey = (unsigned long long*)wk;
  unsigned long long* bufA = skey + n;
  unsigned long long* bufB = bufA + n;
  double* rx = (double*)(bufB + n);
  double* ry = rx + n;
  int* table = (int*)(ry + n);
  int* nxt = table + HASH;
  int* jmpA = nxt + n; int* jmpB = jmpA + n; int* offA = jmpB + n; int* offB = offA + n; int* posv = offB + n;
  if (tid == 0) { n_loops = 0; bad = 0; }
  for (unsigned i = tid; i < HASH; i += T) table[i] = -1;
  for (int i = tid; i < n; i += T) {
    const Seg s = sp[i];
    skey[i] = ((unsigned long long)s.s_lo << 32) | s.s_hi;
    bufA[i] = ((unsigned long long)s.e_lo << 32) | s.e_hi;
  }
  __syncthreads();
  for (int i = tid; i < n; i += T) {
    uint32_t h = hash_key64(skey[i]) & (HASH - 1);
    while (atomicCAS(&table[h], -1, i) != -1) h = (h + 1) & (HASH - 1);
  }
  __syncthreads();
  for (int i = tid; i < n; i += T) {
    const unsigned long long k = bufA[i];
    uint32_t h = hash_key64(k) & (HASH - 1);
    int t, found = -1;
    while ((t = table[h]) != -1) {
      if (skey[t] == k) { found = t; break; }
      h = (h + 1) & (HASH - 1);
    }
    if (found < 0) { found = i; bad = 1; }   // open contour: self-loop keeps the walk bounded
    nxt[i] = found;
  }
  __syncthreads();
  unsigned long long* labA = bufA;
  unsigned long long* labB = bufB;
  int* ja = jmpA; int* jb = jmpB;
  int* ra = offA; int* rb = offB;
  for (int i = tid; i < n; i += T) { labA[i] = skey[i]; ja[i] = nxt[i]; ra[i] = 0; }
  __syncthreads();
  for (int span = 1; span < n; span <<= 1) {
    for (int i = tid; i < n; i += T) {
      const int j = ja[i];
      const unsigned long long a = labA[i], c = labA[j];
      const bool own = a <= c;
      labB[i] = own ? a : c;
      rb[i] = own ? ra[i] : span + ra[j];
      jb[i] = ja[j];
    }
    __syncthreads();
    unsigned long long* tl = labA; labA = labB; labB = tl;
    int* tj = ja; ja = jb; jb = tj;
    int* tr = ra; ra = rb; rb = tr;
  }
  for (int i = tid; i < n; i += T)
    if (ra[i] == 0) {
      const int l = atomicAdd(&n_loops, 1);
      if (l < SH_MAXLOOPS_G) l_start[l] = i;
    }
  __syncthreads();
  const int nl = n_loops > SH_MAXLOOPS_G ? SH_MAXLOOPS_G : n_loops;
  if (tid == 0) {
    if (n_loops > SH_MAXLOOPS_G) atomicExch(&err[b], SH_ERR_CAPACITY_DEV);      // (more than 1 024 closed loops in one section)
    for (int a = 1; a < nl; ++a) {      // canonical loop order: ascending start key
      const int v = l_start[a]; int c = a - 1;
      while (c >= 0 && skey[l_start[c]] > skey[v]) { l_start[c + 1] = l_start[c]; --c; }
      l_start[c + 1] = v;
    }
    int off = 0;
    for (int l = 0; l < nl; ++l) {
      const int s = l_start[l];
      const int L = ra[nxt[s]] + 1;
      l_len[l] = L; l_off[l] = off; off += L;
      l_key[l] = skey[s];
    }
    if (off != n) bad = 1;        // some segments are on no closed loop
  }
  __syncthreads();
  for (int i = tid; i < n; i += T) {      // ring placement: position from start = (L - r) mod L
    const unsigned long long key = labA[i];
    int l = -1;
    for (int q = 0; q < nl; ++q) if (l_key[q] == key) { l = q; break; }
    const int L = l >= 0 ? l_len[l] : 1;
    const int r = ra[i];
    const int pos = r == 0 ? 0 : L - r;
    posv[i] = l < 0 ? -1 : l_off[l] + pos;
  }
  __syncthreads();
  // (rx / ry do not alias the label buffers here, so no barrier is needed between reading labA and writing them)
  for (int i = tid; i < n; i += T)
    if (posv[i] >= 0 && posv[i] < n) { const Seg sg = sp[i]; seg_start_point(vb, sg.s_lo, sg.s_hi, zpl, &rx[posv[i]], &ry[posv[i]]); }
  __syncthreads();
  // AABB over every loop vertex by all waves (min / max: order free), the per-loop sums by one wave per loop as in the LDS tiers
  const int lane = tid & 63, wave = tid >> 6;
  {
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    for (int i = tid; i < n; i += T) {
      const double qx = rx[i], qy = ry[i];
      x0 = fmin(x0, qx); x1 = fmax(x1, qx); y0 = fmin(y0, qy); y1 = fmax(y1, qy);
    }
    for (int off = 32; off > 0; off >>= 1) {
      x0 = fmin(x0, __shfl_down(x0, off)); x1 = fmax(x1, __shfl_down(x1, off));
      y0 = fmin(y0, __shfl_down(y0, off)); y1 = fmax(y1, __shfl_down(y1, off));
    }
    if (lane == 0) { bbw[wave][0] = x0; bbw[wave][1] = x1; bbw[wave][2] = y0; bbw[wave][3] = y1; }
  }
  for (int l = wave; l < nl; l += T / 64) {
    const int o = l_off[l], L = l_len[l];
    double a2 = 0.0, mx = 0.0, my = 0.0;
    for (int q = lane; q < L; q += 64) {
      const int qn = q + 1 == L ? 0 : q + 1;
      a2 += rx[o + q] * ry[o + qn] - rx[o + qn] * ry[o + q];
      mx += rx[o + q]; my += ry[o + q];
    }
    for (int off = 32; off > 0; off >>= 1) { a2 += __shfl_down(a2, off); mx += __shfl_down(mx, off); my += __shfl_down(my, off); }
    if (lane == 0) {
      l_area[l] = 0.5 * a2;
      mx = (mx + rx[o]) / (double)(L + 1); my = (my + ry[o]) / (double)(L + 1);      // surgical_neck.py:43-46: mean over the CLOSED ring
      l_sel[l] = fabs(mx) + fabs(my);
    }
  }
  __syncthreads();
  int best = 0;
  for (int l = 1; l < nl; ++l) {
    if (select == 0) { if (fabs(l_area[l]) > fabs(l_area[best])) best = l; }
    else { if (l_sel[l] < l_sel[best]) best = l; }
  }
  if (tid == 0) {
    double x0 = bbw[0][0], x1 = bbw[0][1], y0 = bbw[0][2], y1 = bbw[0][3];
    for (int w = 1; w < T / 64; ++w) { x0 = fmin(x0, bbw[w][0]); x1 = fmax(x1, bbw[w][1]); y0 = fmin(y0, bbw[w][2]); y1 = fmax(y1, bbw[w][3]); }
    centroids[2 * (size_t)pl] = 0.5 * (x0 + x1);
    centroids[2 * (size_t)pl + 1] = 0.5 * (y0 + y1);
    int amax = 0;
    for (int l = 1; l < nl; ++l) if (fabs(l_area[l]) > fabs(l_area[amax])) amax = l;
    areas[pl] = nl > 0 ? fabs(l_area[amax]) : 0.0;
    if (areas_total) {
      double tot = 0.0;
      for (int l = 0; l < nl; ++l) tot += l_area[l];
      areas_total[pl] = fabs(tot);
    }
    nloops[pl] = nl;
    ring_n[pl] = nl > 0 ? l_len[best] : 0;
    if (bad || nl == 0) atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV);
  }
  if (ring_out && nl > 0) {
    const int l = best, o = l_off[l], L = l_len[l];
    const bool rev = l_area[l] < 0;           // clockwise loop: traverse backwards from the same start
    for (int q = tid; q <= L; q += T) {
      const int qq = q == L ? 0 : q;
      const int src = rev ? (qq == 0 ? 0 : L - qq) : qq;
      ring_out[2 * q] = rx[o + src];
      ring_out[2 * q + 1] = ry[o + src];
    }
  }
  __syncthreads();
}

__global__ void __launch_bounds__(SH_HUGE_THREADS)
k_slice_link_huge(int N, const int* __restrict__ seg_count, OvfPools P, OvfSet S, double* __restrict__ centroids, double* __restrict__ areas,
                  int* __restrict__ nloops, int* __restrict__ ring_n, int want_ring, int select, int* __restrict__ err, double* __restrict__ areas_total,
                  const double* __restrict__ vobb, const long long* __restrict__ voff, const double* __restrict__ zeff) {
  const int nl = *S.nlist;
  for (int i = blockIdx.x; i < nl; i += gridDim.x) {
    const int pl = S.list[i];
    slice_link_plane_g(pl, N, seg_count, P.segs + S.soff[pl], vobb + 3 * voff[pl / N], zeff[pl], P.work + S.woff[pl], centroids, areas, nloops, ring_n,
                       want_ring ? P.ring + 2 * S.roff[pl] : (double*)nullptr, select, err, areas_total);
  }
}

// resample_polar_plane (k_slices.h) for a listed plane: ring from the pool, cumulative lengths in the plane's workspace
__global__ void __launch_bounds__(SH_RS_THREADS)
k_resample_polar_huge(int N, int M, const int* __restrict__ ring_n, OvfPools P, OvfSet S, const double* __restrict__ centroids,
                      double* __restrict__ ixy, double* __restrict__ itr_start, double* __restrict__ itr_cs, const RsWant W) {
  constexpr int NS = SH_MPROX / SH_RS_THREADS;
  __shared__ int amin_idx;
  __shared__ double wmin[SH_RS_THREADS / 64];
  __shared__ int widx[SH_RS_THREADS / 64];
  const int nlist = *S.nlist, tid = threadIdx.x;
  for (int it = blockIdx.x; it < nlist; it += gridDim.x) {
    const int pl = S.list[it];
    const int kpl = pl % N;
    const bool want_st = W.keep_all || kpl >= W.st_lo, want_cs = W.keep_all || (kpl >= W.cs_lo && kpl < W.cs_hi);      // (k_slices.h, RsWant)
    if (!want_st && !want_cs) continue;
    const int L = ring_n[pl];
    const double* rp = P.ring + 2 * S.roff[pl];
    double* d = (double*)(P.work + S.woff[pl]);
    for (int q = 1 + tid; q <= L; q += SH_RS_THREADS) {
      const double dx = rp[2 * q] - rp[2 * (q - 1)], dy = rp[2 * q + 1] - rp[2 * (q - 1) + 1];
      d[q] = sqrt(dx * dx + dy * dy);
    }
    __syncthreads();
    if (tid == 0) {      // np.cumsum's running sum
      double acc = 0.0;
      d[0] = 0.0;
      for (int q = 1; q <= L; ++q) { acc += d[q]; d[q] = acc; }
    }
    __syncthreads();
    const double dmax = d[L];
    double sx[NS], sy[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int j = tid + u * SH_RS_THREADS;
      const double t = linspace_at(0.0, dmax, M, j);
      const int n = L + 1;
      if (t < d[0]) { sx[u] = rp[0]; sy[u] = rp[1]; }
      else if (!(t < d[n - 1])) { sx[u] = rp[2 * (n - 1)]; sy[u] = rp[2 * (n - 1) + 1]; }
      else {
        int lo = 0, hi = n - 1;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (t >= d[mid]) lo = mid; else hi = mid; }
        const double x0 = d[lo], fx = rp[2 * lo], fy = rp[2 * lo + 1];
        if (x0 == t) { sx[u] = fx; sy[u] = fy; }
        else {
          const double den = d[lo + 1] - x0;
          sx[u] = (rp[2 * (lo + 1)] - fx) / den * (t - x0) + fx;
          sy[u] = (rp[2 * (lo + 1) + 1] - fy) / den * (t - x0) + fy;
        }
      }
    }
    if (W.keep_all) {
      double* oxy = ixy + (size_t)pl * 2 * M;
#pragma unroll
      for (int u = 0; u < NS; ++u) { const int j = tid + u * SH_RS_THREADS; oxy[j] = sx[u]; oxy[M + j] = sy[u]; }
    }
    const double cx = centroids[2 * (size_t)pl], cy = centroids[2 * (size_t)pl + 1];
    for (int pass = 0; pass < 2; ++pass) {
      if (pass == 0 ? !want_st : !want_cs) continue;
      const double ox = pass ? cx : 0.0, oy = pass ? cy : 0.0;
      double best = 1e300;
      int bi = 0x7fffffff;
      double th[NS], rr[NS];
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        const int j = tid + u * SH_RS_THREADS;
        const double x = sx[u] - ox, y = sy[u] - oy;
        th[u] = atan2(y, x);
        rr[u] = sqrt(x * x + y * y);
        if (th[u] < best || (th[u] == best && j < bi)) { best = th[u]; bi = j; }
      }
      for (int off = 32; off > 0; off >>= 1) {
        const double ob = __shfl_down(best, off);
        const int oi = __shfl_down(bi, off);
        if (ob < best || (ob == best && oi < bi)) { best = ob; bi = oi; }
      }
      if ((tid & 63) == 0) { wmin[tid >> 6] = best; widx[tid >> 6] = bi; }
      __syncthreads();
      if (tid == 0) {
        double bv = wmin[0]; int bx = widx[0];
        for (int w = 1; w < SH_RS_THREADS / 64; ++w)
          if (wmin[w] < bv || (wmin[w] == bv && widx[w] < bx)) { bv = wmin[w]; bx = widx[w]; }
        amin_idx = bx;
      }
      __syncthreads();
      const int k0 = amin_idx;
      double* o = (pass ? itr_cs : itr_start) + (size_t)pl * 2 * M;
#pragma unroll
      for (int u = 0; u < NS; ++u) {
        int dst = tid + u * SH_RS_THREADS - k0; if (dst < 0) dst += M;
        o[dst] = th[u];
        o[M + dst] = rr[u];
      }
      __syncthreads();
    }
  }
}

}  // namespace sh
