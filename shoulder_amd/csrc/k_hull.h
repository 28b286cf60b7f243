// k_hull.h -- 3-D convex hull on the device: the hull stage of SH_STAGE_OBB without the host
// (`Trimesh.apply_obb()` -> qhull in the reference, src/shoulder/humerus/mesh.py:82; host form: sh_hull.h).
//
// Round-based quickhull, one workgroup (16 waves) per humerus, input = the survivors of the device prefilter (k_hullpre.h).
// No adjacency structure.  Per round:
//   R0  every alive face that holds outside points offers its farthest one (the face's "apex": a 64-bit key kept by
//       atomicMax when points are assigned); about 64 of them, drawn by a hash of (point, round), are the round's candidates
//   R1  a candidate's visible faces = plane tests against ALL alive faces (planes live in LDS); every undirected edge of a
//       visible face is claimed in a hash table with the candidate's priority, the smallest claim wins
//   R2  a candidate goes ahead when every edge of its visible faces is its own: the visible regions of the candidates that go
//       ahead share no face and are not edge-adjacent -- then their cones are independent (a cone's faces are reachable only
//       through its ring, which the other apex does not see), and inserting them together equals inserting them one by one
//   R3  horizon = directed edges of the visible faces whose reverse is no edge of a visible face; it must be ONE simple loop
//       (with a tolerance it can pinch: the humerus is then reported for the host quickhull, which has the retry logic)
//   R4  slots for the new faces: free slots of earlier rounds first (deterministic prefix sums, no atomics on the order)
//   R5  the visible faces die, a new face (a, b, apex) per horizon edge
//   R6  the outside points of the dead faces move to the first new face (creation order) of their killer that sees them
// A humerus takes 85-110 rounds (about 1 500 insertions, 10-25 per round once ~100 faces hold points; the candidate draw must
// be re-made every round -- a face's apex does not change until the face dies -- and incoherent in space, or chains of
// neighbours all lose to one another: with the lowest slot or the lowest direction bucket winning it took 1 000 rounds).
// tests/hostcheck/hull_rounds_ref.h is the sequential restatement of exactly these rules; on the CPU it returns the host
// quickhull's triangles on all fixtures, and tests/test_gpu_hull.py pins this kernel to the same hulls.
// Emitted record = the host's (k_obb.h): hull vertices (original float32 coordinates as doubles, numbered by point index),
// unit normals (from the triangle rotated to its smallest vertex first, coordinates centred on the bounding-box midpoint: the
// host quickhull writes its normals the same way, so both paths hand k_obb_candidates the same bits), edges (va, vb, f, g).
#pragma once
#include "k_hullpre.h"
#include "k_obb.h"      // HullCap: the per-humerus strides of the hull record

namespace sh {

#define HD_THREADS 1024
#define HD_NW (HD_THREADS / 64)
#define HD_SLOTS 3072         // face slots (alive + not yet reused); a humerus ends with ~2 700 faces
#define HD_NMAX 8192          // input points (survivors of the prefilter: ~6 300 of a humerus's 16 222 vertices)
#define HD_KC 64              // candidates per round (4 per wave)
#define HD_NJ (HD_KC / HD_NW)
#define HD_VL 32              // visible faces of a candidate kept in LDS (more: global scratch)
#define HD_NSL 1024           // new-face slots of a round kept in LDS (more: global scratch)
#define HD_VMAX 256           // visible faces per candidate
#define HD_TBL 32768          // entries of the edge table
#define HD_MAXROUNDS 4096
#define HD_EPS_REL 1e-10      // as sh_hull.h: points closer than 1e-10 * bbox diagonal to the hull count as inside

// global scratch of one humerus (sh_ctx allocates B of each; all L2 resident)
struct HullScratch {
  int* fv;                      // [HD_SLOTS][3] face vertices (point indices)
  int* vis;                     // [HD_KC][HD_VMAX] visible faces of the round's candidates
  int* ev;                      // [HD_KC][3 * HD_VMAX][2] their directed edges
  int* hor;                     // [HD_KC][HD_VMAX + 2][2] horizon edges (a -> b)
  int* newslot;                 // [HD_SLOTS] slots of the round's new faces
  int* freestack;               // [HD_SLOTS] dead slots of earlier rounds
  unsigned long long* tkeys;    // [HD_TBL] edge table: key = stamp << 26 | min << 13 | max  (0 = empty)
  unsigned* tvals;              // [HD_TBL] smallest claiming priority / owning face slot
};

__device__ inline unsigned hd_hash32(unsigned x) { x *= 2654435761u; x ^= x >> 15; x *= 2246822519u; x ^= x >> 13; return x; }
__device__ inline unsigned hd_hash_key(unsigned long long k) { return hd_hash32((unsigned)k ^ (unsigned)(k >> 29) * 0x9E3779B1u); }

// exclusive prefix of v over the workgroup in thread order; *total = sum.  s_w: HD_NW + 1 ints.  (two barriers)
__device__ inline int hd_block_scan(int v, int* s_w, int* total) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  int inc = v;
  for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(inc, off); if (lane >= off) inc += t; }
  if (lane == 63) s_w[wave] = inc;
  __syncthreads();
  if (threadIdx.x == 0) { int acc = 0; for (int w = 0; w < HD_NW; ++w) { const int t = s_w[w]; s_w[w] = acc; acc += t; } s_w[HD_NW] = acc; }
  __syncthreads();
  *total = s_w[HD_NW];
  const int r = s_w[wave] + inc - v;
  __syncthreads();      // s_w may be reused at once
  return r;
}

// workgroup arg-best of (value, index): larger value wins (want_max) or smaller; ties -> smaller index.  s_v / s_i: HD_NW.
__device__ inline void hd_block_argbest(double& v, int& i, bool want_max, double* s_v, int* s_i) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int off = 32; off > 0; off >>= 1) {
    const double ov = __shfl_down(v, off);
    const int oi = __shfl_down(i, off);
    const bool better = want_max ? (ov > v) : (ov < v);
    if (oi >= 0 && (i < 0 || better || (ov == v && oi < i))) { v = ov; i = oi; }
  }
  if (lane == 0) { s_v[wave] = v; s_i[wave] = i; }
  __syncthreads();
  if (threadIdx.x == 0) {
    double bv = s_v[0]; int bi = s_i[0];
    for (int w = 1; w < HD_NW; ++w) {
      const double ov = s_v[w]; const int oi = s_i[w];
      const bool better = want_max ? (ov > bv) : (ov < bv);
      if (oi >= 0 && (bi < 0 || better || (ov == bv && oi < bi))) { bv = ov; bi = oi; }
    }
    s_v[0] = bv; s_i[0] = bi;
  }
  __syncthreads();
  v = s_v[0]; i = s_i[0];
  __syncthreads();
}

struct HdPlane { double nx, ny, nz, d; };

__device__ inline void hd_point(const float* P, int i, const double* c, double* o) {
  o[0] = (double)P[3 * i] - c[0]; o[1] = (double)P[3 * i + 1] - c[1]; o[2] = (double)P[3 * i + 2] - c[2];
}
// plane of (a, b, cc): Builder::set_plane of sh_hull.h, operation for operation
__device__ inline HdPlane hd_plane(const double* a, const double* b, const double* cc) {
  const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w[3] = {cc[0] - a[0], cc[1] - a[1], cc[2] - a[2]};
  const double nx = u[1] * w[2] - u[2] * w[1], ny = u[2] * w[0] - u[0] * w[2], nz = u[0] * w[1] - u[1] * w[0];
  const double l = sqrt(nx * nx + ny * ny + nz * nz);
  HdPlane p;
  if (l == 0.0) { p.nx = 0; p.ny = 0; p.nz = 1; p.d = a[2]; return p; }
  p.nx = nx / l; p.ny = ny / l; p.nz = nz / l;
  p.d = p.nx * a[0] + p.ny * a[1] + p.nz * a[2];
  return p;
}
__device__ inline double hd_dist(const HdPlane& f, const double* p) { return f.nx * p[0] + f.ny * p[1] + f.nz * p[2] - f.d; }
__device__ inline unsigned long long hd_key(double d, int q) {      // (distance truncated to 43 bits, smaller index wins ties); d > 0
  return ((unsigned long long)__double_as_longlong(d) & ~0x1FFFFFull) | (unsigned long long)(0x1FFFFF - q);
}

// edge table: claim (atomicMin of the priority) / insert-unique / look up.  Returns false when the probe limit is hit.
__device__ inline bool hd_tbl_slot(unsigned long long* tkeys, unsigned long long key, bool insert, int* slot) {
  unsigned h = hd_hash_key(key) & (HD_TBL - 1);
  for (int probe = 0; probe < 2048; ++probe) {
    unsigned long long k;
    if (insert) k = atomicCAS(&tkeys[h], 0ull, key);
    else k = __hip_atomic_load(&tkeys[h], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (k == key || (insert && k == 0ull)) { *slot = (int)h; return true; }
    if (!insert && k == 0ull) return false;
    h = (h + 1) & (HD_TBL - 1);
  }
  return false;
}

// the record of the unit tetrahedron (0, e1, e2, e3): 4 vertices, 4 outward faces, 6 edges (va -> vb in the winding of face f)
__device__ inline void hd_unit_tetrahedron(double* HV, double* NR, int* ED, int* nv, int* nf, int* ne) {
  const double v[12] = {0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0, 1};
  const double r3 = 0.57735026918962573;
  const double nr[12] = {0, 0, -1, 0, -1, 0, -1, 0, 0, r3, r3, r3};      // faces (0,2,1) (0,1,3) (0,3,2) (1,2,3)
  const int ed[24] = {0, 1, 1, 0, 0, 2, 0, 2, 0, 3, 2, 1, 1, 2, 3, 0, 1, 3, 1, 3, 2, 3, 3, 2};
  for (int i = 0; i < 12; ++i) { HV[i] = v[i]; NR[i] = nr[i]; }
  for (int i = 0; i < 24; ++i) ED[i] = ed[i];
  *nv = 4; *nf = 4; *ne = 6;
}

__global__ void __launch_bounds__(HD_THREADS)
k_hull_rounds(const float* __restrict__ kept, const long long* __restrict__ koff, HullScratch sc,
              double* __restrict__ hv, double* __restrict__ normals, int* __restrict__ edges,
              int* __restrict__ nv_out, int* __restrict__ nf_out, int* __restrict__ ne_out,
              int* __restrict__ fail_out /*[B]: 0 or a positive reason*/, int* __restrict__ rounds_out /*[B] (nullable)*/,
              const int* __restrict__ skip /*[B] (nullable): 1 = this humerus' record is already in place (host quickhull), leave it*/,
              const HullCap hc) {
  __shared__ HdPlane s_pl[HD_SLOTS];                       // 98 304 B
  __shared__ unsigned long long s_key[HD_SLOTS];           // 24 576 B  apex keys; face ids at the end
  __shared__ short s_conf[HD_NMAX];                        // 16 384 B  conflict face of a point (-1 inside, -2 inserted); vertex ids at the end
  __shared__ unsigned char s_alive[HD_SLOTS];              // 0 dead, 1 alive, 2 + ci: killed in this round by candidate ci
  __shared__ int c_face[HD_KC], c_pt[HD_KC], c_ok[HD_KC], c_nvis[HD_KC], c_nh[HD_KC], c_off[HD_KC], c_koff[HD_KC];
  __shared__ unsigned c_prio[HD_KC];
  __shared__ short s_visl[HD_KC][HD_VL];                   // visible faces of a candidate (the common case: <= 32)
  __shared__ int2 s_hz[HD_NW][64];                         // per wave: horizon edges being compacted
  __shared__ short s_newslot[HD_NSL];
  __shared__ int s_ncand, s_tot[3];
  __shared__ int s_w[HD_NW + 1];
  __shared__ double s_rv[HD_NW];
  __shared__ int s_ri[HD_NW];
  __shared__ double s_c[3];
  __shared__ int s_fail;
  __shared__ unsigned s_minp;

  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  if (skip != nullptr && skip[b] != 0) { if (tid == 0) { fail_out[b] = 0; if (rounds_out) rounds_out[b] = 0; } return; }
  const float* P = kept + 3 * koff[b];
  const int n = (int)(koff[b + 1] - koff[b]);
  int* fv = sc.fv + (size_t)b * HD_SLOTS * 3;
  int* visg = sc.vis + (size_t)b * HD_KC * HD_VMAX;
  int* evg = sc.ev + (size_t)b * HD_KC * 3 * HD_VMAX * 2;
  int* horg = sc.hor + (size_t)b * HD_KC * (HD_VMAX + 2) * 2;
  int* newslot = sc.newslot + (size_t)b * HD_SLOTS;
  int* freestack = sc.freestack + (size_t)b * HD_SLOTS;
  unsigned long long* tkeys = sc.tkeys + (size_t)b * HD_TBL;
  unsigned* tvals = sc.tvals + (size_t)b * HD_TBL;
  double* HV = hv + (size_t)b * hc.v * 3;
  double* NR = normals + (size_t)b * hc.f * 3;
  int* ED = edges + (size_t)b * hc.e * 4;
  // failure: the reason for the host (which then runs its own quickhull for this batch) and a small WELL-FORMED record -- the unit
  // tetrahedron -- so that the kernels already queued behind this one (candidate boxes, frame, slices ...) work on finite, in-range
  // data for this humerus; its status word says that its landmarks are void
#define HD_FAIL(code) do { if (tid == 0) { fail_out[b] = (code); if (rounds_out) rounds_out[b] = 0; hd_unit_tetrahedron(HV, NR, ED, nv_out + b, nf_out + b, ne_out + b); } return; } while (0)
  if (n < 4) HD_FAIL(1);
  if (n > HD_NMAX) HD_FAIL(40);
  if (tid == 0) s_fail = 0;

  // ---- bounding box -> centre (midpoint: order independent), eps
  double c[3], eps, diag;
  {
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int i = tid; i < n; i += HD_THREADS)
      for (int k = 0; k < 3; ++k) { const double v = (double)P[3 * i + k]; lo[k] = fmin(lo[k], v); hi[k] = fmax(hi[k], v); }
    for (int k = 0; k < 3; ++k) {
      double v = lo[k]; int i = 0;
      hd_block_argbest(v, i, false, s_rv, s_ri); lo[k] = v;
      v = hi[k]; i = 0;
      hd_block_argbest(v, i, true, s_rv, s_ri); hi[k] = v;
    }
    for (int k = 0; k < 3; ++k) c[k] = 0.5 * (lo[k] + hi[k]);
    diag = sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    eps = HD_EPS_REL * diag;
    if (tid < 3) s_c[tid] = c[tid];
  }
  // ---- initial simplex (first index wins every tie, as in sh_hull.h)
  int i0, i1, i2, i3;
  {
    double v = 1e300; int i = -1;
    for (int q = tid; q < n; q += HD_THREADS) { const double x = (double)P[3 * q] - c[0]; if (i < 0 || x < v) { v = x; i = q; } }
    hd_block_argbest(v, i, false, s_rv, s_ri); i0 = i;
    v = -1e300; i = -1;
    for (int q = tid; q < n; q += HD_THREADS) { const double x = (double)P[3 * q] - c[0]; if (i < 0 || x > v) { v = x; i = q; } }
    hd_block_argbest(v, i, true, s_rv, s_ri); i1 = i;
    if (i0 == i1) HD_FAIL(2);
    double p0[3], p1[3];
    hd_point(P, i0, c, p0); hd_point(P, i1, c, p1);
    const double e[3] = {p1[0] - p0[0], p1[1] - p0[1], p1[2] - p0[2]};
    v = 0.0; i = -1;
    for (int q = tid; q < n; q += HD_THREADS) {
      double pq[3]; hd_point(P, q, c, pq);
      const double w[3] = {pq[0] - p0[0], pq[1] - p0[1], pq[2] - p0[2]};
      const double cx = e[1] * w[2] - e[2] * w[1], cy = e[2] * w[0] - e[0] * w[2], cz = e[0] * w[1] - e[1] * w[0];
      const double a2 = cx * cx + cy * cy + cz * cz;
      if (a2 > v) { v = a2; i = q; }
    }
    hd_block_argbest(v, i, true, s_rv, s_ri); i2 = i;
    if (i2 < 0) HD_FAIL(3);
    double p2[3]; hd_point(P, i2, c, p2);
    const HdPlane tmp = hd_plane(p0, p1, p2);
    v = 0.0; i = -1;
    for (int q = tid; q < n; q += HD_THREADS) { double pq[3]; hd_point(P, q, c, pq); const double d = fabs(hd_dist(tmp, pq)); if (d > v) { v = d; i = q; } }
    hd_block_argbest(v, i, true, s_rv, s_ri); i3 = i;
    if (i3 < 0 || v <= eps) HD_FAIL(4);
    double p3[3]; hd_point(P, i3, c, p3);
    if (hd_dist(tmp, p3) > 0) { const int t = i1; i1 = i2; i2 = t; }
  }
  for (int f = tid; f < HD_SLOTS; f += HD_THREADS) { s_alive[f] = 0; s_key[f] = 0ull; }
  for (int i = tid; i < HD_TBL; i += HD_THREADS) { tkeys[i] = 0ull; tvals[i] = 0xFFFFFFFFu; }
  __syncthreads();
  if (tid < 4) {
    const int init[4][3] = {{i0, i1, i2}, {i0, i3, i1}, {i1, i3, i2}, {i2, i3, i0}};
    double a[3], bb[3], cc[3];
    hd_point(P, init[tid][0], c, a); hd_point(P, init[tid][1], c, bb); hd_point(P, init[tid][2], c, cc);
    s_pl[tid] = hd_plane(a, bb, cc);
    fv[3 * tid] = init[tid][0]; fv[3 * tid + 1] = init[tid][1]; fv[3 * tid + 2] = init[tid][2];
    s_alive[tid] = 1;
  }
  __syncthreads();
  for (int q = tid; q < n; q += HD_THREADS) {
    short cf = -1;
    if (q == i0 || q == i1 || q == i2 || q == i3) cf = -2;
    else {
      double pq[3]; hd_point(P, q, c, pq);
      for (int f = 0; f < 4; ++f) { const double d = hd_dist(s_pl[f], pq); if (d > eps) { cf = (short)f; atomicMax(&s_key[f], hd_key(d, q)); break; } }
    }
    s_conf[q] = cf;
  }
  int nslots = 4, nfree = 0, tbl_used = 0, rounds = 0;
  unsigned long long stamp = 0;
  __syncthreads();

  constexpr int RNG = HD_SLOTS / HD_NW;                     // slots a wave scans in R0 (192 = 3 x 64)
#ifdef SH_HULL_PROF
  unsigned long long pt0 = __builtin_amdgcn_s_memrealtime(), pacc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#define HD_STAMP(i) do { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); pacc[i] += t_ - pt0; pt0 = t_; } while (0)
#else
#define HD_STAMP(i) do { } while (0)
#endif
  for (int round = 0; round < HD_MAXROUNDS; ++round) {
    // ---- R0: candidates.  Every wave scans its own slot range; counts and the smallest priority meet in LDS
    if (tid == 0) { s_ncand = 0; s_minp = 0xFFFFFFFFu; }
    __syncthreads();
    int pt_[RNG / 64]; unsigned h_[RNG / 64]; bool cand_[RNG / 64];
    {
      int wc = 0; unsigned wmin = 0xFFFFFFFFu;
#pragma unroll
      for (int it = 0; it < RNG / 64; ++it) {
        const int f = wave * RNG + it * 64 + lane;
        cand_[it] = f < nslots && s_alive[f] == 1 && s_key[f] != 0ull;
        pt_[it] = cand_[it] ? 0x1FFFFF - (int)(s_key[f] & 0x1FFFFF) : 0;
        h_[it] = hd_hash32((unsigned)(pt_[it] + round * 0x9E3779B));
        wc += __popcll(__ballot(cand_[it]));
        if (cand_[it]) wmin = min(wmin, h_[it]);
      }
      for (int off = 32; off > 0; off >>= 1) wmin = min(wmin, (unsigned)__shfl_down((int)wmin, off));
      if (lane == 0 && wc) { atomicAdd(&s_ncand, wc); atomicMin(&s_minp, wmin); }
    }
    __syncthreads();
    const int ncand = s_ncand;
    if (ncand == 0) break;
    const unsigned minp = s_minp;
    rounds = round + 1;
    const unsigned T = ncand <= 64 ? 0xFFFFFFFFu : (unsigned)((64ull << 32) / (unsigned long long)ncand);
    {
      int sc_ = 0;
#pragma unroll
      for (int it = 0; it < RNG / 64; ++it) { cand_[it] = cand_[it] && (h_[it] <= T || h_[it] == minp); sc_ += __popcll(__ballot(cand_[it])); }
      if (lane == 0) s_w[wave] = sc_;
    }
    __syncthreads();
    int nsel = 0;
    {
      int base = 0;
      for (int w = 0; w < HD_NW; ++w) { const int t = s_w[w]; if (w < wave) base += t; nsel += t; }
#pragma unroll
      for (int it = 0; it < RNG / 64; ++it) {
        const unsigned long long m = __ballot(cand_[it]);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (cand_[it] && pos < HD_KC) {
          c_face[pos] = wave * RNG + it * 64 + lane; c_pt[pos] = pt_[it]; c_prio[pos] = (h_[it] & ~127u) | (unsigned)pos; c_ok[pos] = 0; c_nh[pos] = 0; c_nvis[pos] = 0;
        }
        base += __popcll(m);
      }
    }
    if (nsel > HD_KC) nsel = HD_KC;
    if (tbl_used > HD_TBL / 4) {      // stale keys of earlier rounds (other stamps) only lengthen the probes: sweep them out now and then
      for (int i = tid; i < HD_TBL; i += HD_THREADS) { tkeys[i] = 0ull; tvals[i] = 0xFFFFFFFFu; }
      tbl_used = 0;
    }
    stamp = (unsigned long long)(round + 1) << 26;
    __syncthreads();
    HD_STAMP(0);
    // ---- R1: visible faces + edge claims; wave w serves candidates w, w + 16, ... (the same wave in every phase).  A candidate
    // that sees at most 21 faces (63 directed edges: the rule, a later round's apex sees ~6) keeps one edge per lane in
    // registers from here to R5; larger ones go through the global scratch.
    int ea[HD_NJ], eb[HD_NJ], ha[HD_NJ], hb[HD_NJ], nvis_[HD_NJ], tsl[HD_NJ];
    {
      // one pass over the planes serves the wave's four candidates (a plane is read from LDS once)
      double pq[HD_NJ][3];
      bool valid[HD_NJ];
#pragma unroll
      for (int j = 0; j < HD_NJ; ++j) {
        const int ci = wave + j * HD_NW;
        ea[j] = eb[j] = ha[j] = hb[j] = -1; nvis_[j] = 0; tsl[j] = -1;
        valid[j] = ci < nsel;
        hd_point(P, c_pt[valid[j] ? ci : 0], c, pq[j]);
      }
      for (int f0 = 0; f0 < nslots; f0 += 64) {
        const int f = f0 + lane;
        const bool inb = f < nslots && s_alive[f] == 1;
        const HdPlane pl = s_pl[inb ? f : 0];
#pragma unroll
        for (int j = 0; j < HD_NJ; ++j) {
          const int ci = wave + j * HD_NW;
          const bool vis = inb && valid[j] && hd_dist(pl, pq[j]) > eps;
          const unsigned long long m = __ballot(vis);
          if (m) {
            const int pos = nvis_[j] + __popcll(m & ((1ull << lane) - 1ull));
            if (vis) { if (pos < HD_VL) s_visl[ci][pos] = (short)f; else if (pos < HD_VMAX) visg[ci * HD_VMAX + pos] = f; }
            nvis_[j] += __popcll(m);
          }
        }
      }
      bool any_big = false;
#pragma unroll
      for (int j = 0; j < HD_NJ; ++j) {
        const int ci = wave + j * HD_NW;
        if (!valid[j]) continue;
        if (lane == 0) { c_nvis[ci] = nvis_[j]; if (nvis_[j] > HD_VMAX) atomicMax(&s_fail, 20); if (nvis_[j] == 0) atomicMax(&s_fail, 25); }
        any_big |= nvis_[j] > HD_VL;
        nvis_[j] = min(nvis_[j], HD_VMAX);
      }
      if (any_big) __threadfence_block();      // (the part of a long list that went to global scratch is read back by other lanes)
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      // edges of the short lists (<= 21 faces): one per lane; the loads / atomics of the four candidates are issued together
      unsigned long long key_[HD_NJ];
#pragma unroll
      for (int j = 0; j < HD_NJ; ++j) {
        const int ci = wave + j * HD_NW;
        key_[j] = 0ull;
        if (valid[j] && 3 * nvis_[j] <= 64 && lane < 3 * nvis_[j]) {
          const int f = (int)s_visl[ci][lane / 3], k = lane % 3;
          ea[j] = fv[3 * f + k]; eb[j] = fv[3 * f + (k + 1) % 3];
        }
      }
      unsigned long long got[HD_NJ]; unsigned hh[HD_NJ];
#pragma unroll
      for (int j = 0; j < HD_NJ; ++j) {
        got[j] = ~0ull; hh[j] = 0;
        if (ea[j] >= 0) {
          key_[j] = stamp | ((unsigned long long)min(ea[j], eb[j]) << 13) | (unsigned long long)max(ea[j], eb[j]);
          hh[j] = hd_hash_key(key_[j]) & (HD_TBL - 1);
          got[j] = atomicCAS(&tkeys[hh[j]], 0ull, key_[j]);
        }
      }
#pragma unroll
      for (int j = 0; j < HD_NJ; ++j) {
        if (ea[j] < 0) continue;
        int sl = (int)hh[j];
        if (got[j] != 0ull && got[j] != key_[j]) {      // first probe taken by another edge: walk on
          if (!hd_tbl_slot(tkeys, key_[j], true, &sl)) { atomicMax(&s_fail, 26); sl = -1; }
        }
        tsl[j] = sl;
        if (sl >= 0) atomicMin(&tvals[sl], c_prio[wave + j * HD_NW]);
      }
      // long lists: the sequential form through global scratch
#pragma unroll
      for (int j = 0; j < HD_NJ; ++j) {
        const int ci = wave + j * HD_NW;
        if (!valid[j] || 3 * nvis_[j] <= 64) continue;
        const int nvis = nvis_[j];
        const unsigned prio = c_prio[ci];
        for (int e = lane; e < 3 * nvis; e += 64) {
          const int f = e / 3 < HD_VL ? (int)s_visl[ci][e / 3] : visg[ci * HD_VMAX + e / 3], k = e % 3;
          const int a = fv[3 * f + k], bq = fv[3 * f + (k + 1) % 3];
          evg[(ci * 3 * HD_VMAX + e) * 2] = a; evg[(ci * 3 * HD_VMAX + e) * 2 + 1] = bq;
          const unsigned long long key = stamp | ((unsigned long long)min(a, bq) << 13) | (unsigned long long)max(a, bq);
          int sl;
          if (hd_tbl_slot(tkeys, key, true, &sl)) atomicMin(&tvals[sl], prio); else atomicMax(&s_fail, 26);
        }
      }
    }
    __syncthreads();
    HD_STAMP(1);
    if (s_fail) break;
    // ---- R2 + R3: ownership, horizon
    unsigned tv[HD_NJ];
#pragma unroll
    for (int j = 0; j < HD_NJ; ++j) tv[j] = tsl[j] >= 0 ? __hip_atomic_load(&tvals[tsl[j]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
#pragma unroll
    for (int j = 0; j < HD_NJ; ++j) {
      const int ci = wave + j * HD_NW;
      if (ci >= nsel) continue;
      const int nvis = c_nvis[ci], ne3 = 3 * nvis;
      const unsigned prio = c_prio[ci];
      const bool fast = ne3 <= 64;
      bool mine = true;
      if (fast) {      // the claim remembered its table slot
        if (lane < ne3) mine = tsl[j] >= 0 && tv[j] == prio;
      } else
      for (int e = lane; e < ne3; e += 64) {
        const int a = evg[(ci * 3 * HD_VMAX + e) * 2], bq = evg[(ci * 3 * HD_VMAX + e) * 2 + 1];
        const unsigned long long key = stamp | ((unsigned long long)min(a, bq) << 13) | (unsigned long long)max(a, bq);
        int sl;
        if (!hd_tbl_slot(tkeys, key, false, &sl)) { atomicMax(&s_fail, 27); mine = false; }
        else if (__hip_atomic_load(&tvals[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != prio) mine = false;
      }
      const bool ok = __all(mine);
      if (!ok) continue;
      int nh = 0;
      if (fast) {
        // every lane holds one directed edge: its reverse is searched among the others by shuffles
        const int a = ea[j], bq = eb[j];
        bool twin = false;
        for (int e2 = 0; e2 < ne3; ++e2) { const int a2 = __shfl(a, e2), b2 = __shfl(bq, e2); twin |= (a2 == bq && b2 == a); }
        const bool hz = lane < ne3 && !twin;
        const unsigned long long m = __ballot(hz);
        nh = __popcll(m);
        if (hz) s_hz[wave][__popcll(m & ((1ull << lane) - 1ull))] = make_int2(a, bq);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_wave_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (nh < 3) { if (lane == 0) atomicMax(&s_fail, 21); continue; }
        const int2 he = lane < nh ? s_hz[wave][lane] : make_int2(-1, -2);
        ha[j] = he.x; hb[j] = he.y;
        // one simple loop: from edge 0 follow end -> the unique edge starting there; back at edge 0 after exactly nh steps
        int cur = 0, steps = 0, bad = 0;
        do {
          const int target = __shfl(hb[j], cur);
          const unsigned long long mm = __ballot(lane < nh && ha[j] == target);
          if (__popcll(mm) != 1) { bad = 22; break; }
          cur = __ffsll((long long)mm) - 1;
        } while (++steps < nh && cur != 0);
        if (!bad && (cur != 0 || steps != nh)) bad = 23;
        if (bad) { if (lane == 0) atomicMax(&s_fail, bad); continue; }
      } else {
        for (int e0 = 0; e0 < ne3; e0 += 64) {
          const int e = e0 + lane;
          bool hz = false; int a = 0, bq = 0;
          if (e < ne3) {
            a = evg[(ci * 3 * HD_VMAX + e) * 2]; bq = evg[(ci * 3 * HD_VMAX + e) * 2 + 1];
            bool twin = false;
            for (int e2 = 0; e2 < ne3; ++e2)
              if (evg[(ci * 3 * HD_VMAX + e2) * 2] == bq && evg[(ci * 3 * HD_VMAX + e2) * 2 + 1] == a) { twin = true; break; }
            hz = !twin;
          }
          const unsigned long long m = __ballot(hz);
          const int pos = nh + __popcll(m & ((1ull << lane) - 1ull));
          if (hz && pos < HD_VMAX + 2) { horg[(ci * (HD_VMAX + 2) + pos) * 2] = a; horg[(ci * (HD_VMAX + 2) + pos) * 2 + 1] = bq; }
          nh += __popcll(m);
        }
        __threadfence_block();
        if (nh < 3 || nh > HD_VMAX + 2) { if (lane == 0) atomicMax(&s_fail, 21); continue; }
        int cur = 0, steps = 0, bad = 0;
        do {
          const int target = horg[(ci * (HD_VMAX + 2) + cur) * 2 + 1];
          int found = -1, cnt = 0;
          for (int k0 = 0; k0 < nh; k0 += 64) {
            const int k = k0 + lane;
            const bool hit = k < nh && horg[(ci * (HD_VMAX + 2) + k) * 2] == target;
            const unsigned long long mm = __ballot(hit);
            if (mm && found < 0) found = k0 + __ffsll((long long)mm) - 1;
            cnt += __popcll(mm);
          }
          if (cnt != 1) { bad = 22; break; }
          cur = found;
        } while (++steps < nh && cur != 0);
        if (!bad && (cur != 0 || steps != nh)) bad = 23;
        if (bad) { if (lane == 0) atomicMax(&s_fail, bad); continue; }
      }
      if (lane == 0) { c_nh[ci] = nh; c_ok[ci] = 1; }
    }
    __syncthreads();
    HD_STAMP(2);
    if (s_fail) break;
    // ---- R4: slots.  Wave 0: offsets of the new faces / of the dead faces per candidate (candidate order: one per lane)
    if (wave == 0) {
      const bool okc = lane < nsel && c_ok[lane];
      int v1 = okc ? c_nh[lane] : 0, v2 = okc ? c_nvis[lane] : 0, v3 = lane < nsel ? 3 * min(c_nvis[lane], HD_VMAX) : 0;
      int i1 = v1, i2 = v2, i3 = v3;
      for (int off = 1; off < 64; off <<= 1) {
        const int t1 = __shfl_up(i1, off), t2 = __shfl_up(i2, off), t3 = __shfl_up(i3, off);
        if (lane >= off) { i1 += t1; i2 += t2; i3 += t3; }
      }
      if (lane < nsel) { c_off[lane] = i1 - v1; c_koff[lane] = i2 - v2; }
      if (lane == 63) { s_tot[0] = i1; s_tot[1] = i2; s_tot[2] = i3; }
    }
    __syncthreads();
    const int total_new = s_tot[0], total_dead = s_tot[1];
    tbl_used += s_tot[2];
    const int pops = min(total_new, nfree), nslots_new = nslots + (total_new - pops), nfree_mid = nfree - pops;
    if (nslots_new > HD_SLOTS) { if (tid == 0) s_fail = 24; __syncthreads(); break; }
    for (int jn = tid; jn < total_new; jn += HD_THREADS) {
      const int sl = jn < nfree ? freestack[nfree - 1 - jn] : nslots + (jn - nfree);
      if (jn < HD_NSL) s_newslot[jn] = (short)sl;
      if (total_new > HD_NSL) newslot[jn] = sl;
    }
    __syncthreads();
    HD_STAMP(3);
    // ---- R5: kill, create
#pragma unroll
    for (int j = 0; j < HD_NJ; ++j) {
      const int ci = wave + j * HD_NW;
      if (ci >= nsel || !c_ok[ci]) continue;
      const int nvis = c_nvis[ci], nh = c_nh[ci], off = c_off[ci], koffc = c_koff[ci], pt = c_pt[ci];
      const bool fast = 3 * nvis <= 64;
      for (int i = lane; i < nvis; i += 64) {
        const int f = i < HD_VL ? (int)s_visl[ci][i] : visg[ci * HD_VMAX + i];
        s_alive[f] = (unsigned char)(2 + ci); s_key[f] = 0ull;
        freestack[nfree_mid + koffc + i] = f;
      }
      double pp[3]; hd_point(P, pt, c, pp);
      for (int k = lane; k < nh; k += 64) {
        const int sl = total_new > HD_NSL ? newslot[off + k] : (int)s_newslot[off + k];
        const int a = fast ? ha[j] : horg[(ci * (HD_VMAX + 2) + k) * 2], bq = fast ? hb[j] : horg[(ci * (HD_VMAX + 2) + k) * 2 + 1];
        double pa[3], pb[3];
        hd_point(P, a, c, pa); hd_point(P, bq, c, pb);
        s_pl[sl] = hd_plane(pa, pb, pp);
        fv[3 * sl] = a; fv[3 * sl + 1] = bq; fv[3 * sl + 2] = pt;
        s_alive[sl] = 1; s_key[sl] = 0ull;
      }
    }
    __syncthreads();
    HD_STAMP(4);
    // ---- R6: the points of the dead faces move
    for (int q = tid; q < n; q += HD_THREADS) {
      const int f = s_conf[q];
      if (f < 0) continue;
      const int st = s_alive[f];
      if (st < 2) continue;
      const int ci = st - 2;
      if (q == c_pt[ci]) { s_conf[q] = -2; continue; }
      double pq[3]; hd_point(P, q, c, pq);
      short cf = -1;
      const int nh = c_nh[ci], off = c_off[ci];
      for (int k = 0; k < nh; ++k) {
        const int sl = total_new > HD_NSL ? newslot[off + k] : (int)s_newslot[off + k];
        const double d = hd_dist(s_pl[sl], pq);
        if (d > eps) { cf = (short)sl; atomicMax(&s_key[sl], hd_key(d, q)); break; }
      }
      s_conf[q] = cf;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < HD_NJ; ++j) {
      const int ci = wave + j * HD_NW;
      if (ci >= nsel || !c_ok[ci]) continue;
      const int nvis = c_nvis[ci];
      for (int i = lane; i < nvis; i += 64) s_alive[i < HD_VL ? (int)s_visl[ci][i] : visg[ci * HD_VMAX + i]] = 0;
    }
    nslots = nslots_new; nfree = nfree_mid + total_dead;
    HD_STAMP(5);
  }
#ifdef SH_HULL_PROF
  if (tid == 0 && b == 0) printf("hull prof (100 MHz ticks): R0 %llu R1 %llu R2R3 %llu R4 %llu R5 %llu R6+ %llu rounds %d\n", pacc[0], pacc[1], pacc[2], pacc[3], pacc[4], pacc[5], rounds);
#endif
  __syncthreads();
  if (s_fail) HD_FAIL(s_fail);
  if (rounds >= HD_MAXROUNDS) HD_FAIL(28);

  // ---- emit the record.  Face ids (slot order) -> s_key; vertex ids (point order) -> s_conf
  int nf = 0, nvh = 0, neh = 0;
  for (int q = tid; q < n; q += HD_THREADS) s_conf[q] = 0;
  __syncthreads();
  for (int f0 = 0; f0 < nslots; f0 += HD_THREADS) {
    const int f = f0 + tid;
    const int flag = (f < nslots && s_alive[f] == 1) ? 1 : 0;
    int t2;
    const int pos = nf + hd_block_scan(flag, s_w, &t2);
    if (flag) { s_key[f] = (unsigned long long)pos; for (int k = 0; k < 3; ++k) s_conf[fv[3 * f + k]] = 1; }
    nf += t2;
  }
  __syncthreads();
  for (int q0 = 0; q0 < n; q0 += HD_THREADS) {
    const int q = q0 + tid;
    const int flag = (q < n && s_conf[q]) ? 1 : 0;
    int t2;
    const int pos = nvh + hd_block_scan(flag, s_w, &t2);
    if (flag) {
      s_conf[q] = (short)pos;
      if (pos < hc.v) { HV[3 * pos] = (double)P[3 * q]; HV[3 * pos + 1] = (double)P[3 * q + 1]; HV[3 * pos + 2] = (double)P[3 * q + 2]; }
    }
    nvh += t2;
  }
  if (nvh > hc.v || nf > hc.f) HD_FAIL(41);
  // normals: triangle rotated to its smallest vertex first, centred coordinates (sh_hull.h writes the same)
  for (int f = tid; f < nslots; f += HD_THREADS) {
    if (s_alive[f] != 1) continue;
    int v[3] = {fv[3 * f], fv[3 * f + 1], fv[3 * f + 2]};
    int r = 0;
    if (v[1] < v[r]) r = 1;
    if (v[2] < v[r]) r = 2;
    double a[3], bb[3], cc[3];
    hd_point(P, v[r], c, a); hd_point(P, v[(r + 1) % 3], c, bb); hd_point(P, v[(r + 2) % 3], c, cc);
    const HdPlane pl = hd_plane(a, bb, cc);
    const int fid = (int)s_key[f];
    NR[3 * fid] = pl.nx; NR[3 * fid + 1] = pl.ny; NR[3 * fid + 2] = pl.nz;
  }
  // edges: directed edge -> face in the table, then every a < b edge looks up its reverse
  for (int i = tid; i < HD_TBL; i += HD_THREADS) { tkeys[i] = 0ull; tvals[i] = 0xFFFFFFFFu; }
  __syncthreads();
  for (int e = tid; e < 3 * nslots; e += HD_THREADS) {
    const int f = e / 3, k = e % 3;
    if (s_alive[f] != 1) continue;
    const int a = fv[3 * f + k], bq = fv[3 * f + (k + 1) % 3];
    const unsigned long long key = (1ull << 40) | ((unsigned long long)a << 13) | (unsigned long long)bq;
    int sl;
    if (!hd_tbl_slot(tkeys, key, true, &sl)) { atomicMax(&s_fail, 29); continue; }
    if (atomicMin(&tvals[sl], (unsigned)f) != 0xFFFFFFFFu) atomicMax(&s_fail, 30);      // a directed edge twice: not a manifold
  }
  __syncthreads();
  for (int e0 = 0; e0 < 3 * nslots; e0 += HD_THREADS) {
    const int e = e0 + tid, f = e / 3, k = e % 3;
    int flag = 0, a = 0, bq = 0;
    if (e < 3 * nslots && s_alive[f] == 1) { a = fv[3 * f + k]; bq = fv[3 * f + (k + 1) % 3]; flag = a < bq ? 1 : 0; }
    int t2;
    const int pos = neh + hd_block_scan(flag, s_w, &t2);
    if (flag) {
      const unsigned long long key = (1ull << 40) | ((unsigned long long)bq << 13) | (unsigned long long)a;
      int sl;
      if (!hd_tbl_slot(tkeys, key, false, &sl)) atomicMax(&s_fail, 31);
      else if (pos < hc.e) {
        const int g = (int)__hip_atomic_load(&tvals[sl], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        ED[4 * pos] = s_conf[a]; ED[4 * pos + 1] = s_conf[bq]; ED[4 * pos + 2] = (int)s_key[f]; ED[4 * pos + 3] = (int)s_key[g];
      }
    }
    neh += t2;
  }
  __syncthreads();
  if (s_fail) HD_FAIL(s_fail);
  if (neh > hc.e || nvh - neh + nf != 2 || 2 * neh != 3 * nf) HD_FAIL(32);
  if (tid == 0) { nv_out[b] = nvh; nf_out[b] = nf; ne_out[b] = neh; fail_out[b] = 0; if (rounds_out) rounds_out[b] = rounds; }
#undef HD_FAIL
}

// a humerus whose hull the device gave up is flagged in the per-mesh status words (the host re-runs the batch with its own hull)
__global__ void k_hull_flag(const int* __restrict__ fail, int* __restrict__ err, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B && fail[b] != 0) err[b] = SH_ERR_HULL_DEV;
}

}  // namespace sh
