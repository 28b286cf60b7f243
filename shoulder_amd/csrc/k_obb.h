// k_obb.h -- oriented bounding box + head-end detection on the device
// (reference src/shoulder/humerus/mesh.py:63-125; trimesh `oriented_bounds` semantics restated in
// oracle/obb.py, canonical rule B-3).
//   k_obb_candidates  one workgroup per (16 hull faces, humerus): for every face normal find the silhouette
//                     edges of the hull (= edges of the 2-D hull of its projection on the face plane) and
//                     for each take the enclosing rectangle -> min area x height = candidate volume
//   k_obb_face_area2 / k_obb_bounds / k_obb_select   lower bound of every direction's box volume (Cauchy: projected area =
//                     1/2 sum_f A_f |n . n_f|, times the height) -> only the directions whose bound does not exceed the best
//                     exact volume found so far are handed to k_obb_candidates (two passes: seed, survivors)
//   k_obb_pick        argmin volume -> axes ordered by ascending extent, signs fixed by vertex 0,
//                     box centred at the origin  -> T_pre (CT -> raw OBB)
//   k_obb_end_points  sections at 0.95*zmin / 0.95*zmax (mesh.py:91-99), crossing points only
//   k_obb_ends        circle-fit residual of both ends (mesh.py:102) -> flip (mesh.py:105-124)
// Hull record per humerus (host quickhull, sh_hull.h): hv [HV][3] f64 CT coords, normals [HF][3],
// edges [HE][4] = (va, vb, face f, face g).
#pragma once
#include <type_traits>
#include "k_te.h"

namespace sh {

// hull record capacities (HBM): vertices / faces / edges per humerus.  The fixtures' hulls have 1 368 / 2 732 / 4 098; the hull of
// a 519 k-triangle humerus 4 209 / 8 414 / 12 621; a convex region sampled more densely keeps more of its vertices on the hull.
// These are the capacities a context STARTS with; a batch with a larger hull grows the record (sh_ctx::hcap) and every kernel
// takes the per-humerus strides as an argument.
#define SH_HV 16384
#define SH_HF 32768
#define SH_HE 49152
struct HullCap { int v, f, e; };      // per-humerus strides of hull.hv / hull.normals (+ the per-face obb.* arrays) / hull.edges
#define SH_ENDCAP 8192      // crossing points of an end section (mesh.py:91-107) a context starts with; ~330 at the fixture resolution, ~1 300 on a 519 k-triangle mesh.
                            // A run that meets more records how many (overflow counter word 7) and sh_collect grows the buffer and runs the batch again

__device__ inline void obb_basis(const double* n, double* u, double* v) { plane_basis(n, u, v); }

#define SH_OBB_THREADS 256
#define SH_OBB_TILE 16           // hull faces (candidate directions) per workgroup
#define SH_OBB_GROUP 4           // faces whose rectangle scans run together (2 would fit three workgroups per CU: measured slower)
#define SH_SIL_MAX 512           // silhouette edges per direction

// ---- pruning bounds.  The box of direction n has volume height(n) x min-area rectangle(n), and the rectangle is at least as
// large as the projection of the hull on the plane normal to n, whose area is 1/2 sum_f A_f |n . n_f| (Cauchy).  The bound is
// cheap and regular (one dot product per (direction, vertex) and (direction, face)); a direction whose bound exceeds an exact
// candidate volume by more than rounding (1e-9 relative, the sums carry ~1e-13) cannot be the minimum and is never evaluated.
// Which directions survive never changes the minimum k_obb_pick takes: an equal or smaller volume is never dropped.

// 2 x area of every hull face from the edge records: the directed edges (a -> b) of face f sum a x b to 2 A_f n_f
__global__ void __launch_bounds__(256)
k_obb_face_area2(const double* __restrict__ hv, const double* __restrict__ normals, const int* __restrict__ edges, const int* __restrict__ ne_,
                 double* __restrict__ area2 /*[B][hc.f], zero*/, const HullCap hc) {
  const int b = blockIdx.y, e = blockIdx.x * 256 + threadIdx.x;
  if (e >= ne_[b]) return;
  const double* P = hv + (size_t)b * hc.v * 3;
  const double* NN = normals + (size_t)b * hc.f * 3;
  const int4 ed = *(const int4*)(edges + ((size_t)b * hc.e + e) * 4);
  const double a[3] = {P[3 * ed.x] - P[0], P[3 * ed.x + 1] - P[1], P[3 * ed.x + 2] - P[2]};
  const double q[3] = {P[3 * ed.y] - P[0], P[3 * ed.y + 1] - P[1], P[3 * ed.y + 2] - P[2]};
  const double cr[3] = {a[1] * q[2] - a[2] * q[1], a[2] * q[0] - a[0] * q[2], a[0] * q[1] - a[1] * q[0]};
  atomicAdd(&area2[(size_t)b * hc.f + ed.z], dot3(cr, NN + 3 * (size_t)ed.z));
  atomicAdd(&area2[(size_t)b * hc.f + ed.w], -dot3(cr, NN + 3 * (size_t)ed.w));
}

// 128 directions per workgroup, two per lane (a record read serves both); the SH_OBB_BND_SPLIT waves share the vertices and
// faces (a contiguous slice each, combined in LDS in wave order), read through wave-uniform addresses: scalar loads, eight
// records per trip so that a trip pays the scalar-cache latency once; many short waves hide the rest.  Also resets the
// directions' candidate records ("not evaluated") and folds the smallest bound of the humerus.
#define SH_OBB_BND_SPLIT 4
#define SH_OBB_BND_DIRS 128
#define SH_OBB_BND_THREADS (64 * SH_OBB_BND_SPLIT)
__global__ void __launch_bounds__(SH_OBB_BND_THREADS)
k_obb_bounds(const double* __restrict__ hv, const int* __restrict__ nv_, const double* __restrict__ normals, const int* __restrict__ nf_,
             const double* __restrict__ area2, double* __restrict__ lb_out /*[B][hc.f]*/, unsigned long long* __restrict__ lbmin_enc /*[B], ~0*/,
             double* __restrict__ cand_vol, int* __restrict__ cand_edge, int ntiles, int B, const HullCap hc) {
  constexpr int S = SH_OBB_BND_SPLIT;
  __shared__ double s_mn[S][2][64], s_mx[S][2][64], s_s[S][2][64];
  // XCD-aware order as in k_obb_candidates: linear id L -> humerus 8 * chunk + L % 8, so all tiles of a humerus read its hull
  // record through ONE XCD's L2 (round 3's counters: 128 MB from the fabric for 7.7 MB of records -- every XCD fetched every record)
  const int Lid = blockIdx.x, chunk = Lid / (8 * ntiles), rr = Lid - chunk * 8 * ntiles;
  const int b = chunk * 8 + (rr & 7), tile = rr >> 3;
  if (b >= B) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nv = nv_[b], nf = nf_[b];
  if (tile * SH_OBB_BND_DIRS >= nf) return;
  const double* P = hv + (size_t)b * hc.v * 3;
  const double* NN = normals + (size_t)b * hc.f * 3;
  const double* A2 = area2 + (size_t)b * hc.f;
  int jd[2];
  double n[2][3];
#pragma unroll
  for (int d = 0; d < 2; ++d) {
    jd[d] = tile * SH_OBB_BND_DIRS + d * 64 + lane;
    const int jc = jd[d] < nf ? jd[d] : nf - 1;
    n[d][0] = NN[3 * jc]; n[d][1] = NN[3 * jc + 1]; n[d][2] = NN[3 * jc + 2];
  }
  const int vper = ((nv + S - 1) / S + 7) & ~7, fper = ((nf + S - 1) / S + 7) & ~7;      // slices start on multiples of 8 records (64-byte aligned runs)
  double mn[2] = {1e300, 1e300}, mx[2] = {-1e300, -1e300}, s[2] = {0.0, 0.0};
  auto vert = [&](double x, double y, double z) {
#pragma unroll
    for (int d = 0; d < 2; ++d) { const double h = fma(n[d][0], x, fma(n[d][1], y, n[d][2] * z)); mn[d] = fmin(mn[d], h); mx[d] = fmax(mx[d], h); }
  };
  auto face = [&](double x, double y, double z, double a) {
#pragma unroll
    for (int d = 0; d < 2; ++d) s[d] = fma(fabs(fma(n[d][0], x, fma(n[d][1], y, n[d][2] * z))), fabs(a), s[d]);
  };
  {
    int i0 = wave * vper;
    const int i1 = min(nv, i0 + vper);
    for (; i0 + 8 <= i1; i0 += 8) {
      double p[24];
#pragma unroll
      for (int u = 0; u < 24; ++u) p[u] = P[3 * i0 + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) vert(p[3 * u], p[3 * u + 1], p[3 * u + 2]);
    }
    for (; i0 < i1; ++i0) vert(P[3 * i0], P[3 * i0 + 1], P[3 * i0 + 2]);
  }
  {
    int f0 = wave * fper;
    const int f1 = min(nf, f0 + fper);
    for (; f0 + 8 <= f1; f0 += 8) {
      double q[24], a[8];
#pragma unroll
      for (int u = 0; u < 24; ++u) q[u] = NN[3 * f0 + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) a[u] = A2[f0 + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) face(q[3 * u], q[3 * u + 1], q[3 * u + 2], a[u]);
    }
    for (; f0 < f1; ++f0) face(NN[3 * f0], NN[3 * f0 + 1], NN[3 * f0 + 2], A2[f0]);
  }
#pragma unroll
  for (int d = 0; d < 2; ++d) { s_mn[wave][d][lane] = mn[d]; s_mx[wave][d][lane] = mx[d]; s_s[wave][d][lane] = s[d]; }
  __syncthreads();
  if (wave >= 2) return;
  const int d = wave, j = tile * SH_OBB_BND_DIRS + d * 64 + lane;      // waves 0 and 1 finish one direction set each
  double fmn = s_mn[0][d][lane], fmx = s_mx[0][d][lane], fs = s_s[0][d][lane];
#pragma unroll
  for (int w = 1; w < S; ++w) { fmn = fmin(fmn, s_mn[w][d][lane]); fmx = fmax(fmx, s_mx[w][d][lane]); fs += s_s[w][d][lane]; }
  const double lb = 0.25 * fs * (fmx - fmn);
  unsigned long long enc = ~0ull;
  if (j < nf) {
    lb_out[(size_t)b * hc.f + j] = lb;
    cand_vol[(size_t)b * hc.f + j] = 1e300;
    cand_edge[(size_t)b * hc.f + j] = 0x7fffffff;
    if (lb >= 0.0) enc = (unsigned long long)__double_as_longlong(lb);      // (a NaN bound takes no part; its direction survives below)
  }
  for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_down(enc, off); enc = o < enc ? o : enc; }
  if (lane == 0 && enc != ~0ull) atomicMin(&lbmin_enc[b], enc);
}

// one workgroup per humerus: the directions of the next k_obb_candidates pass, in face order.
//   pass 0 (seed)      the first 16 directions whose bound is within 5 % of the humerus's smallest bound (never empty): their
//                      exact volumes give the first best_enc
//   pass 1 (survivors) every other direction whose bound does not exceed best_enc
//   pass 2             every direction not seeded, whatever its bound (SHOULDER_OBB_PRUNE=0, the A/B of the pruning: seed
//                      tile, then all the rest)
__global__ void __launch_bounds__(256)
k_obb_select(const double* __restrict__ lb_, const int* __restrict__ nf_, const unsigned long long* __restrict__ lbmin_enc,
             const unsigned long long* __restrict__ best_enc, int pass, int* __restrict__ dir_list, int* __restrict__ dir_count,
             unsigned char* __restrict__ seeded /*[B][hc.f]*/, const HullCap hc) {
  __shared__ int s_w[4];
  __shared__ int s_base;
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nf = nf_[b];
  const double* lb = lb_ + (size_t)b * hc.f;
  int* out = dir_list + (size_t)b * hc.f;
  unsigned char* sd = seeded + (size_t)b * hc.f;
  double thr;
  if (pass == 0) {
    const unsigned long long m = lbmin_enc[b];
    thr = m == ~0ull ? 1e300 : __longlong_as_double((long long)m) * 1.05;
  } else if (pass == 1) {
    const unsigned long long ub = best_enc[b];
    thr = ub == ~0ull ? 1e300 : __longlong_as_double((long long)ub);
  } else thr = 1e300;
  if (tid == 0) s_base = 0;
  __syncthreads();
  for (int j0 = 0; j0 < nf; j0 += 256) {
    const int j = j0 + tid;
    bool keep = false;
    if (j < nf) {
      const double v = lb[j];
      if (pass == 0) keep = !(v > thr);
      else keep = !sd[j] && !(v * (1.0 - 1e-9) > thr);
    }
    const unsigned long long m = __ballot(keep);
    if (lane == 0) s_w[wave] = __popcll(m);
    __syncthreads();
    int pos = s_base + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) pos += s_w[w];
    if (pass == 0) {
      if (keep && pos >= SH_OBB_TILE) keep = false;      // the seed pass is one tile
      if (j < nf) sd[j] = keep ? 1 : 0;
    }
    if (keep) out[pos] = j;
    __syncthreads();
    if (tid == 0) s_base += s_w[0] + s_w[1] + s_w[2] + s_w[3];
    __syncthreads();
  }
  if (tid == 0) dir_count[b] = pass == 0 ? min(s_base, SH_OBB_TILE) : s_base;
}

// One workgroup per (tile of 16 hull faces, humerus).  The hull record (vertices, normals, edges: ~160 KB) is read
// from L2 once per tile instead of once per face -- with one face per workgroup the kernel was bound by those
// re-reads (11 TB/s of L2 traffic), not by arithmetic:
//   heights   every lane keeps min/max of n_j . p over its vertices for the 16 directions (registers)
//   fronts    16-bit mask per hull face: which of the 16 directions see it from the front
//   edges     mask[f] ^ mask[g] = directions for which the edge is on the silhouette -> per-direction lists
//   scans     four directions at a time: project the silhouette start vertices (LDS), then every silhouette
//             edge takes the extents of all of them (ns x ns, fp64) -> min-area rectangle
// Two capacity tiers (the hull record in HBM is sized for the large one, hc.v / hc.f / hc.e; only this kernel keeps per-face
// and per-silhouette state in LDS):  <16, 4, 512, 8192, unsigned short>  hulls of up to 8 192 faces -- every fixture (2 732) and
// everything the device hull produces (3 072 slots) -- two workgroups per CU;  <8, 1, 2048, 32768, unsigned> the hull of a dense
// mesh (a 519 k-triangle humerus: 8 414 faces), one workgroup per CU, chosen by the host when a hull of the launch needs it.
// A third tier without limits, <8, 1, 0, 0, unsigned, true>: face masks, silhouette lists and projected start points in a workspace
// in global memory (ObbWs: one slice per workgroup; a grid of ws.nwg workgroups walks the tiles), for hulls above 32 768 faces and
// for directions whose silhouette has more edges than the LDS tiers hold (a run that meets one records the count in ws.need and
// sh_collect runs the batch again on the tier that holds it).  Same arithmetic, same (area, edge) order: same candidate records.
struct ObbWs {
  unsigned char* fmask;            // [nwg][hc.f]
  unsigned* lists;                 // [nwg][8][silcap]
  double2* sxy;                    // [nwg][silcap]
  double* area;                    // [nwg][silcap]
  int silcap, nwg;
  unsigned long long* need;        // [2]: the largest silhouette a direction had, in edges; 1 when a hull had more faces than the tier's masks
};
template <int T, int G, int SIL, int HFCAP, typename LT, bool GLOB>
__device__ __forceinline__ void
obb_candidates_tile(const int Lid, const double* __restrict__ hv, const int* __restrict__ nv_, const double* __restrict__ normals, const int* __restrict__ nf_,
                 const int* __restrict__ edges, const int* __restrict__ ne_, double* __restrict__ cand_vol, int* __restrict__ cand_edge,
                 int* __restrict__ err, unsigned long long* __restrict__ best_enc /*[B]: bits of the smallest candidate volume so far, ~0 = none*/,
                 const int* __restrict__ dir_list /*[B][hc.f]: the directions (hull faces) to evaluate*/, const int* __restrict__ dir_count /*[B]*/,
                 int ntiles, int B, int skip_on /*0: no in-kernel skip either (A/B of the pruning)*/, const HullCap hc, const ObbWs ws) {
  constexpr int NW = SH_OBB_THREADS / 64;
  constexpr int FB = (int)sizeof(LT) * 8 - 1;      // flag bit of a list entry: the edge's first face is the front face
  typedef typename std::conditional<(T > 8), unsigned short, unsigned char>::type MT;
  __shared__ double tn[T][3], tu[T][3], tv[T][3];
  __shared__ double red[NW][2 * T];
  __shared__ double hlo[T], hhi[T];
  __shared__ MT fmask_s[GLOB ? 1 : HFCAP];
  __shared__ LT lists_s[GLOB ? 1 : T * SIL];      // [T][silcap]: edge id | (first face is the front face) << FB
  __shared__ int cnt[T];
  __shared__ double2 sxy_s[GLOB ? 1 : G * SIL];   // [G][silcap]
  MT* const fmask = GLOB ? (MT*)(ws.fmask + (size_t)blockIdx.x * hc.f) : fmask_s;
  LT* const lists = GLOB ? (LT*)(ws.lists + (size_t)blockIdx.x * T * ws.silcap) : lists_s;
  double2* const sxy = GLOB ? ws.sxy + (size_t)blockIdx.x * G * ws.silcap : sxy_s;
  const int silcap = GLOB ? ws.silcap : SIL;
#define LIX(row, col) (GLOB ? (size_t)(row) * (size_t)silcap + (size_t)(col) : (size_t)((row) * SIL + (col)))
  __shared__ unsigned long long g_area[G];
  __shared__ int g_edge[G];
  __shared__ double g_hull2[NW][G];  // per wave: twice the signed area of the projected hull (shoelace over the directed silhouette edges)
  __shared__ int g_skip[G];
  __shared__ int s_dir[T];
  // XCD-aware order: consecutive workgroups go to the 8 XCDs in turn, each with its own L2.  Linear id L -> humerus
  // 8 * chunk + L % 8, so every tile of a humerus runs on one XCD and its hull record (160 KB, re-read by all 171 tiles)
  // stays in that XCD's L2: 8 records at a time per XCD instead of all B of them.
  const int chunk = Lid / (8 * ntiles), rr = Lid - chunk * 8 * ntiles;
  const int b = chunk * 8 + (rr & 7);
  if (b >= B) return;
  const int f0 = (rr >> 3) * T, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nv = nv_[b], nf = nf_[b], ne = ne_[b];
  if (!GLOB && nf > HFCAP) { if (tid == 0) { atomicExch(&err[b], SH_ERR_CAPACITY_DEV); atomicMax(ws.need + 1, 1ull); } return; }
  const int ndir = dir_count[b];
  if (f0 >= ndir) return;
  const int nt = min(T, ndir - f0);                    // directions in this tile: entries f0 .. f0 + nt - 1 of the humerus's list
  const double* P = hv + (size_t)b * hc.v * 3;
  const double* NN = normals + (size_t)b * hc.f * 3;
  const int* E = edges + (size_t)b * hc.e * 4;
  if (tid < T) {
    const int dir = dir_list[(size_t)b * hc.f + f0 + (tid < nt ? tid : 0)];
    s_dir[tid] = dir;
    const double* N = NN + 3 * (size_t)dir;
    double n[3] = {N[0], N[1], N[2]}, u[3], v[3];
    obb_basis(n, u, v);
    for (int k = 0; k < 3; ++k) { tn[tid][k] = n[k]; tu[tid][k] = u[k]; tv[tid][k] = v[k]; }
    cnt[tid] = 0;
  }
  __syncthreads();
  // ---- heights
  {
    double mn[T], mx[T];
#pragma unroll
    for (int j = 0; j < T; ++j) { mn[j] = 1e300; mx[j] = -1e300; }
    // (the three record sweeps below are bound by the latency of their global loads at two workgroups per CU: every
    //  sweep keeps the loads of several iterations in flight)
    for (int i0 = tid; i0 < nv; i0 += 4 * SH_OBB_THREADS) {
      double p[4][3];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = i0 + u * SH_OBB_THREADS, ic = i < nv ? i : i0;
        p[u][0] = P[3 * ic]; p[u][1] = P[3 * ic + 1]; p[u][2] = P[3 * ic + 2];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int j = 0; j < T; ++j) { double h = dot3(p[u], tn[j]); mn[j] = fmin(mn[j], h); mx[j] = fmax(mx[j], h); }      // (a repeated vertex changes no min / max)
    }
#pragma unroll
    for (int j = 0; j < T; ++j) {
      for (int off = 32; off > 0; off >>= 1) { mn[j] = fmin(mn[j], __shfl_down(mn[j], off)); mx[j] = fmax(mx[j], __shfl_down(mx[j], off)); }
      if (lane == 0) { red[wave][j] = mn[j]; red[wave][T + j] = mx[j]; }
    }
  }
  // ---- front masks
  for (int f0_ = tid; f0_ < nf; f0_ += 4 * SH_OBB_THREADS) {
    double q[4][3];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f2 = f0_ + u * SH_OBB_THREADS, fc = f2 < nf ? f2 : f0_;
      q[u][0] = NN[3 * fc]; q[u][1] = NN[3 * fc + 1]; q[u][2] = NN[3 * fc + 2];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int f2 = f0_ + u * SH_OBB_THREADS;
      unsigned m = 0;
#pragma unroll
      for (int j = 0; j < T; ++j) m |= (dot3(q[u], tn[j]) > 0 ? 1u : 0u) << j;
      if (f2 < nf && (GLOB || f2 < HFCAP)) fmask[f2] = (MT)m;
    }
  }
  __syncthreads();
  if (tid < T) {
    double lo = red[0][tid], hi = red[0][T + tid];
    for (int w = 1; w < NW; ++w) { lo = fmin(lo, red[w][tid]); hi = fmax(hi, red[w][T + tid]); }
    hlo[tid] = lo; hhi[tid] = hi;
  }
  // ---- silhouette edges of every direction: the two incident faces see it from opposite sides
  for (int e0 = tid; e0 < ne; e0 += 4 * SH_OBB_THREADS) {
    int4 ed4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) { const int e = e0 + u * SH_OBB_THREADS; ed4[u] = *(const int4*)(E + 4 * (e < ne ? e : e0)); }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int e = e0 + u * SH_OBB_THREADS;
      if (e >= ne) break;
      const int4 ed = ed4[u];
      const unsigned m1 = fmask[ed.z];
      unsigned x = (m1 ^ fmask[ed.w]) & ((1u << nt) - 1u);
      while (x) {
        const int j = __ffs(x) - 1;
        x &= x - 1;
        const int s = atomicAdd(&cnt[j], 1);
        if (s < silcap) lists[LIX(j, s)] = (LT)((unsigned)e | (((m1 >> j) & 1u) << FB));
      }
    }
  }
  __syncthreads();
  // ---- rectangle scans, G directions at a time
  for (int g0 = 0; g0 < nt; g0 += G) {
    int ns[G], pre[G + 1];
    pre[0] = 0;
#pragma unroll
    for (int jj = 0; jj < G; ++jj) {
      const int j = g0 + jj;
      int c = j < nt ? cnt[j] : 0;
      if (c > silcap) { if (tid == 0) { atomicExch(&err[b], SH_ERR_CAPACITY_DEV); atomicMax(ws.need, (unsigned long long)c); } c = silcap; }
      ns[jj] = c; pre[jj + 1] = pre[jj] + c;
    }
    if (tid < G) { g_area[tid] = (unsigned long long)__double_as_longlong(1e300); g_edge[tid] = 0x7fffffff; g_skip[tid] = 0; }
    __syncthreads();
    double h2[G];
#pragma unroll
    for (int k = 0; k < G; ++k) h2[k] = 0.0;
    // (edge stored with its first face's winding: va -> vb; directed along the FRONT face every silhouette vertex
    //  is the start of exactly one edge, so the start vertices enumerate the projection's 2-D hull once)
    for (int it = tid; it < pre[G]; it += SH_OBB_THREADS) {
      int jj = 0;
#pragma unroll
      for (int k = 1; k < G; ++k) jj += it >= pre[k] ? 1 : 0;
      const int s = it - pre[jj], j = g0 + jj;
      const unsigned rec = lists[LIX(j, s)];
      const int e = (int)(rec & ((1u << FB) - 1u));
      const bool fwd = (rec >> FB) != 0;
      const double* p = P + 3 * (size_t)(fwd ? E[4 * e] : E[4 * e + 1]);
      const double* q = P + 3 * (size_t)(fwd ? E[4 * e + 1] : E[4 * e]);
      const double2 a = make_double2(dot3(p, tu[j]), dot3(p, tv[j]));
      sxy[LIX(jj, s)] = a;
      const double cr = a.x * dot3(q, tv[j]) - dot3(q, tu[j]) * a.y;      // start x end of the directed edge
#pragma unroll
      for (int k = 0; k < G; ++k) h2[k] += jj == k ? cr : 0.0;
    }
    // (sum in the lane, then over the wave, then over the four waves in a fixed order: a double atomicAdd from every edge onto
    //  four LDS words was half of this kernel's time)
#pragma unroll
    for (int k = 0; k < G; ++k) {
      for (int off = 32; off > 0; off >>= 1) h2[k] += __shfl_down(h2[k], off);
      if (lane == 0) g_hull2[wave][k] = h2[k];
    }
    __syncthreads();
    // Lower bound of a direction's box volume: area of the projected hull x height <= min-area rectangle x height.  A direction
    // whose bound already exceeds the smallest volume any workgroup of this humerus has found cannot win: its ns^2 scan is
    // skipped (the survivor set depends on scheduling, the minimum does not: an equal or smaller volume is never skipped).
    if (tid < G && g0 + tid < nt) {
      const unsigned long long ub = *(volatile unsigned long long*)&best_enc[b];
      double hull2 = 0.0;
      for (int w = 0; w < NW; ++w) hull2 += g_hull2[w][tid];
      const double lb = 0.5 * fabs(hull2) * (hhi[g0 + tid] - hlo[g0 + tid]);
      if (skip_on && ub != ~0ull && lb * (1.0 - 1e-9) > __longlong_as_double((long long)ub)) g_skip[tid] = 1;
#if defined(SH_ABL_OBB) && SH_ABL_OBB == 1
      g_skip[tid] = 0;      // ablation: no pruning
#elif defined(SH_ABL_OBB) && SH_ABL_OBB == 2
      g_skip[tid] = 1;      // ablation (wrong results): no rectangle scan at all -> the cost of the sweeps
#endif
    }
    __syncthreads();
    if constexpr (GLOB) {
      // (any silhouette length: the areas of a direction's edges go to the workspace, the (area, edge) minimum is resolved in a second sweep)
      double* const ar = ws.area + (size_t)blockIdx.x * G * silcap;
      const int nit = pre[G];
      for (int it = tid; it < nit; it += SH_OBB_THREADS) {
        int jj = 0;
#pragma unroll
        for (int k = 1; k < G; ++k) jj += it >= pre[k] ? 1 : 0;
        const int s = it - pre[jj], j = g0 + jj, n2 = g_skip[jj] ? 0 : ns[jj];
        const int e = (int)((unsigned)lists[LIX(j, s)] & ((1u << FB) - 1u));
        double ex = 0.0, ey = 0.0, l = 0.0, area = 1e300;
        if (n2 > 0) {
          const double* pa3 = P + 3 * (size_t)E[4 * e]; const double* pc3 = P + 3 * (size_t)E[4 * e + 1];
          ex = dot3(pc3, tu[j]) - dot3(pa3, tu[j]); ey = dot3(pc3, tv[j]) - dot3(pa3, tv[j]);
          l = sqrt(ex * ex + ey * ey);
        }
        if (l != 0.0) {
          ex /= l; ey /= l;
          double amin = 1e300, amax = -1e300, bmin = 1e300, bmax = -1e300;
          const double2* sq = sxy + LIX(jj, 0);
#pragma unroll 4
          for (int s2 = 0; s2 < n2; ++s2) {
            const double2 q = sq[s2];
            const double pa = q.x * ex + q.y * ey, pb = q.y * ex - q.x * ey;
            amin = fmin(amin, pa); amax = fmax(amax, pa); bmin = fmin(bmin, pb); bmax = fmax(bmax, pb);
          }
          area = (amax - amin) * (bmax - bmin);
          atomicMin(&g_area[jj], (unsigned long long)__double_as_longlong(area));
        }
        ar[LIX(jj, s)] = area;
      }
      __syncthreads();
      for (int it = tid; it < nit; it += SH_OBB_THREADS) {
        int jj = 0;
#pragma unroll
        for (int k = 1; k < G; ++k) jj += it >= pre[k] ? 1 : 0;
        const int s = it - pre[jj], j = g0 + jj;
        const double area = ar[LIX(jj, s)];
        if (area < 1e299 && (unsigned long long)__double_as_longlong(area) == g_area[jj])
          atomicMin(&g_edge[jj], (int)((unsigned)lists[LIX(j, s)] & ((1u << FB) - 1u)));
      }
      __syncthreads();
    } else {
    // every lane takes item tid of each pass of 256 (edge s of direction jj) and remembers (area, edge, direction);
    // lexicographic (area, edge) minimum per direction in two steps: areas are non-negative doubles, so their bit
    // patterns order like the values
    constexpr int NP = G * SIL / SH_OBB_THREADS;
    double a_[NP];
    int e_[NP], j_[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      a_[p] = 1e300; e_[p] = 0x7fffffff; j_[p] = -1;
      const int it = p * SH_OBB_THREADS + tid;
      if (it < pre[G]) {
        int jj = 0;
#pragma unroll
        for (int k = 1; k < G; ++k) jj += it >= pre[k] ? 1 : 0;
        const int s = it - pre[jj], j = g0 + jj, n2 = g_skip[jj] ? 0 : ns[jj];
        const int e = (int)((unsigned)lists[LIX(j, s)] & ((1u << FB) - 1u));
        double ex = 0.0, ey = 0.0, l = 0.0;
        if (n2 > 0) {      // (a skipped direction costs no gathers here)
          const double* pa3 = P + 3 * (size_t)E[4 * e]; const double* pc3 = P + 3 * (size_t)E[4 * e + 1];
          ex = dot3(pc3, tu[j]) - dot3(pa3, tu[j]); ey = dot3(pc3, tv[j]) - dot3(pa3, tv[j]);
          l = sqrt(ex * ex + ey * ey);
        }
        if (l != 0.0) {
          ex /= l; ey /= l;
          double amin = 1e300, amax = -1e300, bmin = 1e300, bmax = -1e300;
          const double2* sq = sxy + LIX(jj, 0);
#pragma unroll 4
          for (int s2 = 0; s2 < n2; ++s2) {
            const double2 q = sq[s2];
            const double pa = q.x * ex + q.y * ey, pb = q.y * ex - q.x * ey;
            amin = fmin(amin, pa); amax = fmax(amax, pa); bmin = fmin(bmin, pb); bmax = fmax(bmax, pb);
          }
          a_[p] = (amax - amin) * (bmax - bmin); e_[p] = e; j_[p] = jj;
          atomicMin(&g_area[jj], (unsigned long long)__double_as_longlong(a_[p]));
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int p = 0; p < NP; ++p)
      if (j_[p] >= 0 && (unsigned long long)__double_as_longlong(a_[p]) == g_area[j_[p]]) atomicMin(&g_edge[j_[p]], e_[p]);
    __syncthreads();
    }
    if (tid < G && g0 + tid < nt) {
      const int j = g0 + tid;
      const double vol = g_skip[tid] ? 1e300 : __longlong_as_double((long long)g_area[tid]) * (hhi[j] - hlo[j]);
      cand_vol[(size_t)b * hc.f + s_dir[j]] = vol;
      cand_edge[(size_t)b * hc.f + s_dir[j]] = g_edge[tid];
      if (vol < 1e299) atomicMin(&best_enc[b], (unsigned long long)__double_as_longlong(vol));
    }
    __syncthreads();
  }
#undef LIX
}

template <int T, int G, int SIL, int HFCAP, typename LT, bool GLOB = false>
__global__ void __launch_bounds__(SH_OBB_THREADS)
k_obb_candidates(const double* __restrict__ hv, const int* __restrict__ nv_, const double* __restrict__ normals, const int* __restrict__ nf_,
                 const int* __restrict__ edges, const int* __restrict__ ne_, double* __restrict__ cand_vol, int* __restrict__ cand_edge,
                 int* __restrict__ err, unsigned long long* __restrict__ best_enc, const int* __restrict__ dir_list, const int* __restrict__ dir_count,
                 int ntiles, int B, int skip_on, const HullCap hc, const ObbWs ws) {
  if constexpr (GLOB) {      // a grid of ws.nwg workgroups walks the tiles, every workgroup with its slice of the workspace
    const int ntot = ntiles * ((B + 7) / 8) * 8;
    for (int Lid = blockIdx.x; Lid < ntot; Lid += gridDim.x) {
      obb_candidates_tile<T, G, SIL, HFCAP, LT, GLOB>(Lid, hv, nv_, normals, nf_, edges, ne_, cand_vol, cand_edge, err, best_enc, dir_list, dir_count, ntiles, B, skip_on, hc, ws);
      __syncthreads();      // (the tile's shared state is done with)
    }
  } else {
    obb_candidates_tile<T, G, SIL, HFCAP, LT, GLOB>(blockIdx.x, hv, nv_, normals, nf_, edges, ne_, cand_vol, cand_edge, err, best_enc, dir_list, dir_count, ntiles, B, skip_on, hc, ws);
  }
}

// one workgroup (256) per humerus
__global__ void __launch_bounds__(256)
k_obb_pick(const double* __restrict__ hv, const double* __restrict__ normals, const int* __restrict__ nf_, const int* __restrict__ edges,
           const double* __restrict__ cand_vol, const int* __restrict__ cand_edge, const float* __restrict__ verts,
           const long long* __restrict__ voff, double* __restrict__ T_pre, double* __restrict__ zb_pre, int* __restrict__ err, const HullCap hc) {
  __shared__ double wv[4];
  __shared__ int wi[4];
  __shared__ double R[9];
  __shared__ double mm[6 * 4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nf = nf_[b];
  double best = 1e300;
  int bf = 0x7fffffff;
  for (int f = tid; f < nf; f += 256) {
    double v = cand_vol[(size_t)b * hc.f + f];
    if (v < best || (v == best && f < bf)) { best = v; bf = f; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    double ob = __shfl_down(best, off);
    int of = __shfl_down(bf, off);
    if (ob < best || (ob == best && of < bf)) { best = ob; bf = of; }
  }
  if ((tid & 63) == 0) { wv[tid >> 6] = best; wi[tid >> 6] = bf; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) if (wv[w] < best || (wv[w] == best && wi[w] < bf)) { best = wv[w]; bf = wi[w]; }
    if (bf == 0x7fffffff || !(best < 1e299)) { atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV); bf = 0; }
    const double* N = normals + ((size_t)b * hc.f + bf) * 3;
    double n[3] = {N[0], N[1], N[2]}, u[3], v[3];
    obb_basis(n, u, v);
    int e = cand_edge[(size_t)b * hc.f + bf];
    const int* E = edges + ((size_t)b * hc.e + (e < 0 || e >= hc.e ? 0 : e)) * 4;
    const double* pa = hv + ((size_t)b * hc.v + min(max(E[0], 0), hc.v - 1)) * 3;      // (clamped: a void record must not turn into a wild read)
    const double* pc = hv + ((size_t)b * hc.v + min(max(E[1], 0), hc.v - 1)) * 3;
    double d[3] = {pc[0] - pa[0], pc[1] - pa[1], pc[2] - pa[2]};
    double ex = dot3(d, u), ey = dot3(d, v);
    double l = sqrt(ex * ex + ey * ey);
    ex /= l; ey /= l;
    for (int k = 0; k < 3; ++k) { R[k] = n[k]; R[3 + k] = ex * u[k] + ey * v[k]; }
    cross3(R, R + 3, R + 6);
  }
  __syncthreads();
  // extents of all mesh vertices along the three candidate axes
  const float* V = verts + 3 * voff[b];
  const long long nvert = voff[b + 1] - voff[b];
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (long long i = tid; i < nvert; i += 256) {
    double p[3] = {(double)V[3 * i], (double)V[3 * i + 1], (double)V[3 * i + 2]};
    for (int a = 0; a < 3; ++a) { double q = dot3(R + 3 * a, p); lo[a] = fmin(lo[a], q); hi[a] = fmax(hi[a], q); }
  }
  for (int a = 0; a < 3; ++a) {
    for (int off = 32; off > 0; off >>= 1) { lo[a] = fmin(lo[a], __shfl_down(lo[a], off)); hi[a] = fmax(hi[a], __shfl_down(hi[a], off)); }
    if ((tid & 63) == 0) { mm[(2 * a) * 4 + (tid >> 6)] = lo[a]; mm[(2 * a + 1) * 4 + (tid >> 6)] = hi[a]; }
  }
  __syncthreads();
  if (tid == 0) {
    double ext[3], mid[3];
    for (int a = 0; a < 3; ++a) {
      double l0 = fmin(fmin(mm[(2 * a) * 4], mm[(2 * a) * 4 + 1]), fmin(mm[(2 * a) * 4 + 2], mm[(2 * a) * 4 + 3]));
      double h0 = fmax(fmax(mm[(2 * a + 1) * 4], mm[(2 * a + 1) * 4 + 1]), fmax(mm[(2 * a + 1) * 4 + 2], mm[(2 * a + 1) * 4 + 3]));
      ext[a] = h0 - l0; mid[a] = 0.5 * (l0 + h0);
    }
    // stable ascending order of the extents: x shortest, z longest
    int o[3] = {0, 1, 2};
    for (int i = 1; i < 3; ++i) { int t = o[i]; int j = i - 1; while (j >= 0 && ext[o[j]] > ext[t]) { o[j + 1] = o[j]; --j; } o[j + 1] = t; }
    double Rx[3], Ry[3], Rz[3], cx = mid[o[0]], cz = mid[o[2]];
    for (int k = 0; k < 3; ++k) { Rx[k] = R[3 * o[0] + k]; Ry[k] = R[3 * o[1] + k]; Rz[k] = R[3 * o[2] + k]; }
    double p0[3] = {(double)V[0], (double)V[1], (double)V[2]};
    if (dot3(Rx, p0) - cx < 0) { for (int k = 0; k < 3; ++k) Rx[k] = -Rx[k]; cx = -cx; }
    if (dot3(Rz, p0) - cz < 0) { for (int k = 0; k < 3; ++k) Rz[k] = -Rz[k]; cz = -cz; }
    double Yn[3];
    cross3(Rz, Rx, Yn);
    double s = dot3(Yn, Ry) < 0 ? -1.0 : 1.0;
    double cy = s * mid[o[1]];
    double* T = T_pre + 16 * b;
    for (int k = 0; k < 3; ++k) { T[k] = Rx[k]; T[4 + k] = Yn[k]; T[8 + k] = Rz[k]; }
    T[3] = -cx; T[7] = -cy; T[11] = -cz;
    T[12] = T[13] = T[14] = 0.0; T[15] = 1.0;
    zb_pre[2 * b] = -0.5 * ext[o[2]];
    zb_pre[2 * b + 1] = 0.5 * ext[o[2]];
  }
}

// crossing points of the two end sections (z = 0.95*zmin, 0.95*zmax in the raw box frame)
__global__ void k_obb_end_points(const float* __restrict__ verts, const int* __restrict__ faces, const long long* __restrict__ voff,
                                 const long long* __restrict__ foff, const double* __restrict__ T_pre, const double* __restrict__ zb_pre,
                                 double* __restrict__ endpts /*[B][2][cap][2]*/, int* __restrict__ endcnt /*[B][2]*/, int cap) {
  int b = blockIdx.y;
  const double* T = T_pre + 16 * b;
  const float* V = verts + 3 * voff[b];
  long long f0 = foff[b], nf = foff[b + 1] - f0;
  double zpl[2] = {0.95 * zb_pre[2 * b], 0.95 * zb_pre[2 * b + 1]};
  for (long long fi = blockIdx.x * (long long)blockDim.x + threadIdx.x; fi < nf; fi += (long long)gridDim.x * blockDim.x) {
    const int* f = faces + 3 * (f0 + fi);
    int id[3] = {f[0], f[1], f[2]};
    double X[3], Y[3], Z[3];
    for (int k = 0; k < 3; ++k) {
      double o[3];
      xform_pt(T, (double)V[3 * (size_t)id[k]], (double)V[3 * (size_t)id[k] + 1], (double)V[3 * (size_t)id[k] + 2], o);
      X[k] = o[0]; Y[k] = o[1]; Z[k] = o[2];
    }
    for (int e = 0; e < 2; ++e) {
      double d[3];
      int s[3];
      for (int j = 0; j < 3; ++j) { d[j] = Z[j] - zpl[e]; s[j] = d[j] < -SH_SECTION_TOL ? -1 : 1; }
      if (s[0] == s[1] && s[1] == s[2]) continue;
      int dn = 0;
      for (int j = 0; j < 3; ++j) if (s[j] == 1 && s[(j + 1) % 3] == -1) dn = j;
      int a = dn, c = (dn + 1) % 3;
      int l = id[a] < id[c] ? a : c, h = id[a] < id[c] ? c : a;
      double t = d[l] / (d[l] - d[h]);
      int slot = atomicAdd(&endcnt[2 * b + e], 1);
      if (slot < cap) {
        double* p = endpts + (((size_t)b * 2 + e) * cap + slot) * 2;
        p[0] = X[l] + t * (X[h] - X[l]);
        p[1] = Y[l] + t * (Y[h] - Y[l]);
      }
    }
  }
}

// circle_fit_residual (sh_scalar.h) with every sum over the points done by one wave:
// lane-strided loads + shuffle reductions; all lanes run the same Levenberg-Marquardt iteration.
__device__ inline double wave_circle_eval(const double* xy, int n, double cx, double cy, double* g, double* H) {
  const int lane = threadIdx.x & 63;
  double sr = 0, sux = 0, suy = 0;
  for (int i = lane; i < n; i += 64) {
    double dx = xy[2 * i] - cx, dy = xy[2 * i + 1] - cy;
    double r = sqrt(dx * dx + dy * dy);
    sr += r; sux += dx / r; suy += dy / r;
  }
  sr = wave_sum(sr); sux = wave_sum(sux); suy = wave_sum(suy);
  const double rm = sr / n, mux = sux / n, muy = suy / n;
  double f2 = 0, g0 = 0, g1 = 0, h0 = 0, h1 = 0, h2 = 0;
  for (int i = lane; i < n; i += 64) {
    double dx = xy[2 * i] - cx, dy = xy[2 * i + 1] - cy;
    double r = sqrt(dx * dx + dy * dy);
    double f = r - rm;
    double jx = -(dx / r - mux), jy = -(dy / r - muy);
    f2 += f * f; g0 += jx * f; g1 += jy * f; h0 += jx * jx; h1 += jx * jy; h2 += jy * jy;
  }
  g[0] = wave_sum(g0); g[1] = wave_sum(g1); H[0] = wave_sum(h0); H[1] = wave_sum(h1); H[2] = wave_sum(h2);
  return wave_sum(f2);
}

__device__ inline double wave_circle_fit_residual(const double* xy, int n) {
  const int lane = threadIdx.x & 63;
  double sx = 0, sy = 0;
  for (int i = lane; i < n; i += 64) { sx += xy[2 * i]; sy += xy[2 * i + 1]; }
  double cx = wave_sum(sx) / n, cy = wave_sum(sy) / n;
  double lam = 1e-3, g[2], H[3];
  double f2 = wave_circle_eval(xy, n, cx, cy, g, H);
  for (int it = 0; it < 200; ++it) {
    double a = H[0] * (1 + lam), b = H[1], d = H[2] * (1 + lam);
    double det = a * d - b * b;
    if (det == 0) break;
    double stx = -(d * g[0] - b * g[1]) / det, sty = -(-b * g[0] + a * g[1]) / det;
    double g2[2], H2[3];
    double f2n = wave_circle_eval(xy, n, cx + stx, cy + sty, g2, H2);
    if (f2n <= f2) {
      cx += stx; cy += sty;
      bool done = (fabs(stx) + fabs(sty)) < 1e-13 * (1.0 + fabs(cx) + fabs(cy));
      f2 = f2n; g[0] = g2[0]; g[1] = g2[1]; H[0] = H2[0]; H[1] = H2[1]; H[2] = H2[2];
      lam *= 0.2;
      if (done) break;
    } else {
      lam *= 10.0;
      if (lam > 1e12) break;
    }
  }
  return f2;
}

// one workgroup of 128 lanes per humerus: wave 0 = zmin end, wave 1 = zmax end
__global__ void __launch_bounds__(128)
k_obb_ends(const double* __restrict__ endpts, const int* __restrict__ endcnt, const double* __restrict__ T_pre,
           double* __restrict__ resid /*[B][2]*/, double* __restrict__ T_obb, int* __restrict__ flipped, int* __restrict__ err, int B, int cap,
           unsigned long long* __restrict__ need /*high-water mark of the points an end section asked for*/) {
  __shared__ double res[2];
  const int b = blockIdx.x, e = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int n = endcnt[2 * b + e];
  double r;
  if (n > cap) { if (lane == 0) { atomicExch(&err[b], SH_ERR_CAPACITY_DEV); atomicMax(need, (unsigned long long)n); } n = cap; }
  if (n < 3) { if (lane == 0) atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV); r = 1e300; }
  else r = wave_circle_fit_residual(endpts + ((size_t)b * 2 + e) * cap * 2, n);
  if (lane == 0) { res[e] = r; resid[2 * b + e] = r; }
  __syncthreads();
  if (threadIdx.x == 0) {
    // mesh.py:91-112: zmin end first; the head is at zmax only if its residual is strictly smaller
    bool flip = !(res[1] < res[0]);
    flipped[b] = flip ? 1 : 0;
    const double* Tp = T_pre + 16 * b;
    double* To = T_obb + 16 * b;
    for (int k = 0; k < 16; ++k) To[k] = Tp[k];
    if (flip) for (int k = 0; k < 4; ++k) { To[k] = -Tp[k]; To[8 + k] = -Tp[8 + k]; }     // diag(-1,1,-1,1) * T_pre
  }
}

}  // namespace sh
