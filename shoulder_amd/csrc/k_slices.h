// k_slices.h -- the slice layer on the device (reference src/shoulder/humerus/slice.py, K3/K5-K9).
//
// Data layout in HBM (per ctx, B meshes, set = full/distal/proximal with N planes):
//   verts      float32 [sumV][3]        CT vertices as in the STL
//   verts_obb  float64 [sumV][3]        T_obb * v   (mesh.py:82,117: the mutated `obb.mesh`)
//   zs, zeff   float64 [B][N]           linspace z of each plane, and z_orig + (zs - z_orig)
//   seg_count  int32   [B][N]
//   segs       Seg     [B][N][SH_MAXSEG] crossing segments (16 B: the two edge keys), slot order = atomic arrival order
//   centroids  float64 [B][N][2], areas float64 [B][N], nloops int32 [B][N]
//   ring_n     int32   [B][N]           vertices of the largest loop (open count)
//   ring       float64 [B][N][SH_MAXSEG+1][2]  largest loop, CCW, canonical start, closed
//   ixy / itr_start / itr_centered_start  float64 [B][N][2][M]   (proximal set only)
// All kernels are HBM/latency bound integer + fp64 work; no MFMA here.
#pragma once
#include "sh_scalar.h"

namespace sh {

// A crossing segment is the two mesh edges it runs between.  The start point is NOT stored (it was, as two doubles: half of the
// slice layer's 1.34 GB per step): the join computes it from the edge's two vertices and the plane height with the expression the
// section uses (seg_start_point), once per segment, with the planes of a humerus joined on ONE XCD so that its verts_obb (389 KB)
// is served by that XCD's L2.
struct __attribute__((aligned(16))) Seg {
  uint32_t s_lo, s_hi, e_lo, e_hi;  // mesh-edge keys (min vid, max vid) of the start / end crossing
};
static_assert(sizeof(Seg) == 16, "Seg must be 16 bytes");
// crossing of mesh edge (lo, hi) with the plane at height z: oracle/section.py's formula, operation for operation
__device__ inline void seg_start_point(const double* __restrict__ vb /*verts_obb of the mesh*/, uint32_t lo, uint32_t hi, double z, double* x, double* y) {
  const double* pl = vb + 3 * (size_t)lo;
  const double* ph = vb + 3 * (size_t)hi;
  const double dl = pl[2] - z, dh = ph[2] - z;
  const double t = dl / (dl - dh);
  *x = pl[0] + t * (ph[0] - pl[0]);
  *y = pl[1] + t * (ph[1] - pl[1]);
}

// order-preserving double <-> uint64 for atomic min/max
__device__ inline unsigned long long enc_f64(double v) {
  unsigned long long b = (unsigned long long)__double_as_longlong(v);
  return (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
}
__device__ inline double dec_f64(unsigned long long e) {
  unsigned long long b = (e & 0x8000000000000000ull) ? (e & 0x7FFFFFFFFFFFFFFFull) : ~e;
  return __longlong_as_double((long long)b);
}

// ---- verts_obb = T_b * v, plus z bounds (mesh.py:85 `mesh.bounds[:, -1]`) -------------------
__global__ void k_transform_verts(const float* __restrict__ verts, const long long* __restrict__ voff,
                                  const double* __restrict__ T, double* __restrict__ vobb,
                                  unsigned long long* __restrict__ zb_enc /*[B][2] min,max*/) {
  int b = blockIdx.y;
  long long v0 = voff[b], nv = voff[b + 1] - v0;
  const double* Tb = T + 16 * b;
  double zmin = 1e300, zmax = -1e300;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < nv; i += (long long)gridDim.x * blockDim.x) {
    const float* p = verts + 3 * (v0 + i);
    double o[3];
    xform_pt(Tb, (double)p[0], (double)p[1], (double)p[2], o);
    double* q = vobb + 3 * (v0 + i);
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
    zmin = fmin(zmin, o[2]); zmax = fmax(zmax, o[2]);
  }
  for (int off = 32; off > 0; off >>= 1) {
    zmin = fmin(zmin, __shfl_down(zmin, off));
    zmax = fmax(zmax, __shfl_down(zmax, off));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMin(&zb_enc[2 * b], enc_f64(zmin));
    atomicMin(&zb_enc[2 * b + 1], ~enc_f64(zmax));      // (the maximum as the minimum of the complement: both words start as all ones = one uniform fill)
  }
}

__global__ void k_decode_bounds(const unsigned long long* zb_enc, double* zb, int B) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 2 * B) zb[i] = dec_f64((i & 1) ? ~zb_enc[i] : zb_enc[i]);
}

// NumPy pairwise summation (np.add.reduce on a contiguous float64 vector)
__device__ inline double np_pairwise_sum(const double* a, int n) {
  if (n < 8) {
    double r = 0.0;
    for (int i = 0; i < n; ++i) r += a[i];
    return r;
  } else if (n <= 128) {
    double r[8];
    for (int k = 0; k < 8; ++k) r[k] = a[k];
    int i;
    for (i = 8; i < n - (n % 8); i += 8)
      for (int k = 0; k < 8; ++k) r[k] += a[i + k];
    double res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) res += a[i];
    return res;
  } else {
    int n2 = n / 2;
    n2 -= n2 % 8;
    return np_pairwise_sum(a, n2) + np_pairwise_sum(a + n2, n - n2);
  }
}

// ---- plane heights of one slice set (slice.py:16-19, :219-224, :248-253, :271-276) -----------
// kind 0: full  linspace(.99 zmax, .99 zmin, N)   kind 1: proximal linspace(.99 zmax, neck_z, N)
// kind 2: distal linspace(.99 zmin, 0, N)             kind 3: the one plane z = neck_z
// kind 4: ProxObb area scan linspace(.99 zmin, .99 zmax, N) of the raw box bounds, every plane cut at its own z
//         (`mesh.section(plane_origin=[0,0,z])` in a loop, mesh.py:157-160)
// The launch also does the set's bookkeeping, which were launches of their own (round 4): the crossing counters of its planes and
// the large-tier counter start at zero, the overflow tier's per-set words are reset (when that tier runs), and the first set behind
// k_transform_verts decodes the box's z bounds (were k_decode_bounds) for everything after it.
struct PlaneAux {
  const unsigned long long* zb_enc; double* zb_out;      // nullable: z bounds from the encoded atomics of k_transform_verts, also written to zb_out
  int* seg_count; int* nlarge;                           // [B][N] crossing counters of this set; its large-tier counter
  int* ovf_nlist; unsigned long long* ovf_ctr;           // nullable: overflow tier (k_ovf.h): list length of the set, pool counters [0] and [2]
};
// Up to two slice sets that hang on the same inputs go through the set's launches TOGETHER (round 4: full + distal behind the box
// frame, neck contour + proximal behind neck_z; they were four launch groups per step): one launch for their plane heights, one
// pass over the mesh for their sections, one grid for their joins.  Per set nothing changes.
struct ManyLoops { int* list; int* n; unsigned long long* missed; };      // planes with more than SH_MAXLOOPS loops: see slice_link_plane
struct SliceSetDev {
  int N, kind, select;
  const double* zb;                                      // [B][2] z bounds the heights are taken from (kind 4: the raw box's)
  double* zs; double* zeff;                              // [B][N]
  int* seg_count; Seg* segs;                             // [B][N], [B][N][SH_MAXSEG]
  double* centroids; double* areas; int* nloops; int* ring_n; double* ring /*nullable*/; double* areas_total /*nullable*/;
  int* nlarge; ManyLoops many; unsigned long long* ovf_missed;
  PlaneAux aux;
};
struct SliceSets { SliceSetDev s[2]; int n; const double* vobb; const long long* voff; };      // (vobb / voff: the join recomputes the crossing points)

__global__ void k_make_planes(SliceSets sets, const double* __restrict__ neck_z, int B) {
  const SliceSetDev& S = sets.s[blockIdx.y];
  const int kind = S.kind, N = S.N;
  const double* zb = S.zb;
  double* zs = S.zs; double* zeff = S.zeff;
  const PlaneAux A = S.aux;
  int b = blockIdx.x;
  if (b >= B) return;
  for (int k = threadIdx.x; k < N; k += blockDim.x) A.seg_count[(size_t)b * N + k] = 0;
  if (b == 0 && threadIdx.x == 0) {
    *A.nlarge = 0;
    if (A.ovf_nlist) { A.ovf_nlist[0] = 0; A.ovf_nlist[1] = 0 /*planes with many loops*/; if (A.ovf_ctr) { A.ovf_ctr[0] = 0ull; A.ovf_ctr[2] = 0ull; } }
  }
  if (kind == 3) {   // one plane at neck_z: `mesh.section(plane_origin=[0,0,neck_z])` (surgical_neck.py:37-39)
    if (threadIdx.x == 0) { zs[b] = neck_z[b]; zeff[b] = neck_z[b]; }
    return;
  }
  double zmin, zmax;
  if (A.zb_enc) {
    zmin = dec_f64(A.zb_enc[2 * b]); zmax = dec_f64(~A.zb_enc[2 * b + 1]);
    if (threadIdx.x == 0) { A.zb_out[2 * b] = zmin; A.zb_out[2 * b + 1] = zmax; }
  } else { zmin = zb[2 * b]; zmax = zb[2 * b + 1]; }
  double a, e;
  if (kind == 4) {
    for (int k = threadIdx.x; k < N; k += blockDim.x) { double z = linspace_at(zmin * 0.99, zmax * 0.99, N, k); zs[(size_t)b * N + k] = z; zeff[(size_t)b * N + k] = z; }
    return;
  }
  if (kind == 0) { a = 0.99 * zmax; e = 0.99 * zmin; }
  else if (kind == 1) { a = 0.99 * zmax; e = neck_z[b]; }
  else { a = 0.99 * zmin; e = 0.0; }
  double* z = zs + (size_t)b * N;
  for (int k = threadIdx.x; k < N; k += blockDim.x) z[k] = linspace_at(a, e, N, k);
  __syncthreads();
  __shared__ double z_orig;
  if (threadIdx.x == 0) z_orig = np_pairwise_sum(z, N) / (double)N;   // slice.py:18 np.mean
  __syncthreads();
  for (int k = threadIdx.x; k < N; k += blockDim.x) zeff[(size_t)b * N + k] = z_orig + (z[k] - z_orig);  // :19 + section_multiplane
}

// ---- K5: triangle-centric multi-plane section ------------------------------------------------
// One lane per triangle; each crossing (triangle, plane) appends one Seg to that plane's slot
// range.  Sign rule and crossing-point formula: oracle/section.py (canonical rules).
// Slots: returning atomics on one counter per (humerus, plane) are what bounded this kernel (~150 arrivals per counter,
// executed at the memory side).  A workgroup therefore counts its crossings per plane in LDS first (pass A), reserves
// one range per plane it touches with a single global add, and hands out the slots of that range from LDS (pass B):
// file order keeps a workgroup's 256 triangles close in z, so ~90 segments share one global atomic instead of ~5.
#define SH_EMIT_MAXN 640      // planes per set (SH_NPROX = 600 is the largest)
__global__ void __launch_bounds__(256)
k_slice_emit(const double* __restrict__ vobb, const int* __restrict__ faces,
             const long long* __restrict__ voff, const long long* __restrict__ foff, SliceSets sets) {
  __shared__ int hist[SH_EMIT_MAXN];      // the planes of set 0, then those of set 1 (host: N0 + N1 <= SH_EMIT_MAXN)
  int b = blockIdx.y;
  long long f0 = foff[b], nf = foff[b + 1] - f0;
  const double* vb = vobb + 3 * voff[b];
  const int nsets = sets.n, Ntot = sets.s[0].N + (nsets > 1 ? sets.s[1].N : 0);
  for (long long f_base = blockIdx.x * (long long)blockDim.x; f_base < nf; f_base += (long long)gridDim.x * blockDim.x) {
    for (int k = threadIdx.x; k < Ntot; k += blockDim.x) hist[k] = 0;
    const long long fi = f_base + threadIdx.x;
    const bool live = fi < nf;
    const int* f = faces + 3 * (f0 + (live ? fi : 0));
    int id[3] = {f[0], f[1], f[2]};
    double Z[3];
    for (int k = 0; k < 3; ++k) Z[k] = vb[3 * (size_t)id[k] + 2];
    double fzmin = fmin(Z[0], fmin(Z[1], Z[2])), fzmax = fmax(Z[0], fmax(Z[1], Z[2]));
    int lo_[2] = {1, 1}, hi_[2] = {0, 0};      // empty range: a lane without a triangle or without planes
    for (int si = 0; si < nsets; ++si) {
      const int N = sets.s[si].N;
      const double* zp = sets.s[si].zeff + (size_t)b * N;
      const double z_first = zp[0], z_last = zp[N - 1];
      const double inv_step = (double)(N - 1) / (z_last - z_first);
      double ka = (fzmin - z_first) * inv_step, kb = (fzmax - z_first) * inv_step;
      double klo = fmin(ka, kb), khi = fmax(ka, kb);
      int lo = 1, hi = 0;
      if (live && !(khi < -1.0 || klo > (double)N)) {
        lo = (int)floor(fmax(klo, 0.0)) - 1; hi = (int)ceil(fmin(khi, (double)(N - 1))) + 1;
        lo = lo < 0 ? 0 : lo;
        hi = hi > N - 1 ? N - 1 : hi;
      }
      if (live && N == 1) { lo = 0; hi = 0; }
      lo_[si] = lo; hi_[si] = hi;
    }
    __syncthreads();
    // pass A: crossings per plane of this workgroup's triangles
    for (int si = 0; si < nsets; ++si) {
      const int N = sets.s[si].N, hoff = si ? sets.s[0].N : 0;
      const double* zp = sets.s[si].zeff + (size_t)b * N;
      for (int k = lo_[si]; k <= hi_[si]; ++k) {
        const double z = zp[k];
        const int s0 = Z[0] - z < -SH_SECTION_TOL ? -1 : 1, s1 = Z[1] - z < -SH_SECTION_TOL ? -1 : 1, s2 = Z[2] - z < -SH_SECTION_TOL ? -1 : 1;
        if (!(s0 == s1 && s1 == s2)) atomicAdd(&hist[hoff + k], 1);
      }
    }
    __syncthreads();
    for (int k = threadIdx.x; k < Ntot; k += blockDim.x) {
      const int n = hist[k];
      if (n > 0) {      // first slot of this workgroup's range
        const int si = k >= sets.s[0].N ? 1 : 0, kk = si ? k - sets.s[0].N : k;
        hist[k] = atomicAdd(&sets.s[si].seg_count[(size_t)b * sets.s[si].N + kk], n);
      }
    }
    __syncthreads();
    // pass B: the segments, slots handed out from LDS
    for (int si = 0; si < nsets; ++si) {
      const int N = sets.s[si].N, hoff = si ? sets.s[0].N : 0;
      const double* zp = sets.s[si].zeff + (size_t)b * N;
      Seg* segs = sets.s[si].segs;
      for (int k = lo_[si]; k <= hi_[si]; ++k) {
        const double z = zp[k];
        double d[3];
        int s[3];
        for (int j = 0; j < 3; ++j) { d[j] = Z[j] - z; s[j] = d[j] < -SH_SECTION_TOL ? -1 : 1; }
        if (s[0] == s[1] && s[1] == s[2]) continue;
        const int slot = atomicAdd(&hist[hoff + k], 1);
        int up = 0, dn = 0;
        for (int j = 0; j < 3; ++j) {
          int jn = (j + 1) % 3;
          if (s[j] == -1 && s[jn] == 1) up = j;
          if (s[j] == 1 && s[jn] == -1) dn = j;
        }
        Seg sg;
        {  // start = crossing on the edge walked downwards (+ -> -)
          int a = dn, c = (dn + 1) % 3;
          int l = id[a] < id[c] ? a : c, h = id[a] < id[c] ? c : a;
          sg.s_lo = (uint32_t)id[l]; sg.s_hi = (uint32_t)id[h];      // (its point: seg_start_point, in the join)
        }
        {
          int a = up, c = (up + 1) % 3;
          int l = id[a] < id[c] ? a : c, h = id[a] < id[c] ? c : a;
          sg.e_lo = (uint32_t)id[l]; sg.e_hi = (uint32_t)id[h];
        }
        // a plane with more crossings than slots is an overflow plane: seg_count keeps counting, k_ovf.h sections it again into the pool
        if (slot < SH_MAXSEG) segs[((size_t)b * N + k) * SH_MAXSEG + slot] = sg;
      }
    }
    __syncthreads();      // hist is zeroed again at the top of the next chunk
  }
}

// ---- K5b-K7: join segments into loops, orient, areas, AABB centre, largest loop --------------
// One workgroup per (mesh, plane).  LDS hash join on edge keys -> successor list -> pointer
// jumping for (a) the minimum edge key of each loop (canonical start, B-1) and (b) the rank of
// every segment from that start.
#define SH_LINK_THREADS 256
#define SH_MAXLOOPS 32
#define SH_SMALLSEG 384      // tier boundary: planes with at most this many segments take the small-LDS instantiation (humerus sections: mean 140-210, max ~330)

__device__ inline uint32_t hash_key64(unsigned long long k) {
  k ^= k >> 33; k *= 0xff51afd7ed558ccdull; k ^= k >> 33;
  return (uint32_t)k;
}

// select: 0 = largest loop (slice.py:53-59), 1 = loop whose closed-ring vertex mean is nearest
// the origin in L1 (surgical_neck.py:40-48)
// Two instantiations share the grid: CAP = SH_SMALLSEG (17.5 KB of LDS, 8 workgroups = every wave slot of a CU) takes the planes
// with up to 384 segments -- every section of a humerus at the fixture resolution -- and CAP = SH_MAXSEG (46 KB) the rest; the
// other tier's planes exit at once.
// A plane with more than SH_MAXLOOPS closed loops (a mesh with dozens of components in one section: fragments, trabecular
// cavities) is handed to the overflow tier's join (k_ovf.h: loop tables of 1 024 entries): listed (ManyLoops), given pool ranges by
// k_ovf_plan_loops.  With that tier skipped for a resident batch (no list): `missed` tells sh_collect to run again with it.

template <int CAP>
__device__ inline void slice_link_plane(const int pl, int N, const int* __restrict__ seg_count, const Seg* __restrict__ segs,
             const double* __restrict__ vobb, const long long* __restrict__ voff, const double* __restrict__ zeff,
             double* __restrict__ centroids, double* __restrict__ areas, int* __restrict__ nloops,
             int* __restrict__ ring_n, double* __restrict__ ring /*nullable*/, int select, int* __restrict__ err,
             double* __restrict__ areas_total /*nullable: |sum of the signed loop areas| = Path2D.area*/, ManyLoops many, int* __restrict__ nlarge = nullptr) {
  // LDS per plane decides how many planes a CU joins at once (the join is a chain of short dependent steps): 44 bytes per
  // segment.  bufB holds the hash table until the label ping-pong starts; the crossing points stay in HBM (read twice, L2 hits).
  constexpr int HASH = CAP <= 384 ? 512 : 2048;
  static_assert(HASH * 4 <= CAP * 8 && HASH > CAP, "the hash table lives in bufB");
  __shared__ unsigned long long skey[CAP];
  __shared__ unsigned long long bufA[CAP];   // ekey, then label ping, then rank arrays or loop map, then ring x
  __shared__ unsigned long long bufB[CAP];   // hash table, then label pong, then rank arrays or loop map, then ring y
  __shared__ int nxt[CAP], jmpA[CAP], jmpB[CAP], offA[CAP], offB[CAP];
  int* const table = (int*)bufB;
  __shared__ unsigned long long l_key[SH_MAXLOOPS];
  __shared__ int l_start[SH_MAXLOOPS], l_len[SH_MAXLOOPS], l_off[SH_MAXLOOPS];
  __shared__ double l_area[SH_MAXLOOPS], l_sel[SH_MAXLOOPS];
  __shared__ int n_loops, bad;
  __shared__ double bbv[4];

  const int b = pl / N;                // pl = b*N + k
  const int tid = threadIdx.x;
  int cnt = seg_count[pl];
  if (CAP == SH_SMALLSEG ? cnt > SH_SMALLSEG : cnt <= SH_SMALLSEG) {      // the other tier's plane
    if (CAP == SH_SMALLSEG && nlarge && tid == 0) atomicAdd(nlarge, 1);        // (tells the large-tier sweeps that they have work)
    return;
  }
  if (cnt > SH_MAXSEG) return;      // an overflow plane: k_slice_link_huge (k_ovf.h)
  const int n = cnt;
  if (tid == 0) { n_loops = 0; bad = 0; }
  for (int i = tid; i < HASH; i += SH_LINK_THREADS) table[i] = -1;
  const Seg* sp = segs + (size_t)pl * SH_MAXSEG;
  const double* vb = vobb + 3 * voff[b];
  const double zpl = zeff[pl];
  for (int i = tid; i < n; i += SH_LINK_THREADS) {
    Seg s = sp[i];
    skey[i] = ((unsigned long long)s.s_lo << 32) | s.s_hi;
    bufA[i] = ((unsigned long long)s.e_lo << 32) | s.e_hi;
  }
  __syncthreads();
#if defined(SH_ABL_LINK) && SH_ABL_LINK == 1
  return;
#endif
  if (n < 3) {
    if (tid == 0) {
      centroids[2 * (size_t)pl] = 0; centroids[2 * (size_t)pl + 1] = 0; areas[pl] = 0; nloops[pl] = 0; ring_n[pl] = 0;
      if (areas_total) areas_total[pl] = 0;      // an empty section of the area scan is legal (plane past a ragged cut)
      else atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV);
    }
    return;
  }
  for (int i = tid; i < n; i += SH_LINK_THREADS) {
    uint32_t h = hash_key64(skey[i]) & (HASH - 1);
    while (atomicCAS(&table[h], -1, i) != -1) h = (h + 1) & (HASH - 1);
  }
  __syncthreads();
  for (int i = tid; i < n; i += SH_LINK_THREADS) {
    unsigned long long k = bufA[i];
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
#if defined(SH_ABL_LINK) && SH_ABL_LINK == 2
  return;
#endif
  // One pointer-jumping pass gives both what the walk needs (round 2; two passes before: labels, then ranks): every node carries,
  // for the stretch of 2^k successors starting at itself, the smallest start key on it and the distance to that key's node.
  // Joining a stretch with the one behind it keeps the smaller key (its own on a tie: a stretch longer than the loop meets
  // the same node again), so once the stretches cover the loop every node knows the loop's canonical start (B-1: the
  // minimum edge key) and its forward distance to it.
  unsigned long long* labA = bufA;
  unsigned long long* labB = bufB;
  int* ja = jmpA; int* jb = jmpB;
  int* ra = offA; int* rb = offB;
  for (int i = tid; i < n; i += SH_LINK_THREADS) { labA[i] = skey[i]; ja[i] = nxt[i]; ra[i] = 0; }
  __syncthreads();
  for (int span = 1; span < n; span <<= 1) {
    for (int i = tid; i < n; i += SH_LINK_THREADS) {
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
#if defined(SH_ABL_LINK) && SH_ABL_LINK == 3
  return;
#endif
  // labA[i] = start key of i's loop, ra[i] = forward steps from i to the start node (0: i is a start node)
  for (int i = tid; i < n; i += SH_LINK_THREADS)
    if (ra[i] == 0) {
      int l = atomicAdd(&n_loops, 1);
      if (l < SH_MAXLOOPS) l_start[l] = i;
    }
  __syncthreads();
  if (n_loops > SH_MAXLOOPS) {      // (uniform: every lane leaves; nothing of this plane has been written yet)
    if (tid == 0) {
      if (many.list) many.list[atomicAdd(many.n, 1)] = pl;
      else if (many.missed) atomicExch(many.missed, 1ull);
      else atomicExch(&err[b], SH_ERR_CAPACITY_DEV);
    }
    return;
  }
  const int nl = n_loops;
  if (tid == 0) {
    // canonical loop order: ascending start key
    for (int a = 1; a < nl; ++a) {
      int v = l_start[a]; int c = a - 1;
      while (c >= 0 && skey[l_start[c]] > skey[v]) { l_start[c + 1] = l_start[c]; --c; }
      l_start[c + 1] = v;
    }
    int off = 0;
    for (int l = 0; l < nl; ++l) {
      int s = l_start[l];
      int L = ra[nxt[s]] + 1;
      l_len[l] = L; l_off[l] = off; off += L;
      l_key[l] = skey[s];
    }
    if (off != n) bad = 1;        // some segments are on no closed loop
  }
  __syncthreads();
  // ring placement: position from start = (L - r) mod L
  // label buffers are dead from here on: reuse them for the ordered ring
  double* rx = (double*)bufA;
  double* ry = (double*)bufB;
  int my_pos[(CAP + SH_LINK_THREADS - 1) / SH_LINK_THREADS];
  {
    int c = 0;
    for (int i = tid; i < n; i += SH_LINK_THREADS, ++c) {
      const unsigned long long key = labA[i];
      int l = -1;
      for (int q = 0; q < nl; ++q) if (l_key[q] == key) { l = q; break; }      // (one or two loops per section)
      const int L = l >= 0 ? l_len[l] : 1;
      const int r = ra[i];
      const int pos = r == 0 ? 0 : L - r;
      my_pos[c] = l < 0 ? -1 : l_off[l] + pos;
    }
  }
  __syncthreads();
  {
    int c = 0;
    for (int i = tid; i < n; i += SH_LINK_THREADS, ++c)
      if (my_pos[c] >= 0 && my_pos[c] < n) { const Seg sg = sp[i]; seg_start_point(vb, sg.s_lo, sg.s_hi, zpl, &rx[my_pos[c]], &ry[my_pos[c]]); }
  }
  __syncthreads();
#if defined(SH_ABL_LINK) && SH_ABL_LINK == 4
  return;
#endif
  // AABB over every loop vertex (trimesh Path2D.centroid, slice.py:38) by the last wave, while the other three take the loops
  // (all four waves reducing four doubles each through the LDS crossbar, then 64-bit LDS atomics, was a quarter of the kernel).
  // The ordered ring in LDS holds every crossing point once -- min / max do not care about the order; a section with
  // segments on no closed loop is flagged above and its numbers are void anyway.
  const int lane = tid & 63, wave = tid >> 6;
  if (wave == SH_LINK_THREADS / 64 - 1) {
    double x0 = 1e300, x1 = -1e300, y0 = 1e300, y1 = -1e300;
    for (int i = lane; i < n; i += 64) {
      const double qx = rx[i], qy = ry[i];
      x0 = fmin(x0, qx); x1 = fmax(x1, qx); y0 = fmin(y0, qy); y1 = fmax(y1, qy);
    }
    for (int off = 32; off > 0; off >>= 1) {
      x0 = fmin(x0, __shfl_down(x0, off)); x1 = fmax(x1, __shfl_down(x1, off));
      y0 = fmin(y0, __shfl_down(y0, off)); y1 = fmax(y1, __shfl_down(y1, off));
    }
    if (lane == 0) { bbv[0] = x0; bbv[1] = x1; bbv[2] = y0; bbv[3] = y1; }
  } else {
    // per-loop shoelace area and the selection score: one wave per loop, lane-strided terms in ring order, fixed shuffle
    // tree (deterministic; one lane walking the ring alone cost as much as the whole join)
    for (int l = wave; l < nl; l += SH_LINK_THREADS / 64 - 1) {
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
        // surgical_neck.py:43-46: mean over the CLOSED ring (first vertex counted twice)
        mx = (mx + rx[o]) / (double)(L + 1); my = (my + ry[o]) / (double)(L + 1);
        l_sel[l] = fabs(mx) + fabs(my);
      }
    }
  }
#if defined(SH_ABL_LINK) && SH_ABL_LINK == 7
  return;
#endif
  __syncthreads();
  // the chosen loop: every lane picks it for itself (at most a handful of broadcast reads) -- one lane choosing it for all,
  // between two barriers and in front of its scalar stores, was a quarter of the kernel
  int best = 0;
  for (int l = 1; l < nl; ++l) {
    if (select == 0) { if (fabs(l_area[l]) > fabs(l_area[best])) best = l; }
    else { if (l_sel[l] < l_sel[best]) best = l; }
  }
  if (tid == 0) {
    double x0 = bbv[0], x1 = bbv[1], y0 = bbv[2], y1 = bbv[3];
    centroids[2 * (size_t)pl] = 0.5 * (x0 + x1);
    centroids[2 * (size_t)pl + 1] = 0.5 * (y0 + y1);
    int amax = 0;
    for (int l = 1; l < nl; ++l) if (fabs(l_area[l]) > fabs(l_area[amax])) amax = l;
    areas[pl] = fabs(l_area[amax]);
    if (areas_total) {      // outer loops minus holes (mesh.py:160 `slice.area`), loops in canonical order
      double tot = 0.0;
      for (int l = 0; l < nl; ++l) tot += l_area[l];
      areas_total[pl] = fabs(tot);
    }
    nloops[pl] = nl;
    ring_n[pl] = l_len[best];
    if (bad) atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV);
  }
#if defined(SH_ABL_LINK) && SH_ABL_LINK == 5
  return;
#endif
  if (ring) {
    int l = best, o = l_off[l], L = l_len[l];
    bool rev = l_area[l] < 0;           // clockwise loop: traverse backwards from the same start
    double* out = ring + (size_t)pl * (SH_MAXSEG + 1) * 2;
    for (int q = tid; q <= L; q += SH_LINK_THREADS) {
      int qq = q == L ? 0 : q;
      int src = rev ? (qq == 0 ? 0 : L - qq) : qq;
      out[2 * q] = rx[o + src];
      out[2 * q + 1] = ry[o + src];
    }
  }
}

// small tier: one workgroup per plane.  Large tier: a small grid sweeps all planes and works on the few (usually none)
// with more than SH_SMALLSEG segments -- a workgroup per plane would pay its 70 KB LDS allocation 40 000 times for nothing.
__global__ void __launch_bounds__(SH_LINK_THREADS)
k_slice_link(SliceSets sets, int B, int* __restrict__ err) {
  // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 names a group of workgroups that share an L2): with the batch a
  // multiple of 8, workgroup id joins a plane of humerus (id % 8) + 8 j, so the vertices of 1/8 of the batch -- what the crossing
  // points are computed from -- stay in one XCD's L2 (3 MB at B = 64).  Placement changes speed only.
  const int ntot = sets.s[0].N + (sets.n > 1 ? sets.s[1].N : 0);
  int id = (int)blockIdx.x;
  if (B % 8 == 0) { const int x = id & 7, j = id >> 3; id = (x + 8 * (j / ntot)) * ntot + j % ntot; }      // -> humerus-major plane id over both sets
  const int b = id / ntot, kk = id - b * ntot;
  const int si = kk >= sets.s[0].N ? 1 : 0;
  const SliceSetDev& S = sets.s[si];
  slice_link_plane<SH_SMALLSEG>(b * S.N + (kk - (si ? sets.s[0].N : 0)), S.N, S.seg_count, S.segs, sets.vobb, sets.voff, S.zeff, S.centroids, S.areas, S.nloops, S.ring_n,
                                S.ring, S.select, err, S.areas_total, S.many, S.nlarge);
}
__global__ void __launch_bounds__(SH_LINK_THREADS)
k_slice_link_large(SliceSets sets, int B, int* __restrict__ err) {
  for (int si = 0; si < sets.n; ++si) {
    const SliceSetDev& S = sets.s[si];
    if (*S.nlarge == 0) continue;      // (the usual case: the sweep below is ~75 dependent loads per workgroup for nothing)
    const int nplanes = B * S.N;
    for (int pl = blockIdx.x; pl < nplanes; pl += gridDim.x) {
      // An overflow plane belongs to k_slice_link_huge (k_ovf.h).  When the host skipped that tier for a resident batch "known" to need
      // none and the planes have moved since (another frame, other parameters), nobody joins this plane: say so, sh_collect runs the
      // batch again with the tier on instead of handing out the previous run's section.
      const int cnt = S.seg_count[pl];
      if (cnt > SH_MAXSEG && S.ovf_missed && threadIdx.x == 0) atomicExch(S.ovf_missed, 1ull);
      if (cnt <= SH_SMALLSEG || cnt > SH_MAXSEG) continue;
      slice_link_plane<SH_MAXSEG>(pl, S.N, S.seg_count, S.segs, sets.vobb, sets.voff, S.zeff, S.centroids, S.areas, S.nloops, S.ring_n, S.ring, S.select, err, S.areas_total, S.many);
      __syncthreads();
    }
  }
}

// ---- K8/K9: arclength resampling + polar images (slice.py:65-147, :166-206) -------------------
// One workgroup per (mesh, plane).  cumsum is sequential (np.cumsum order); each sample is one
// np.interp evaluation; theta = atan2(y,x), r = sqrt(x^2+y^2); rows rolled to argmin(theta).
#define SH_RS_THREADS 128      // two waves per plane: 16 planes per CU hide each other's serial stretches (ring load, running sum)
// Round 2: the samples stay in registers (four per lane; M = 512, 128 lanes) -- 9 KB of LDS per plane instead of 25 KB, so a CU
// works on eight planes at once; np.cumsum's running sum (one lane, order kept) loads eight lengths at a time instead of
// paying an LDS round trip per element (it was half of a plane's latency); x and y share one search per sample.
// Which of a plane's three products leave the kernel.  They are 24 KB per plane -- 944 MB per step at B = 64, the largest stream of
// the geometry chain, and beside another lane's UNet pass every GB of the chain's traffic costs that pass ~0.28 ms (DESIGN.md
// section 6) -- and the stages behind read a part of them only: the polar rows about the origin (itr_start) from plane SH_ANP_ROW0
// on (k_anp_rows, k_sphere_partial), the centred ones (itr_cs) inside the groove's cut-off range (k_groove_rows, k_groove_tail),
// the resampled contour (ixy) never.  keep_all (sh_set_keep_products: the slice layer's parity tests fetch every plane): all of it.
struct RsWant { int keep_all, st_lo, cs_lo, cs_hi; };
template <int CAP>
__device__ inline void resample_polar_plane(const int pl, int N, int M, const int* __restrict__ ring_n, const double* __restrict__ ring,
                 const double* __restrict__ centroids, double* __restrict__ ixy,
                 double* __restrict__ itr_start, double* __restrict__ itr_cs, const long long* __restrict__ ovf_roff, const RsWant W) {
  static_assert(SH_MPROX % SH_RS_THREADS == 0, "whole samples per lane");
  constexpr int NS = SH_MPROX / SH_RS_THREADS;      // samples per lane
  __shared__ double rx[CAP + 1], ry[CAP + 1], d[CAP + 1];
  __shared__ int amin_idx;
  __shared__ double wmin[SH_RS_THREADS / 64];
  __shared__ int widx[SH_RS_THREADS / 64];
  const int tid = threadIdx.x;
  const int L = ring_n[pl];
  if (CAP == SH_SMALLSEG ? L > SH_SMALLSEG : L <= SH_SMALLSEG) return;      // the other tier's plane
  if (ovf_roff[pl] >= 0) return;                                           // ring in the overflow pool: k_resample_polar_huge (k_ovf.h)
  const int kpl = pl % N;
  const bool want_st = W.keep_all || kpl >= W.st_lo, want_cs = W.keep_all || (kpl >= W.cs_lo && kpl < W.cs_hi);
  if (!want_st && !want_cs) return;                                        // (the same for every lane of the workgroup)
  const double* rp = ring + (size_t)pl * (SH_MAXSEG + 1) * 2;
  for (int q = tid; q <= L; q += SH_RS_THREADS) { rx[q] = rp[2 * q]; ry[q] = rp[2 * q + 1]; }
  __syncthreads();
  // segment lengths by all lanes, then the running sum in np.cumsum's order by one
  for (int q = 1 + tid; q <= L; q += SH_RS_THREADS) {
    double dx = rx[q] - rx[q - 1], dy = ry[q] - ry[q - 1];
    d[q] = sqrt(dx * dx + dy * dy);
  }
  __syncthreads();
  if (tid == 0) {
    double acc = 0.0;
    d[0] = 0.0;
    int q = 1;
    for (; q + 8 <= L + 1; q += 8) {
      double v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = d[q + u];
#pragma unroll
      for (int u = 0; u < 8; ++u) { acc += v[u]; v[u] = acc; }
#pragma unroll
      for (int u = 0; u < 8; ++u) d[q + u] = v[u];
    }
    for (; q <= L; ++q) { acc += d[q]; d[q] = acc; }
  }
  __syncthreads();
#if defined(SH_ABL_RS) && SH_ABL_RS == 2
  return;
#endif
  const double dmax = d[L];
  double sx[NS], sy[NS];
#pragma unroll
  for (int u = 0; u < NS; ++u) {
    const int j = tid + u * SH_RS_THREADS;
    const double t = linspace_at(0.0, dmax, M, j);
    // interp1 (sh_common.h) for x and y with one search: same comparisons, same arithmetic per coordinate
    const int n = L + 1;
    if (t < d[0]) { sx[u] = rx[0]; sy[u] = ry[0]; }
    else if (!(t < d[n - 1])) { sx[u] = rx[n - 1]; sy[u] = ry[n - 1]; }
    else {
      int lo = 0, hi = n - 1;
      while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (t >= d[mid]) lo = mid; else hi = mid; }
      const double x0 = d[lo], fx = rx[lo], fy = ry[lo];
      if (x0 == t) { sx[u] = fx; sy[u] = fy; }
      else {
        const double den = d[lo + 1] - x0;
        sx[u] = (rx[lo + 1] - fx) / den * (t - x0) + fx;
        sy[u] = (ry[lo + 1] - fy) / den * (t - x0) + fy;
      }
    }
  }
  if (W.keep_all) {
    double* oxy = ixy + (size_t)pl * 2 * M;
#pragma unroll
    for (int u = 0; u < NS; ++u) { const int j = tid + u * SH_RS_THREADS; oxy[j] = sx[u]; oxy[M + j] = sy[u]; }
  }
#if defined(SH_ABL_RS) && SH_ABL_RS == 3
  return;
#endif
  const double cx = centroids[2 * (size_t)pl], cy = centroids[2 * (size_t)pl + 1];
  for (int pass = 0; pass < 2; ++pass) {
#if defined(SH_ABL_RS) && SH_ABL_RS == 4
    if (pass == 1) return;
#endif
    if (pass == 0 ? !want_st : !want_cs) continue;
    const double ox = pass ? cx : 0.0, oy = pass ? cy : 0.0;
    double best = 1e300;
    int bi = 0x7fffffff;
    double th[NS], rr[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int j = tid + u * SH_RS_THREADS;
      const double x = sx[u] - ox, y = sy[u] - oy;
#if defined(SH_ABL_RS) && SH_ABL_RS == 1
      th[u] = y + x;      // ablation (wrong results): what the kernel costs without its atan2 / sqrt
      rr[u] = x * x + y * y;
#else
      th[u] = atan2(y, x);
      rr[u] = sqrt(x * x + y * y);
#endif
      if (th[u] < best || (th[u] == best && j < bi)) { best = th[u]; bi = j; }
    }
    for (int off = 32; off > 0; off >>= 1) {
      double ob = __shfl_down(best, off);
      int oi = __shfl_down(bi, off);
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
    const int k0 = amin_idx;      // row rolled so that sample k0 comes first: out[j] = in[(j + k0) % M]
    double* o = (pass ? itr_cs : itr_start) + (size_t)pl * 2 * M;
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      int dst = tid + u * SH_RS_THREADS - k0; if (dst < 0) dst += M;
      o[dst] = th[u];
      o[M + dst] = rr[u];
    }
    __syncthreads();      // wmin / widx / amin_idx are reused by the next pass
  }
}

__global__ void __launch_bounds__(SH_RS_THREADS)
k_resample_polar(int N, int M, const int* __restrict__ ring_n, const double* __restrict__ ring, const double* __restrict__ centroids,
                 double* __restrict__ ixy, double* __restrict__ itr_start, double* __restrict__ itr_cs, const long long* __restrict__ ovf_roff, const RsWant W) {
  resample_polar_plane<SH_SMALLSEG>(blockIdx.x, N, M, ring_n, ring, centroids, ixy, itr_start, itr_cs, ovf_roff, W);
}
__global__ void __launch_bounds__(SH_RS_THREADS)
k_resample_polar_large(int nplanes, int N, int M, const int* __restrict__ ring_n, const double* __restrict__ ring, const double* __restrict__ centroids,
                       double* __restrict__ ixy, double* __restrict__ itr_start, double* __restrict__ itr_cs, const int* __restrict__ nlarge,
                       const long long* __restrict__ ovf_roff, const RsWant W) {
  if (*nlarge == 0) return;      // no plane with more than SH_SMALLSEG segments, hence no ring that long
  for (int pl = blockIdx.x; pl < nplanes; pl += gridDim.x) {
    if (ring_n[pl] <= SH_SMALLSEG || ovf_roff[pl] >= 0) continue;
    resample_polar_plane<SH_MAXSEG>(pl, N, M, ring_n, ring, centroids, ixy, itr_start, itr_cs, ovf_roff, W);
    __syncthreads();
  }
}

}  // namespace sh
