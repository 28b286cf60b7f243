// k_te.h -- trans-epicondylar axis, coordinate system and the landmark record
// (reference src/shoulder/humerus/epicondyle.py:29-101, src/shoulder/bone.py:146-157).
//   k_te_rows   :33-40  min-area rectangle of every distal slice in the cut (one lane per slice)
//   k_te_ends   :39-89  widest slice -> end slivers -> centroids -> farthest pair (needs the distal set only: runs beside the UNet)
//   k_te_orient :90-96  medial end first (needs the head's central axis, i.e. the anatomic neck)
//   k_pack      bone.py:146-157 construct_csys(canal, TE) + fill sh_landmarks (CT coordinates)
#pragma once
#include "../../include/shoulder_hip.h"
#include "k_unet.h"
#include "k_ovf.h"

namespace sh {

#define SH_TE_ROW0 2
#define SH_TE_NROWS 37          // int((1-.99)*200)=2 .. int((1-.8)*200)=39 (slice.py:157-164)

// Convex hull of a ring by one wave: gift wrapping with a 64-lane tournament per step.  Melkman's deque walk (one lane,
// ~20 dependent LDS round trips per vertex: ~300 us for a 300-point ring -- the whole kernel was one slice's walk long) gives the
// strictly convex vertices in counter-clockwise order starting at the hull vertex with the highest ring index; so does this:
// start at the lowest point (lowest y, then x: extreme, hence on the hull), from the current vertex p take the point q with every
// other point on the left of p -> q (of collinear candidates the farthest: collinear points are dropped, as Melkman's `<= 0`
// pops do), until the start comes round again; then rotate.  The two disagree only where an orientation determinant is
// within rounding of zero; the minimum-area rectangle does not notice (extents move by ~1e-13).
// xy: n points in LDS; hull: LDS, >= n ints.  Returns the hull size (every lane); hull[(k + *rot) % nh], k = 0.., is Melkman's list.
__device__ inline int wave_hull_wrap(const double* xy, int n, int* hull, int lane, int* rot) {
  // lowest point
  double by = 1e300, bx = 1e300; int bi = 0x7fffffff;
  for (int i = lane; i < n; i += 64) {
    const double x = xy[2 * i], y = xy[2 * i + 1];
    if (y < by || (y == by && (x < bx || (x == bx && i < bi)))) { by = y; bx = x; bi = i; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    const double oy = __shfl_xor(by, off), ox = __shfl_xor(bx, off); const int oi = __shfl_xor(bi, off);
    if (oy < by || (oy == by && (ox < bx || (ox == bx && oi < bi)))) { by = oy; bx = ox; bi = oi; }
  }
  const int p0 = bi;
  int p = p0, nh = 0;
  double px = bx, py = by;
  for (;;) {
    if (lane == 0) hull[nh] = p;
    ++nh;
    if (nh > n) break;                               // (cannot happen on a ring of distinct points; keeps a degenerate input bounded)
    // this lane's best candidate among its points
    int q = -1; double qx = 0.0, qy = 0.0, qd = 0.0;
    for (int i = lane; i < n; i += 64) {
      const double x = xy[2 * i], y = xy[2 * i + 1];
      const double dx = x - px, dy = y - py, d = dx * dx + dy * dy;
      if (d == 0.0) continue;                        // p itself
      if (q < 0) { q = i; qx = x; qy = y; qd = d; continue; }
      const double o = (qx - px) * (y - py) - (qy - py) * (x - px);      // orient2(p, q, r)
      if (o < 0.0 || (o == 0.0 && d > qd)) { q = i; qx = x; qy = y; qd = d; }
    }
    for (int off = 32; off > 0; off >>= 1) {
      const int oq = __shfl_xor(q, off);
      const double ox = __shfl_xor(qx, off), oy = __shfl_xor(qy, off), od = __shfl_xor(qd, off);
      if (oq < 0) continue;
      if (q < 0) { q = oq; qx = ox; qy = oy; qd = od; continue; }
      const double o = (qx - px) * (oy - py) - (qy - py) * (ox - px);
      // (both lanes of a pair must come to the same winner: on an exact tie of direction and distance the lower ring index)
      if (o < 0.0 || (o == 0.0 && (od > qd || (od == qd && oq < q)))) { q = oq; qx = ox; qy = oy; qd = od; }
    }
    if (q < 0 || q == p0) break;
    p = q; px = qx; py = qy;
  }
  if (nh > n) nh = n;
  __syncthreads();                                   // lane 0's hull[] stores are visible
  // Melkman's list starts at the hull vertex with the highest ring index: the caller reads hull[(k + rot) % nh]
  int mi = -1, mk = 0;
  for (int k = lane; k < nh; k += 64) { const int v = hull[k]; if (v > mi) { mi = v; mk = k; } }
  for (int off = 32; off > 0; off >>= 1) { const int oi = __shfl_xor(mi, off), ok = __shfl_xor(mk, off); if (oi > mi) { mi = oi; mk = ok; } }
  *rot = mk;
  return nh;
}

// sh::convex_hull_simple_polygon (sh_scalar.h) for one lane that has to wait for every LDS answer.  The plain form reads two deque
// slots and then their points for every orientation test: ~20 dependent LDS round trips per vertex, 1 us per vertex, the whole
// kernel one slice's walk long (0.35 ms).  Here the deque holds COORDINATES (dqx, dqy: slots 0 .. 2n + 1, the hull ends up in
// dqx/dqy[*first .. *first + nh)), the three points at either end of it stay in registers (B0 B1 B2 from the bottom, T0 T1 T2 from
// the top) and the next ring vertex is fetched while the current one is tested: a vertex that leaves the hull alone waits for
// nothing, a push is two stores, a pop shifts the registers and refills the third one with a load nobody waits for unless a
// second pop follows at once.  Same tests on the same values in the same order as the plain form: the same hull.
//   invariants: Bk = slot bot + k for every k with bot + k <= top, Tk = slot top - k for top - k >= bot (a slot's content never
//   changes while it is live; a push that makes a slot live which the other end's third register stands for refreshes it).
// Returns -1 when the first three vertices are collinear (the caller takes the sort-based hull of the plain form).
__device__ inline double orient2v(double ax, double ay, double bx, double by, double cx, double cy) {
  return (bx - ax) * (cy - ay) - (by - ay) * (cx - ax);
}
__device__ inline int hull_simple_polygon_cached(const double* xy, int n, double* dqx, double* dqy, int* first) {
  const double o = orient2(xy, xy + 2, xy + 4);
  if (o == 0.0) return -1;
  int bot = n - 2, top = bot + 3;
  const double v0x = xy[0], v0y = xy[1], v1x = xy[2], v1y = xy[3], v2x = xy[4], v2y = xy[5];
  // slots bot .. top = 2, (0, 1 | 1, 0), 2
  double b0x = v2x, b0y = v2y, t0x = v2x, t0y = v2y;
  double b1x = o > 0 ? v0x : v1x, b1y = o > 0 ? v0y : v1y;
  double t1x = o > 0 ? v1x : v0x, t1y = o > 0 ? v1y : v0y;
  double b2x = t1x, b2y = t1y, t2x = b1x, t2y = b1y;
  dqx[bot] = v2x; dqy[bot] = v2y; dqx[bot + 1] = b1x; dqy[bot + 1] = b1y; dqx[bot + 2] = t1x; dqy[bot + 2] = t1y; dqx[top] = v2x; dqy[top] = v2y;
#ifdef SH_TE_CLOCK
  long long c_int = 0, c_hull = 0; int n_int = 0, n_hull = 0;
#endif
  double nx = n > 3 ? xy[6] : 0.0, ny = n > 3 ? xy[7] : 0.0;
  for (int i = 3; i < n; ++i) {
#ifdef SH_TE_CLOCK
    const long long cs_ = clock64();
#endif
    const double vx = nx, vy = ny;
    if (i + 1 < n) { nx = xy[2 * i + 2]; ny = xy[2 * i + 3]; }      // (in flight during the tests below)
    double ob = orient2v(b0x, b0y, b1x, b1y, vx, vy);
    double ot = orient2v(t1x, t1y, t0x, t0y, vx, vy);
#ifdef SH_TE_CLOCK
    if (ob > 0 && ot > 0) { c_int += clock64() - cs_; ++n_int; continue; }
#else
    if (ob > 0 && ot > 0) continue;
#endif
    while (top - bot >= 2 && ob <= 0) {
      ++bot;
      b0x = b1x; b0y = b1y; b1x = b2x; b1y = b2y;
      b2x = dqx[bot + 2]; b2y = dqy[bot + 2];      // (slot bot + 2 <= 2n + 1; not live when bot + 2 > top: never used then, see the pushes)
      ob = orient2v(b0x, b0y, b1x, b1y, vx, vy);
    }
    --bot;
    dqx[bot] = vx; dqy[bot] = vy;
    b2x = b1x; b2y = b1y; b1x = b0x; b1y = b0y; b0x = vx; b0y = vy;
    if (top - bot == 2) { t2x = vx; t2y = vy; }      // the new bottom slot is the top's third
    // (the bottom of the deque never reaches its top two slots: `ot` still is the test of slots top - 1, top)
    while (top - bot >= 2 && ot <= 0) {
      --top;
      t0x = t1x; t0y = t1y; t1x = t2x; t1y = t2y;
      t2x = dqx[top >= 2 ? top - 2 : 0]; t2y = dqy[top >= 2 ? top - 2 : 0];
      ot = orient2v(t1x, t1y, t0x, t0y, vx, vy);
    }
    ++top;
    dqx[top] = vx; dqy[top] = vy;
    t2x = t1x; t2y = t1y; t1x = t0x; t1y = t0y; t0x = vx; t0y = vy;
    if (top - bot == 2) { b2x = vx; b2y = vy; }      // the new top slot is the bottom's third
#ifdef SH_TE_CLOCK
    c_hull += clock64() - cs_; ++n_hull;
#endif
  }
#ifdef SH_TE_CLOCK
  dqx[2 * n + 4] = (double)c_int; dqx[2 * n + 5] = (double)n_int; dqx[2 * n + 6] = (double)c_hull; dqx[2 * n + 7] = (double)n_hull;
#endif
  *first = bot;
  return top - bot;
}

// One wave per (humerus, distal slice): lane 0 builds the hull of the ring (Melkman, O(n), LDS deque),
// then the lanes share the hull edges of sh::min_area_rect (same arithmetic per edge; first minimum
// in hull order wins, as in the sequential routine).
// Two capacity tiers share the grid like k_slice_link (CAP = SH_SMALLSEG: 24 KB of LDS; the ring is staged in LDS first --
// lane 0's hull walk is latency-bound when every point comes from global memory).
// -DSH_TE_WRAP selects wave_hull_wrap (measured round 3: 0.21 ms for the hull phase either way -- the 64-lane tournament pays ~40
// ds_bpermute per step for its reduction -- so the established one-lane walk stays the default)
#ifdef SH_TE_WRAP
constexpr bool getenv_te_melkman = false;
#else
constexpr bool getenv_te_melkman = true;
#endif

template <int CAP>
__global__ void __launch_bounds__(64)
k_te_rows(const double* __restrict__ ring, const int* __restrict__ ring_n, double* __restrict__ rects /*[B][37][7]*/, int B,
          const long long* __restrict__ ovf_roff /*distal set: >= 0 = the slice's ring is in the overflow pool (k_te_rows_huge takes it)*/) {
  __shared__ double dqx[2 * CAP + 8], dqy[2 * CAP + 8];      // coordinate deque of the walk; the hull's points afterwards
  __shared__ double s_xy[2 * (CAP + 1)];
  __shared__ int nh_s, first_s;
  const int gid = blockIdx.x, lane = threadIdx.x;
  const int b = gid / SH_TE_NROWS, j = gid % SH_TE_NROWS;
  const size_t pl = (size_t)b * SH_NDIST + SH_TE_ROW0 + j;
  const double* gxy = ring + pl * (SH_MAXSEG + 1) * 2;
  const int n = ring_n[pl];
  if (CAP == SH_SMALLSEG ? n > SH_SMALLSEG : n <= SH_SMALLSEG) return;      // the other tier's slice
  if (ovf_roff[pl] >= 0) return;
  double* o = rects + (size_t)gid * 7;
  if (n < 3) { if (lane < 7) o[lane] = 0.0; return; }
  for (int q = lane; q < 2 * (n + 1); q += 64) s_xy[q] = gxy[q];
  __syncthreads();
#if defined(SH_ABL_TE) && SH_ABL_TE == 1
  return;
#endif
  const double* xy = s_xy;
  const double *hx = dqx, *hy = dqy;
  int nh;
  if (getenv_te_melkman) {
    if (lane == 0) {
      int first = 0;
#ifdef SH_TE_CLOCK
      const long long c0_ = clock64();
#endif
      int h = hull_simple_polygon_cached(xy, n, dqx, dqy, &first);   // rings are simple polygons in boundary order
#ifdef SH_TE_CLOCK
      dqx[2 * CAP + 7] = (double)(clock64() - c0_); dqy[2 * CAP + 7] = (double)n + 1e-3 * h;
      dqy[2 * CAP + 3] = dqx[2 * n + 4]; dqy[2 * CAP + 4] = dqx[2 * n + 5]; dqy[2 * CAP + 5] = dqx[2 * n + 6]; dqy[2 * CAP + 6] = dqx[2 * n + 7];
#endif
      if (h < 0) {      // first three vertices collinear: the sort-based hull, its index arrays in the (idle) upper half of the deque
        int* idx = (int*)(dqx + CAP + 4);
        int* hl = (int*)(dqy + CAP + 4);
        h = convex_hull_simple_polygon(xy, n, idx, hl);
        for (int k = 0; k < h; ++k) { dqx[k] = xy[2 * hl[k]]; dqy[k] = xy[2 * hl[k] + 1]; }
      }
      nh_s = h; first_s = first;
    }
    __syncthreads();
    nh = nh_s;
    hx = dqx + first_s; hy = dqy + first_s;
  } else {
    int rot = 0;
    int* hull = (int*)(dqx + CAP + 4);
    nh = wave_hull_wrap(xy, n, hull, lane, &rot);
    __syncthreads();
    double px[(CAP + 63) / 64], py[(CAP + 63) / 64];      // (gather through registers: hull[] lives in the arrays being written)
#pragma unroll
    for (int t = 0; t < (CAP + 63) / 64; ++t) { const int k = lane + 64 * t; if (k < nh) { int src = k + rot; if (src >= nh) src -= nh; px[t] = xy[2 * hull[src]]; py[t] = xy[2 * hull[src] + 1]; } }
    __syncthreads();
#pragma unroll
    for (int t = 0; t < (CAP + 63) / 64; ++t) { const int k = lane + 64 * t; if (k < nh) { dqx[k] = px[t]; dqy[k] = py[t]; } }
    __syncthreads();
  }
#if defined(SH_ABL_TE) && SH_ABL_TE == 2
  if (nh >= 0) return;
#endif
  double best = 1e300;
  int bi = 0x7fffffff;
  Rect2 r;
  r.cx = r.cy = r.mx = r.my = r.L = r.W = r.area = 0.0;
  for (int i = lane; i < nh; i += 64) {
    int i2 = i + 1 == nh ? 0 : i + 1;
    double ex = hx[i2] - hx[i], ey = hy[i2] - hy[i];
    double ln = hypot(ex, ey);
    if (ln == 0) continue;
    ex /= ln; ey /= ln;
    double nx = -ey, ny = ex;
    double amin = 1e300, amax = -1e300, bmin = 1e300, bmax = -1e300;
    for (int k = 0; k < nh; ++k) {
      double a = hx[k] * ex + hy[k] * ey, bb = hx[k] * nx + hy[k] * ny;
      amin = a < amin ? a : amin; amax = a > amax ? a : amax;
      bmin = bb < bmin ? bb : bmin; bmax = bb > bmax ? bb : bmax;
    }
    double ea = amax - amin, eb = bmax - bmin, area = ea * eb;
    if (area < best) {     // strided i ascends per lane, so '<' keeps the lane's first minimum
      best = area; bi = i;
      double ca = 0.5 * (amax + amin), cb = 0.5 * (bmax + bmin);
      r.cx = ex * ca + nx * cb; r.cy = ey * ca + ny * cb; r.area = area;
      if (ea >= eb) { r.mx = ex; r.my = ey; r.L = ea; r.W = eb; }
      else { r.mx = nx; r.my = ny; r.L = eb; r.W = ea; }
    }
  }
  // wave argmin (area, then edge index)
  double wb = best;
  int wi = bi;
  for (int off = 32; off > 0; off >>= 1) {
    double ob = __shfl_down(wb, off);
    int oi = __shfl_down(wi, off);
    if (ob < wb || (ob == wb && oi < wi)) { wb = ob; wi = oi; }
  }
  wi = __shfl(wi, 0);
  if (bi == wi && wi != 0x7fffffff) { o[0] = r.cx; o[1] = r.cy; o[2] = r.mx; o[3] = r.my; o[4] = r.L; o[5] = r.W; o[6] = r.area; }
  else if (wi == 0x7fffffff && lane < 7) o[lane] = 0.0;
#ifdef SH_TE_CLOCK
  __syncthreads();
  if (lane == 0) { o[5] = dqx[2 * CAP + 7]; o[6] = dqy[2 * CAP + 7]; o[0] = dqy[2 * CAP + 3]; o[1] = dqy[2 * CAP + 4]; o[2] = dqy[2 * CAP + 5]; o[3] = dqy[2 * CAP + 6]; }      // debug build: cycles of the walk, n + h / 1000, cycles / count of untouched and of hull-changing vertices
#endif
  (void)B;
}

// the same for a distal slice whose ring lives in the overflow pool (k_ovf.h): hull deque and hull points in the slice's workspace
__global__ void __launch_bounds__(64)
k_te_rows_huge(OvfPools P, OvfSet S, const int* __restrict__ ring_n, double* __restrict__ rects /*[B][37][7]*/) {
  __shared__ int nh_s;
  const int nlist = *S.nlist, lane = threadIdx.x;
  for (int it = blockIdx.x; it < nlist; it += gridDim.x) {
    const int pl = S.list[it];
    const int b = pl / SH_NDIST, j = pl % SH_NDIST - SH_TE_ROW0;
    if (j < 0 || j >= SH_TE_NROWS) continue;
    const double* xy = P.ring + 2 * S.roff[pl];
    const int n = ring_n[pl];
    double* o = rects + ((size_t)b * SH_TE_NROWS + j) * 7;
    if (n < 3) { if (lane < 7) o[lane] = 0.0; continue; }
    double* hx = (double*)(P.work + S.woff[pl]);
    double* hy = hx + n;
    int* dq = (int*)(hy + n);
    int* hull = dq + 2 * n + 8;
    if (lane == 0) nh_s = convex_hull_simple_polygon(xy, n, dq, hull);
    __syncthreads();
    const int nh = nh_s;
    for (int k = lane; k < nh; k += 64) { hx[k] = xy[2 * hull[k]]; hy[k] = xy[2 * hull[k] + 1]; }
    __syncthreads();
    double best = 1e300;
    int bi = 0x7fffffff;
    Rect2 r;
    r.cx = r.cy = r.mx = r.my = r.L = r.W = r.area = 0.0;
    for (int i = lane; i < nh; i += 64) {
      const int i2 = i + 1 == nh ? 0 : i + 1;
      double ex = hx[i2] - hx[i], ey = hy[i2] - hy[i];
      const double ln = hypot(ex, ey);
      if (ln == 0) continue;
      ex /= ln; ey /= ln;
      const double nx = -ey, ny = ex;
      double amin = 1e300, amax = -1e300, bmin = 1e300, bmax = -1e300;
      for (int k = 0; k < nh; ++k) {
        const double a = hx[k] * ex + hy[k] * ey, bb = hx[k] * nx + hy[k] * ny;
        amin = a < amin ? a : amin; amax = a > amax ? a : amax;
        bmin = bb < bmin ? bb : bmin; bmax = bb > bmax ? bb : bmax;
      }
      const double ea = amax - amin, eb = bmax - bmin, area = ea * eb;
      if (area < best) {
        best = area; bi = i;
        const double ca = 0.5 * (amax + amin), cb = 0.5 * (bmax + bmin);
        r.cx = ex * ca + nx * cb; r.cy = ey * ca + ny * cb; r.area = area;
        if (ea >= eb) { r.mx = ex; r.my = ey; r.L = ea; r.W = eb; }
        else { r.mx = nx; r.my = ny; r.L = eb; r.W = ea; }
      }
    }
    double wb = best;
    int wi = bi;
    for (int off = 32; off > 0; off >>= 1) {
      const double ob = __shfl_down(wb, off);
      const int oi = __shfl_down(wi, off);
      if (ob < wb || (ob == wb && oi < wi)) { wb = ob; wi = oi; }
    }
    wi = __shfl(wi, 0);
    if (bi == wi && wi != 0x7fffffff) { o[0] = r.cx; o[1] = r.cy; o[2] = r.mx; o[3] = r.my; o[4] = r.L; o[5] = r.W; o[6] = r.area; }
    else if (wi == 0x7fffffff && lane < 7) o[lane] = 0.0;
    __syncthreads();
  }
}

// sh::clip_halfplane_pieces (sh_scalar.h) by one wave.  What an edge of the ring contributes to the clipped chains depends on the
// edge alone once the walk starts at an entering edge (inside/outside alternate from there): an entering crossing opens a chain
// with the intersection point and its inner end, a leaving one closes the chain with the intersection point, an inner edge adds
// its end.  So the lanes take the edges 64 at a time: the side tests, the first entering edge (ballot), every edge's points at
// the position a wave prefix sum of the counts gives them, the chains' bounds and line parameters by the rank of their
// crossings -- the same expressions per edge as the sequential routine, the same poly[] and chain tables.  Pairing the
// crossings along the line, stitching the pieces and the (order-dependent) centroid sums stay on lane 0.
// 0.32 ms -> 0.08 ms for k_te_ends; the records are the sequential form's bit for bit.
// ch_i: >= 2 SH_TE_MAXCH ints, ch_d: >= 2 SH_TE_MAXCH doubles (LDS); every lane returns the number of pieces.
__device__ inline int clip_halfplane_pieces_wave(const double* pts, int n, double cx, double cy, double mx, double my, double w0,
                                                 double* cents, int cap, double* scratch, int lane, int* ch_i, double* ch_d) {
  double* poly = scratch;
  int* cb = ch_i; int* ce = ch_i + SH_TE_MAXCH;
  double* sin_ = ch_d; double* sout = ch_d + SH_TE_MAXCH;
  const double px = -my, py = mx;
  bool any_in = false, all_in = true;
  int start = -1, nent_all = 0;
  for (int base = 0; base < n; base += 64) {
    const int i = base + lane;
    bool in = false, ent = false;
    if (i < n) {
      const int j = i + 1 == n ? 0 : i + 1;
      in = ((pts[2 * i] - cx) * mx + (pts[2 * i + 1] - cy) * my - w0) > 0;
      const bool jn = ((pts[2 * j] - cx) * mx + (pts[2 * j + 1] - cy) * my - w0) > 0;
      ent = !in && jn;
    }
    const unsigned long long bi = __ballot(in), bv = __ballot(i < n), be = __ballot(ent);
    any_in |= bi != 0;
    all_in &= bi == bv;
    if (start < 0 && be) start = base + __ffsll((long long)be) - 1;
    nent_all += __popcll(be);
  }
  if (!any_in) return 0;
  if (all_in) {
    if (lane == 0 && cap > 0) { double a; poly_centroid(pts, n, cents, cents + 1, &a); }
    return 1;
  }
  if (nent_all > SH_TE_MAXCH) return -1;      // (the sequential routine gives up at the 33rd chain; nothing is written before)
  int np_ = 0, nent = 0, nlv = 0;
  for (int base = 0; base < n; base += 64) {
    const int k = base + lane;
    int cnt = 0;
    bool ii = false, jj = false;
    double xi = 0, yi = 0, xj = 0, yj = 0, fi = 0, fj = 0;
    if (k < n) {
      int i = start + k; if (i >= n) i -= n;
      const int j = i + 1 == n ? 0 : i + 1;
      xi = pts[2 * i]; yi = pts[2 * i + 1]; xj = pts[2 * j]; yj = pts[2 * j + 1];
      fi = (xi - cx) * mx + (yi - cy) * my - w0;
      fj = (xj - cx) * mx + (yj - cy) * my - w0;
      ii = fi > 0; jj = fj > 0;
      cnt = ii != jj ? (jj ? 2 : 1) : (jj ? 1 : 0);
    }
    int incl = cnt;      // inclusive wave prefix sum
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) { const int t = __shfl_up(incl, off); if (lane >= off) incl += t; }
    const int pos = np_ + incl - cnt;
    const unsigned long long bent = __ballot(ii != jj && jj), blv = __ballot(ii != jj && !jj);
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    if (ii != jj) {
      const double t = fi / (fi - fj);
      const double x = xi + t * (xj - xi), y = yi + t * (yj - yi);
      const double sp = (x - cx) * px + (y - cy) * py;
      poly[2 * pos] = x; poly[2 * pos + 1] = y;
      if (jj) {
        const int c = nent + __popcll(bent & below);
        if (c < SH_TE_MAXCH) { cb[c] = pos; sin_[c] = sp; }
        poly[2 * pos + 2] = xj; poly[2 * pos + 3] = yj;
      } else {
        const int c = nlv + __popcll(blv & below);
        if (c < SH_TE_MAXCH) { sout[c] = sp; ce[c] = pos + 1; }
      }
    } else if (jj) {
      poly[2 * pos] = xj; poly[2 * pos + 1] = yj;
    }
    np_ += __shfl(incl, 63);
    nent += __popcll(bent); nlv += __popcll(blv);
  }
  const int nc = nlv;
  __syncthreads();      // (one wave: orders the LDS / global writes above before lane 0 reads them)
  int npieces = 0;
  if (lane == 0) {
    // pair the crossings along the line: sort 2*nc events by s; (0,1),(2,3),...
    int ev_chain[2 * SH_TE_MAXCH], ev_out[2 * SH_TE_MAXCH];
    double ev_s[2 * SH_TE_MAXCH];
    int ne = 0;
    for (int c = 0; c < nc; ++c) { ev_s[ne] = sin_[c]; ev_chain[ne] = c; ev_out[ne] = 0; ++ne; ev_s[ne] = sout[c]; ev_chain[ne] = c; ev_out[ne] = 1; ++ne; }
    for (int a = 1; a < ne; ++a) {
      double sv = ev_s[a]; int c = ev_chain[a], o = ev_out[a]; int b = a - 1;
      while (b >= 0 && ev_s[b] > sv) { ev_s[b + 1] = ev_s[b]; ev_chain[b + 1] = ev_chain[b]; ev_out[b + 1] = ev_out[b]; --b; }
      ev_s[b + 1] = sv; ev_chain[b + 1] = c; ev_out[b + 1] = o;
    }
    int next_after_out[SH_TE_MAXCH];
    for (int a = 0; a + 1 < ne; a += 2) {
      if (ev_out[a]) next_after_out[ev_chain[a]] = ev_chain[a + 1];
      if (ev_out[a + 1]) next_after_out[ev_chain[a + 1]] = ev_chain[a];
    }
    bool used[SH_TE_MAXCH];
    for (int c = 0; c < nc; ++c) used[c] = false;
    double* comp = poly + 2 * np_;
    for (int c0 = 0; c0 < nc; ++c0) {
      if (used[c0]) continue;
      int m = 0, c = c0;
      while (!used[c]) {
        used[c] = true;
        for (int k = cb[c]; k < ce[c]; ++k) { comp[2 * m] = poly[2 * k]; comp[2 * m + 1] = poly[2 * k + 1]; ++m; }
        c = next_after_out[c];
      }
      if (npieces < cap) { double a; poly_centroid(comp, m, cents + 2 * npieces, cents + 2 * npieces + 1, &a); }
      ++npieces;
    }
  }
  return __shfl(npieces, 0);
}

__global__ void k_te_ends(const double* __restrict__ ring, const int* __restrict__ ring_n, const double* __restrict__ rects,
                          const double* __restrict__ distal_zs, const double* __restrict__ T_obb, double* __restrict__ ends_ct /*[B][6]: the two ends, in piece order*/,
                          int* __restrict__ te_row, int* __restrict__ err, int B, OvfPools P, OvfSet S) {
  // one 64-lane workgroup per humerus: the lanes stage the chosen ring in LDS and clip it together (clip_halfplane_pieces_wave)
  __shared__ double s_xy[2 * (SH_MAXSEG + 1)];
  __shared__ double s_scr[2 * SH_MAXSEG + 8 * SH_TE_MAXCH];
  __shared__ double s_cents[2 * 16], s_chd[2 * SH_TE_MAXCH];
  __shared__ int s_chi[2 * SH_TE_MAXCH];
  __shared__ int s_k;
  const int b = blockIdx.x;
  if (b >= B) return;
  const double* R = rects + (size_t)b * SH_TE_NROWS * 7;
  if (threadIdx.x == 0) {
    int k = 0;
    for (int j = 1; j < SH_TE_NROWS; ++j) if (R[j * 7 + 4] > R[k * 7 + 4]) k = j;     // dist.index(max(dist))
    s_k = k;
  }
  __syncthreads();
  const int k = s_k;
  size_t pl = (size_t)b * SH_NDIST + SH_TE_ROW0 + k;
  const bool ovf = S.roff[pl] >= 0;      // the widest slice is an overflow plane (k_ovf.h): ring in the pool, scratch in its workspace
  const double* gxy = ovf ? P.ring + 2 * S.roff[pl] : ring + pl * (SH_MAXSEG + 1) * 2;
  int n = ring_n[pl];
  if (!ovf) {
    if (n > SH_MAXSEG) n = SH_MAXSEG;
    for (int q = threadIdx.x; q < 2 * (n + 1); q += blockDim.x) s_xy[q] = gxy[q];
  }
  __syncthreads();
  if (threadIdx.x == 0) te_row[b] = SH_TE_ROW0 + k;
  const double* xy = ovf ? gxy : s_xy;
  const double* r = R + k * 7;
  double half = 0.5 * 0.999 * r[4];
  double* cents = s_cents;
  double* scr = ovf ? (double*)(P.work + S.woff[pl]) : s_scr;
  int n1 = clip_halfplane_pieces_wave(xy, n, r[0], r[1], r[2], r[3], half, cents, 8, scr, threadIdx.x, s_chi, s_chd);
  if (n1 < 0) n1 = 0;
  if (n1 > 8) n1 = 8;
  __syncthreads();
  int n2 = clip_halfplane_pieces_wave(xy, n, r[0], r[1], -r[2], -r[3], half, cents + 2 * n1, 8, scr, threadIdx.x, s_chi, s_chd);
  if (n2 < 0) n2 = 0;
  if (n2 > 8) n2 = 8;
  __syncthreads();
  if (threadIdx.x != 0) return;
  int np_ = n1 + n2;
  double* out = ends_ct + 6 * b;
  if (np_ < 2) { atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV); for (int q = 0; q < 6; ++q) out[q] = 0.0; return; }
  int bi = 0, bj = 1;
  double bd = -1.0;
  for (int i = 0; i < np_; ++i)
    for (int j = 0; j < np_; ++j) {     // np.argmax over the full (symmetric) distance matrix: first maximum
      double dx = cents[2 * i] - cents[2 * j], dy = cents[2 * i + 1] - cents[2 * j + 1];
      double d2 = dx * dx + dy * dy;
      if (d2 > bd) { bd = d2; bi = i; bj = j; }
    }
  int i0 = bi < bj ? bi : bj, i1 = bi < bj ? bj : bi;
  double z = distal_zs[pl];
  double Ti[16];
  inv_transform(T_obb + 16 * b, Ti);
  xform_pt(Ti, cents[2 * i0], cents[2 * i0 + 1], z, out);
  xform_pt(Ti, cents[2 * i1], cents[2 * i1 + 1], z, out + 3);
}

// epicondyle.py:90-96: medial end = smaller x in construct_csys(canal axis, head central axis); one lane per humerus
__device__ inline void te_orient_one(const double* ends_ct, const double* T_obb, const double* canal_axis_ct,
                                     const double* axes_obb /*[B][4][3]: +n,-n,+c,-c*/, double* te_axis_ct, int b) {
  const double* e0 = ends_ct + 6 * b;
  const double* e1 = e0 + 3;
  double* out = te_axis_ct + 6 * b;
  if (e0[0] == 0.0 && e0[1] == 0.0 && e0[2] == 0.0 && e1[0] == 0.0 && e1[1] == 0.0 && e1[2] == 0.0) { for (int q = 0; q < 6; ++q) out[q] = 0.0; return; }      // (no axis: k_te_ends flagged it)
  double Ti[16];
  inv_transform(T_obb + 16 * b, Ti);
  double central_ct[6], cs[16];
  xform_pt(Ti, axes_obb[((size_t)b * 4 + 2) * 3], axes_obb[((size_t)b * 4 + 2) * 3 + 1], axes_obb[((size_t)b * 4 + 2) * 3 + 2], central_ct);
  xform_pt(Ti, axes_obb[((size_t)b * 4 + 3) * 3], axes_obb[((size_t)b * 4 + 3) * 3 + 1], axes_obb[((size_t)b * 4 + 3) * 3 + 2], central_ct + 3);
  construct_csys(canal_axis_ct + 6 * b, central_ct, cs);
  double q0[3], q1[3];
  xform_pt(cs, e0[0], e0[1], e0[2], q0);
  xform_pt(cs, e1[0], e1[1], e1[2], q1);
  bool swap = q1[0] < q0[0];       // np.argmin: first minimum
  for (int q = 0; q < 3; ++q) { out[q] = swap ? e1[q] : e0[q]; out[3 + q] = swap ? e0[q] : e1[q]; }
}


// the record of humerus b by its workgroup (every lane copies points, lane 0 writes the scalars)
struct PackArgs {
  sh_landmarks* lm; const double* T_obb; const double* zb; const double* neck_z; const int* neck_index; const int* flipped;
  const double* canal_axis_ct; double* te_axis_ct; const double* groove_axis_ct; const double* bg_theta; const double* groove_pts_ct;
  const double* plane; double* axes_obb; const double* anp_pts_obb; const int* anp_counts; int* err;
  uint32_t mask; int B, bone_kind; const double* canal_cut /*[B][2] or null*/; double cc0, cc1;
  const unsigned long long* ray_t; const double* te_ends_ct; const double* sphere_partial; int metrics;
};
__device__ inline void pack_one(const PackArgs& A, int b) {
  sh_landmarks* lm = A.lm; const double* T_obb = A.T_obb; const double* zb = A.zb; const double* neck_z = A.neck_z; const int* neck_index = A.neck_index;
  const int* flipped = A.flipped; const double* canal_axis_ct = A.canal_axis_ct; const double* te_axis_ct = A.te_axis_ct; const double* groove_axis_ct = A.groove_axis_ct;
  const double* bg_theta = A.bg_theta; const double* groove_pts_ct = A.groove_pts_ct; const double* plane = A.plane; const double* axes_obb = A.axes_obb;
  const double* anp_pts_obb = A.anp_pts_obb; const int* anp_counts = A.anp_counts; const int* err = A.err; const uint32_t mask = A.mask; const int bone_kind = A.bone_kind;
  const double* canal_cut = A.canal_cut; const double cc0 = A.cc0, cc1 = A.cc1;
  sh_landmarks* L = lm + b;
  int tid = threadIdx.x;
  double Ti[16];
  inv_transform(T_obb + 16 * b, Ti);
  if (mask & SH_STAGE_GROOVE)
    for (int i = tid; i < SH_GROOVE_NROWS * 3; i += blockDim.x) L->groove_points[i] = groove_pts_ct[(size_t)b * SH_GROOVE_NROWS * 3 + i];
  if (mask & SH_STAGE_ANP) {
    int K = anp_counts[2 * b];
    if (K > SH_ANP_MAX_PTS) K = SH_ANP_MAX_PTS;
    for (int i = tid; i < SH_ANP_MAX_PTS; i += blockDim.x) {
      double o[3] = {0, 0, 0};
      if (i < K) { const double* p = anp_pts_obb + ((size_t)b * SH_ANP_CAP + i) * 3; xform_pt(Ti, p[0], p[1], p[2], o); }
      L->anp_points[3 * i] = o[0]; L->anp_points[3 * i + 1] = o[1]; L->anp_points[3 * i + 2] = o[2];
    }
  }
  if (tid != 0) return;
  for (int i = 0; i < 16; ++i) L->obb_transform[i] = T_obb[16 * b + i];
  L->z_length = fabs(zb[2 * b]) + fabs(zb[2 * b + 1]);
  L->neck_z = neck_z[b];
  L->neck_index = neck_index[b];
  L->flipped = flipped ? flipped[b] : 0;
  L->status = err[b];
  L->side = 0; L->neckshaft = 0.0; L->retroversion = 0.0; L->radius_curvature = 0.0;
  for (int i = 0; i < 6; ++i) { L->canal_axis[i] = canal_axis_ct[6 * b + i]; }
  if (mask & SH_STAGE_GROOVE) { for (int i = 0; i < 6; ++i) L->groove_axis[i] = groove_axis_ct[6 * b + i]; L->bg_theta = bg_theta[b]; }
  if (mask & SH_STAGE_ANP) {
    const double* pl = plane + 6 * b;
    xform_pt(Ti, pl[0], pl[1], pl[2], L->anp_plane_point);
    for (int r = 0; r < 3; ++r) L->anp_plane_normal[r] = (Ti[r * 4] * pl[3] + Ti[r * 4 + 1] * pl[4]) + Ti[r * 4 + 2] * pl[5];
    for (int ray = 0; ray < 4; ++ray) {
      const double* a = axes_obb + ((size_t)b * 4 + ray) * 3;
      double* dst = ray < 2 ? L->anp_axis_normal + 3 * ray : L->anp_axis_central + 3 * (ray - 2);
      xform_pt(Ti, a[0], a[1], a[2], dst);
    }
    L->n_anp = anp_counts[2 * b];
    L->n_articular = anp_counts[2 * b + 1];
  }
  if (mask & SH_STAGE_TE) for (int i = 0; i < 6; ++i) L->te_axis[i] = te_axis_ct[6 * b + i];
  L->canal_cutoff[0] = canal_cut ? canal_cut[2 * b] : cc0;
  L->canal_cutoff[1] = canal_cut ? canal_cut[2 * b + 1] : cc1;
  if (mask & SH_STAGE_CSYS) {
    if (mask & SH_STAGE_ANP) construct_csys(L->canal_axis, L->anp_axis_normal, L->csys_articular);     // bone.py:57-59 apply_csys_canal_articular
    if (bone_kind == SH_BONE_PROXIMAL) { for (int i = 0; i < 16; ++i) L->csys[i] = L->csys_articular[i]; }
    else construct_csys(L->canal_axis, L->te_axis, L->csys);                                          // bone.py:150
  }
}

// bone.py:155 `mesh_ct.copy().apply_transform(T)` for the batch: float32 CT vertices of mesh b -> float64 vertices in the
// coordinate system of its record (T = landmarks[b].csys: canal / trans-epicondylar, or canal / articular for a cut
// humerus).  HBM bound: 12 B read + 24 B written per vertex, coalesced.
__global__ void k_apply_csys(const sh_landmarks* __restrict__ lm, const float* __restrict__ verts, const long long* __restrict__ voff,
                             double* __restrict__ out) {
  const int b = blockIdx.y;
  double T[16];
  for (int i = 0; i < 16; ++i) T[i] = lm[b].csys[i];
  const long long v0 = voff[b], n = voff[b + 1] - v0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float* p = verts + 3 * (v0 + i);
    double o[3];
    xform_pt(T, (double)p[0], (double)p[1], (double)p[2], o);
    double* q = out + 3 * (v0 + i);
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
  }
}

// utils.unitxyz_to_spherical (utils.py:321-332): theta, phi in degrees
SH_HD void unitxyz_to_spherical_deg(const double* v, double* theta, double* phi) {
  double r = sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]);
  *theta = atan2(v[1], v[0]) * (180.0 / M_PI);
  *phi = acos(v[2] / r) * (180.0 / M_PI);
}

// Metrics of bone_props.py from the packed record + the articular mask; one workgroup per humerus.
//   side :30-47, retroversion :64-85 (landmarks in CT), neckshaft :97-112 on lane 0;
//   radius_curvature :115-148: least-squares sphere through every mask pixel's (r cos t, r sin t, z),
//   solved from the 4x4 normal equations of the mean-centred points (the fit is shift invariant).
// Sphere-fit partial sums over the articular mask, SH_SPH_PARTS workgroups per humerus.  Points are taken
// relative to the neck-plane point (any shift near the data keeps the normal equations well conditioned; the
// fit itself is shift invariant).  partial[b][part][14] = sum q q^T (6), |q|^2, q|q|^2 (3), q (3), count.
#define SH_SPH_PARTS 16
// Round 4: the points' angles are the rows of a uniform theta grid (k_anp_rows: t_j = linspace(t0, t1, 512)[j], stored rolled by the
// row's `roll`), so a wave that owns a row takes ONE sincos per lane (its first column) and walks its other seven columns by a
// rotation of 64 grid steps (or 64 - 512 steps where the column index wraps) -- it took a cos and a sin per mask pixel, and behind a
// UNet pass on the 32 reserved CUs that was 0.33 ms of f64 range reduction on the lane's critical path.  Angle error after seven
// rotations ~1e-15 (radius_curvature is held to 1e-6 mm against the oracle).  32 rows per workgroup, 8 per wave.
__global__ void __launch_bounds__(256)
k_sphere_partial(const unsigned long long* __restrict__ maskbits /*[B][512][8]: k_anp_edge_count*/, const double* __restrict__ raw, const double* __restrict__ t01 /*[B][512][2]: the ends of the rows' theta grids (k_anp_rows)*/,
                 const int* __restrict__ roll /*[B][512]*/, const double* __restrict__ prox_zs, const double* __restrict__ plane, double* __restrict__ partial) {
  __shared__ double sh[14 * 4];
  constexpr int M = SH_MPROX, RPW = SH_ANP_ROWS / SH_SPH_PARTS;      // 32 rows per workgroup
  const int b = blockIdx.y, part = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long* mbits = maskbits + (size_t)b * SH_ANP_ROWS * (M / 64);
  const double* rr = raw + (size_t)b * SH_IMG;
  const double* zz = prox_zs + (size_t)b * SH_NPROX + SH_ANP_ROW0;
  const double m[3] = {plane[6 * b], plane[6 * b + 1], plane[6 * b + 2]};
  double a[14];
  for (int k = 0; k < 14; ++k) a[k] = 0.0;
  for (int rw = wave; rw < RPW; rw += 4) {
    const int row = part * RPW + rw;
    const double t0 = t01[2 * ((size_t)b * SH_ANP_ROWS + row)], t1 = t01[2 * ((size_t)b * SH_ANP_ROWS + row) + 1];      // the grid of k_anp_rows
    const int kbest = roll[(size_t)b * SH_ANP_ROWS + row];
    const double step = (t1 - t0) / (double)(M - 1);
    int j = lane + kbest; if (j >= M) j -= M;                          // grid index of storage column `lane`
    // lanes 0 / 1 also take the two rotation angles; every lane its own first angle
    const double ang = lane == 0 ? 64.0 * step : (lane == 1 ? (double)(64 - M) * step : 0.0);
    double sa, ca;
    sincos(ang, &sa, &ca);
    const double c64 = __shfl(ca, 0), s64 = __shfl(sa, 0), cw = __shfl(ca, 1), sw = __shfl(sa, 1);
    double sn, cs;
    sincos(linspace_at(t0, t1, M, j), &sn, &cs);
    const double z = zz[row] - m[2];
    const size_t base = (size_t)row * M + lane;
#pragma unroll
    for (int k = 0; k < M / 64; ++k) {
      if (mbits[(size_t)row * (M / 64) + k] >> lane & 1ull) {
        const double r = rr[base + 64 * k];
        const double q[3] = {r * cs - m[0], r * sn - m[1], z};
        const double q2 = (q[0] * q[0] + q[1] * q[1]) + q[2] * q[2];
        a[0] += q[0] * q[0]; a[1] += q[0] * q[1]; a[2] += q[0] * q[2]; a[3] += q[1] * q[1]; a[4] += q[1] * q[2]; a[5] += q[2] * q[2];
        a[6] += q2; a[7] += q[0] * q2; a[8] += q[1] * q2; a[9] += q[2] * q2; a[10] += q[0]; a[11] += q[1]; a[12] += q[2]; a[13] += 1.0;
      }
      j += 64;
      const bool wrap = j >= M;
      if (wrap) j -= M;
      const double cr = wrap ? cw : c64, sr = wrap ? sw : s64;
      const double cn = cs * cr - sn * sr;
      sn = sn * cr + cs * sr;
      cs = cn;
    }
  }
  block_sum<14>(a, sh, tid, 4);
  if (tid < 14) partial[((size_t)b * SH_SPH_PARTS + part) * 14 + tid] = a[tid];
}

__device__ inline void metrics_one(sh_landmarks* lm, const double* partial, int* err, int bone_kind, int b) {
  const int tid = threadIdx.x;
  sh_landmarks* L = lm + b;
  double radius = 0.0;
  {
    double a[14];
    for (int k = 0; k < 14; ++k) {      // fixed order: deterministic
      double s = 0.0;
      for (int p = 0; p < SH_SPH_PARTS; ++p) s += partial[((size_t)b * SH_SPH_PARTS + p) * 14 + k];
      a[k] = s;
    }
    const double n = a[13];
    if (tid == 0 && n >= 4.0) {
      // A = [2q, 1]: N = A^T A, g = A^T f with f = |q|^2
      double N[16] = {4 * a[0], 4 * a[1], 4 * a[2], 2 * a[10],
                      4 * a[1], 4 * a[3], 4 * a[4], 2 * a[11],
                      4 * a[2], 4 * a[4], 4 * a[5], 2 * a[12],
                      2 * a[10], 2 * a[11], 2 * a[12], n};
      double g[4] = {2 * a[7], 2 * a[8], 2 * a[9], a[6]};
      double Ni[16];
      if (mat4_inv(N, Ni)) {
        double C[4];
        for (int r = 0; r < 4; ++r) C[r] = ((Ni[r * 4] * g[0] + Ni[r * 4 + 1] * g[1]) + Ni[r * 4 + 2] * g[2]) + Ni[r * 4 + 3] * g[3];
        radius = sqrt(((C[0] * C[0] + C[1] * C[1]) + C[2] * C[2]) + C[3]);
      } else atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV);
    }
  }
  if (tid != 0) return;
  L->radius_curvature = radius;
  // Side (bone_props.py:30-47)
  double T[16], p[3];
  construct_csys(L->canal_axis, L->anp_axis_central, T);
  double sy = 0.0;
  for (int i = 0; i < SH_GROOVE_NROWS; ++i) { xform_pt(T, L->groove_points[3 * i], L->groove_points[3 * i + 1], L->groove_points[3 * i + 2], p); sy += p[1]; }
  L->side = (sy / SH_GROOVE_NROWS <= 0) ? 0 : 1;
  // NeckShaft (:97-112)
  {
    double a0[3], a1[3], v[3], th, ph;
    construct_csys(L->canal_axis, L->anp_axis_normal, T);
    xform_pt(T, L->anp_axis_normal[0], L->anp_axis_normal[1], L->anp_axis_normal[2], a0);
    xform_pt(T, L->anp_axis_normal[3], L->anp_axis_normal[4], L->anp_axis_normal[5], a1);
    for (int k = 0; k < 3; ++k) v[k] = a0[k] - a1[k];
    double nn = norm3(v);
    for (int k = 0; k < 3; ++k) v[k] /= nn;
    unitxyz_to_spherical_deg(v, &th, &ph);
    L->neckshaft = 180.0 - ph;
  }
  // RetroVersion (:64-85) with the neck-normal axis in CT (identity Transform); a proximal humerus has no
  // epicondyles, `ProximalHumerus` has no retroversion() (bone.py:45-51)
  if (bone_kind == SH_BONE_PROXIMAL) L->retroversion = nan("");
  else {
    double a0[3], a1[3], v[3], th, ph;
    xform_pt(L->csys, L->anp_axis_normal[0], L->anp_axis_normal[1], L->anp_axis_normal[2], a0);
    xform_pt(L->csys, L->anp_axis_normal[3], L->anp_axis_normal[4], L->anp_axis_normal[5], a1);
    for (int k = 0; k < 3; ++k) v[k] = a0[k] - a1[k];
    double nn = norm3(v);
    for (int k = 0; k < 3; ++k) v[k] /= nn;
    v[0] = -1.0 * v[0];
    unitxyz_to_spherical_deg(v, &th, &ph);
    L->retroversion = L->side == 1 ? -th : th;
  }
}

// The end of a run as ONE launch (were four: k_rays, k_te_orient, k_pack, k_metrics -- each a handful of lanes per humerus in a
// dependent chain behind the UNet pass): one workgroup per humerus, the steps separated by workgroup barriers.
//   rays (anatomic_neck.py:174-236)  the four axis points from the nearest hit parameters of k_rays_hit
//   trans-epicondylar order (epicondyle.py:90-96)  medial end first, needs the head's central axis
//   record (bone.py:146-157, utils.py:289-318)  landmarks in CT, csys
//   metrics (bone_props.py)  side, neck-shaft, retroversion, radius of curvature from k_sphere_partial's sums
__global__ void __launch_bounds__(256)
k_tail(PackArgs A) {
  const int b = blockIdx.x, tid = threadIdx.x;
  if (b >= A.B) return;
  if ((A.mask & SH_STAGE_ANP) && tid < 4) rays_point(A.plane, A.ray_t, A.axes_obb, A.err, b, tid);
  __syncthreads();
  if ((A.mask & SH_STAGE_TE) && tid == 0) te_orient_one(A.te_ends_ct, A.T_obb, A.canal_axis_ct, A.axes_obb, A.te_axis_ct, b);
  __syncthreads();
  pack_one(A, b);
  __syncthreads();
  if (A.metrics && tid < 64) metrics_one(A.lm, A.sphere_partial, A.err, A.bone_kind, b);
}

}  // namespace sh
