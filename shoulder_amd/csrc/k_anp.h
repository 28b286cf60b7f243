// k_anp.h -- anatomic neck around the UNet (reference src/shoulder/humerus/anatomic_neck.py).
//   k_anp_rows    :40-54   even-theta re-interpolation + roll to the groove angle (one lane per row)
//   k_anp_scale   :56-58   global min-max (taken by k_anp_rows; sklearn MinMaxScaler arithmetic) -> float32 image (:73-75)
//   k_anp_edges   :79-118  mask = logit > 0, |diff(mask, prepend=0)| along theta, compaction -> points
//   k_anp_plane   :123-153 plane fit (covariance, smallest eigenvector) + LSQ-ellipse centre
//   k_rays        :174-236 4 rays (+-normal, +-central) vs all triangles, nearest hit (B-6)
// Buffers: anp.raw [B][512][512] f64, anp.shft_theta [B][512][512] f64, anp.roll [B][512] i32,
// anp.image [B][512][512] f32, anp.logits [B][512][512] f32, anp.points_obb [B][ANP_CAP][3] f64,
// anp.counts [B][2] i32 (edge points, mask pixels), anp.plane [B][6] f64, anp.axes_obb [B][4][3] f64.
#pragma once
#include "k_groove.h"

namespace sh {

#define SH_IMG (SH_ANP_ROWS * SH_MPROX)
#define SH_ANP_CAP SH_IMG      // edge pixels kept per humerus: a mask has no more edge pixels than pixels (was 65 536 + SH_ERR_CAPACITY)

__global__ void k_anp_rows(const double* __restrict__ itr_start /*[B][600][2][512]*/, const double* __restrict__ bg_theta,
                           double* __restrict__ raw, double* __restrict__ t01 /*[B][rows][2]: the ends of the row's sampling grid.  The shifted
                           theta image of anatomic_neck.py:44-52 is linspace(t0, t1, 512) rolled by `roll`: k_anp_edges evaluates it at the
                           edge pixels instead of reading a 134 MB image back*/, int* __restrict__ roll, int B,
                           unsigned long long* __restrict__ mm_enc /*[B][2]: minimum / maximum of the humerus' image, order-preserving encoding, the maximum complemented (both words all ones before this launch)*/) {
  // One wave per image row, the (theta, r) row in LDS.  np.interp's search carries the previous index as a
  // guess; on a sorted xp the answer does not depend on the guess, so when theta[:-1] is non-decreasing every
  // lane interpolates its own samples; otherwise lane 0 replays NumPy's sequential loop exactly.
  __shared__ double s_t[SH_MPROX], s_r[SH_MPROX];
  __shared__ int s_unsorted;
  const int gid = blockIdx.x, lane = threadIdx.x;
  const int b = gid / SH_ANP_ROWS, i = gid % SH_ANP_ROWS;
  const int M = SH_MPROX;
  const double* th = itr_start + ((size_t)b * SH_NPROX + SH_ANP_ROW0 + i) * 2 * M;
  for (int k = lane; k < M; k += 64) { s_t[k] = th[k]; s_r[k] = th[M + k]; }
  if (lane == 0) s_unsorted = 0;
  __syncthreads();
  for (int k = 1 + lane; k < M - 1; k += 64) if (s_t[k] < s_t[k - 1]) s_unsorted = 1;
  const double t0 = s_t[0], t1 = s_t[M - 2];
  const double bg = bg_theta[b];
  // argmin |t_sampling - bg_theta| (first minimum)
  double dbest = 1e300;
  int kb = 0x7fffffff;
  for (int j = lane; j < M; j += 64) {
    double d = fabs(linspace_at(t0, t1, M, j) - bg);
    if (d < dbest) { dbest = d; kb = j; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    double od = __shfl_down(dbest, off);
    int ok = __shfl_down(kb, off);
    if (od < dbest || (od == dbest && ok < kb)) { dbest = od; kb = ok; }
  }
  const int kbest = __shfl(kb, 0);
  if (lane == 0) { roll[gid] = kbest; t01[2 * (size_t)gid] = t0; t01[2 * (size_t)gid + 1] = t1; }
  __syncthreads();
  double* o_r = raw + (size_t)gid * M;
  double lo = 1e300, hi = -1e300;      // the row's share of the image's minimum / maximum (MinMaxScaler, :56-58): a separate pass read the image again
  if (!s_unsorted) {
    // On a sorted xp np.interp's search returns the largest index with xp[i] <= x whatever its guess: the lane's eight samples take
    // that index by a fixed-length branch-free search side by side (nine LDS reads deep in all, where eight guessed searches one
    // after the other were ~12 dependent reads each: the row's latency is what the chain pays for beside a UNet pass), then the
    // arithmetic of np_interp_step.
    constexpr int NS = SH_MPROX / 64, LEN = SH_MPROX - 1;
    double t[NS];
    int pos[NS];
#pragma unroll
    for (int u = 0; u < NS; ++u) { t[u] = linspace_at(t0, t1, M, lane + 64 * u); pos[u] = 0; }
#pragma unroll
    for (int stp = 256; stp >= 1; stp >>= 1) {
#pragma unroll
      for (int u = 0; u < NS; ++u) { const int q = pos[u] + stp; if (q < LEN && s_t[q] <= t[u]) pos[u] = q; }
    }
#pragma unroll
    for (int u = 0; u < NS; ++u) {
      const int j = lane + 64 * u;
      const double x = t[u];
      int i = pos[u];
      if (x > s_t[LEN - 1]) i = LEN; else if (x < s_t[0]) i = -1;
      double v;
      if (i == -1) v = s_r[0];
      else if (i == LEN) v = s_r[LEN - 1];
      else if (i == LEN - 1) v = s_r[i];
      else if (s_t[i] == x) v = s_r[i];
      else {
        const double slope = (s_r[i + 1] - s_r[i]) / (s_t[i + 1] - s_t[i]);
        v = slope * (x - s_t[i]) + s_r[i];
        if (v != v) {      // NaN: NumPy retries from the right neighbour
          v = slope * (x - s_t[i + 1]) + s_r[i + 1];
          if (v != v && s_r[i] == s_r[i + 1]) v = s_r[i];
        }
      }
      int dst = j - kbest; if (dst < 0) dst += M;
      o_r[dst] = v;
      lo = fmin(lo, v); hi = fmax(hi, v);
    }
  } else if (lane == 0) {
    int jg = 0;
    for (int j = 0; j < M; ++j) {
      double t = linspace_at(t0, t1, M, j);
      double v = np_interp_step(t, s_t, s_r, M - 1, &jg);
      int dst = j - kbest; if (dst < 0) dst += M;
      o_r[dst] = v;
      lo = fmin(lo, v); hi = fmax(hi, v);
    }
  }
  for (int off = 32; off > 0; off >>= 1) { lo = fmin(lo, __shfl_down(lo, off)); hi = fmax(hi, __shfl_down(hi, off)); }
  if (lane == 0) { atomicMin(&mm_enc[2 * b], enc_f64(lo)); atomicMin(&mm_enc[2 * b + 1], ~enc_f64(hi)); }
  (void)B;
}

// The sklearn MinMaxScaler arithmetic X * scale_ + min_ over the image (its minimum / maximum: k_anp_rows, order-preserving encoded
// atomics); 64 workgroups per humerus keep small batches busy.
__global__ void __launch_bounds__(256)
k_anp_scale(const double* __restrict__ raw, const unsigned long long* __restrict__ mm_enc, float* __restrict__ image) {
  const int b = blockIdx.y, tid = threadIdx.x;
  const double lo = dec_f64(mm_enc[2 * b]), hi = dec_f64(~mm_enc[2 * b + 1]);
  double rng = hi - lo;
  if (rng == 0.0) rng = 1.0;
  const double sc = 1.0 / rng, mn = 0.0 - lo * sc;
  const double* x = raw + (size_t)b * SH_IMG;
  float* o = image + (size_t)b * SH_IMG;
  for (int i = blockIdx.x * 256 + tid; i < SH_IMG; i += gridDim.x * 256) o[i] = (float)(x[i] * sc + mn);
}

// Edge pixels of the mask in two launches over (8 image rows, humerus) workgroups, one wave per row (coalesced 64-pixel reads,
// edges by ballot): counts per row first, then every row finds its place in the row-major point list from the counts of the
// rows before it (one block per humerus kept 192 CUs idle for 0.3 ms on the critical path behind the network).
__global__ void __launch_bounds__(512)
k_anp_edge_count(const float* __restrict__ logits, int* __restrict__ rowcnt /*[B][SH_ANP_ROWS][2]: mask changes, mask pixels*/,
                 unsigned long long* __restrict__ maskbits /*[B][SH_ANP_ROWS][8]: the mask (logit > 0), one bit per pixel -- k_anp_edges and
                 k_sphere_partial read these 2 MB instead of the 67 MB of logits again*/) {
  const int b = blockIdx.y, lane = threadIdx.x & 63, i = blockIdx.x * 8 + (threadIdx.x >> 6);
  const int M = SH_MPROX;
  const float* lg = logits + ((size_t)b * SH_ANP_ROWS + i) * M;
  float v[SH_MPROX / 64];
#pragma unroll
  for (int c = 0; c < SH_MPROX / 64; ++c) v[c] = lg[c * 64 + lane];      // the whole row in flight at once
  int ne = 0, nm = 0, carry = 0;
#pragma unroll
  for (int c = 0; c < SH_MPROX / 64; ++c) {
    const int m = v[c] > 0.0f ? 1 : 0;
    int prev = __shfl_up(m, 1);
    if (lane == 0) prev = carry;
    ne += __popcll(__ballot(m != prev));      // np.diff(mask, prepend=0) != 0
    const unsigned long long bm = __ballot(m);
    nm += __popcll(bm);
    if (lane == 0) maskbits[((size_t)b * SH_ANP_ROWS + i) * (SH_MPROX / 64) + c] = bm;
    carry = __shfl(m, 63);
  }
  if (lane == 0) { rowcnt[2 * ((size_t)b * SH_ANP_ROWS + i)] = ne; rowcnt[2 * ((size_t)b * SH_ANP_ROWS + i) + 1] = nm; }
}

__global__ void __launch_bounds__(512)
k_anp_edges(const unsigned long long* __restrict__ maskbits, const double* __restrict__ raw, const double* __restrict__ t01, const int* __restrict__ roll,
            const double* __restrict__ prox_zs, const int* __restrict__ rowcnt, double* __restrict__ pts_obb, int* __restrict__ counts,
            int* __restrict__ err) {
  static_assert(SH_ANP_ROWS == 512, "8 row counts per lane");
  const int b = blockIdx.y, lane = threadIdx.x & 63, i = blockIdx.x * 8 + (threadIdx.x >> 6);
  const int M = SH_MPROX;
  // this row's first slot = edges of the rows before it; the whole image's totals for the first row's wave
  const int2* rc = (const int2*)(rowcnt + 2 * (size_t)b * SH_ANP_ROWS);
  int before = 0, tot = 0, totm = 0;
#pragma unroll
  for (int q = 0; q < SH_ANP_ROWS / 64; ++q) {
    const int r = q * 64 + lane;
    const int2 c2 = rc[r];
    before += r < i ? c2.x : 0; tot += c2.x; totm += c2.y;
  }
  for (int off = 32; off > 0; off >>= 1) { before += __shfl_xor(before, off); tot += __shfl_xor(tot, off); totm += __shfl_xor(totm, off); }
  if (i == 0 && lane == 0) {
    counts[2 * b] = tot;
    counts[2 * b + 1] = totm;
    if (tot > SH_ANP_CAP) atomicExch(&err[b], SH_ERR_CAPACITY_DEV);
    if (tot < 6) atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV);
  }
  // the edge pixels as points (r cos t, r sin t, z), in row-major order
  const unsigned long long* mb = maskbits + ((size_t)b * SH_ANP_ROWS + i) * (SH_MPROX / 64);
  const double t0 = t01[2 * ((size_t)b * SH_ANP_ROWS + i)], t1 = t01[2 * ((size_t)b * SH_ANP_ROWS + i) + 1];
  const int kroll = roll[(size_t)b * SH_ANP_ROWS + i];
  const double* r = raw + ((size_t)b * SH_ANP_ROWS + i) * M;
  const double z = prox_zs[(size_t)b * SH_NPROX + SH_ANP_ROW0 + i];
  int o = before, carry = 0;
  unsigned long long v[SH_MPROX / 64];
#pragma unroll
  for (int c = 0; c < SH_MPROX / 64; ++c) v[c] = mb[c];
#pragma unroll
  for (int c = 0; c < SH_MPROX / 64; ++c) {
    const int j = c * 64 + lane;
    const int m = (int)(v[c] >> lane & 1ull);
    int prev = __shfl_up(m, 1);
    if (lane == 0) prev = carry;
    const unsigned long long eb = __ballot(m != prev);
    if (m != prev) {
      const int pos = o + __popcll(eb & ((1ull << lane) - 1ull));
      if (pos < SH_ANP_CAP) {
        double* p = pts_obb + ((size_t)b * SH_ANP_CAP + pos) * 3;
        int js = j + kroll; if (js >= M) js -= M;      // pixel j of the rolled row is sample js of the row's grid (k_anp_rows)
        const double tj = linspace_at(t0, t1, M, js);
        p[0] = r[j] * cos(tj);
        p[1] = r[j] * sin(tj);
        p[2] = z;
      }
    }
    o += __popcll(eb);
    carry = __shfl(m, 63);
  }
}

// deterministic block sum of NV values per thread (fixed tree)
template <int NV>
__device__ inline void block_sum(double* v, double* sh /*[NV][blockDim/64]*/, int tid, int nw) {
  for (int k = 0; k < NV; ++k) {
    double x = v[k];
    for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off);
    if ((tid & 63) == 0) sh[k * nw + (tid >> 6)] = x;
  }
  __syncthreads();
  for (int k = 0; k < NV; ++k) {
    double x = 0.0;
    for (int w = 0; w < nw; ++w) x += sh[k * nw + w];
    v[k] = x;
  }
  __syncthreads();
}

// Plane.best_fit + LsqEllipse centre (anatomic_neck.py:123-153); one block of 256 per humerus
__global__ void __launch_bounds__(256)
k_anp_plane(const double* __restrict__ pts_obb, const int* __restrict__ counts, double* __restrict__ plane /*[B][6]*/, int* __restrict__ err,
            unsigned long long* __restrict__ ray_t /*[B][4]: "no hit yet" for k_rays_hit (was a fill launch)*/) {
  __shared__ double sh[21 * 4];
  int b = blockIdx.x, tid = threadIdx.x;
  if (tid < 4) ray_t[4 * b + tid] = ~0ull;
  int K = counts[2 * b];
  if (K > SH_ANP_CAP) K = SH_ANP_CAP;
  const double* P = pts_obb + (size_t)b * SH_ANP_CAP * 3;
  double* out = plane + 6 * b;
  if (K < 6) { if (tid == 0) for (int k = 0; k < 6; ++k) out[k] = 0.0; return; }
  double s[3] = {0, 0, 0};
  for (int i = tid; i < K; i += 256) { s[0] += P[3 * i]; s[1] += P[3 * i + 1]; s[2] += P[3 * i + 2]; }
  block_sum<3>(s, sh, tid, 4);
  double c[3] = {s[0] / K, s[1] / K, s[2] / K};
  double m[6] = {0, 0, 0, 0, 0, 0};
  for (int i = tid; i < K; i += 256) {
    double x = P[3 * i] - c[0], y = P[3 * i + 1] - c[1], z = P[3 * i + 2] - c[2];
    m[0] += x * x; m[1] += x * y; m[2] += x * z; m[3] += y * y; m[4] += y * z; m[5] += z * z;
  }
  block_sum<6>(m, sh, tid, 4);
  double C[9] = {m[0], m[1], m[2], m[1], m[3], m[4], m[2], m[4], m[5]};
  double w[3], V[9];
  eig_sym3(C, w, V);
  double nrm[3] = {V[0], V[3], V[6]};        // smallest eigenvalue = plane normal
  if (nrm[2] < 0) { nrm[0] = -nrm[0]; nrm[1] = -nrm[1]; nrm[2] = -nrm[2]; }
  double u[3], v[3];
  plane_basis(nrm, u, v);
  // 6x6 scatter of [x^2, xy, y^2, x, y, 1] in plane coordinates about c
  double S[21];
  for (int k = 0; k < 21; ++k) S[k] = 0.0;
  for (int i = tid; i < K; i += 256) {
    double rel[3] = {P[3 * i] - c[0], P[3 * i + 1] - c[1], P[3 * i + 2] - c[2]};
    double x = dot3(rel, u), y = dot3(rel, v);
    double d[6] = {x * x, x * y, y * y, x, y, 1.0};
    int q = 0;
    for (int a = 0; a < 6; ++a)
      for (int e = a; e < 6; ++e) S[q++] += d[a] * d[e];
  }
  block_sum<21>(S, sh, tid, 4);
  if (tid == 0) {
    double S6[36];
    int q = 0;
    for (int a = 0; a < 6; ++a)
      for (int e = a; e < 6; ++e) { S6[a * 6 + e] = S[q]; S6[e * 6 + a] = S[q]; ++q; }
    double ex, ey;
    if (!ellipse_center_from_scatter(S6, &ex, &ey)) { atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV); ex = ey = 0.0; }
    for (int k = 0; k < 3; ++k) { out[k] = c[k] + ex * u[k] + ey * v[k]; out[3 + k] = nrm[k]; }
  }
}

// K23: rays from the plane point: 0 = +normal, 1 = -normal, 2 = +central, 3 = -central
// (central = normal with z zeroed, renormalised); Moller-Trumbore in fp64 against all triangles, nearest forward hit.
// k_rays_hit: (chunk of triangles, humerus) workgroups, every triangle fetched once for the four rays, nearest t per ray by
// atomicMin on its bit pattern (t > 0: patterns order like values; ~0 = no hit).  k_rays: the four end points per humerus.
#define SH_RAY_CHUNKS 16
__device__ inline void ray_dir(const double* pl, int ray, double* d) {
  d[0] = pl[3]; d[1] = pl[4]; d[2] = pl[5];
  if (d[2] < 0) { d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
  if (ray >= 2) { d[2] = 0.0; double n = norm3(d); d[0] /= n; d[1] /= n; d[2] /= n; }
  if (ray & 1) { d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
}
__global__ void __launch_bounds__(256)
k_rays_hit(const double* __restrict__ vobb, const int* __restrict__ faces, const long long* __restrict__ voff,
           const long long* __restrict__ foff, const double* __restrict__ plane, unsigned long long* __restrict__ tmin /*[B][4], ~0*/) {
  const int b = blockIdx.y, tid = threadIdx.x;
  const double* pl = plane + 6 * b;
  const double o[3] = {pl[0], pl[1], pl[2]};
  double d[4][3];
#pragma unroll
  for (int ray = 0; ray < 4; ++ray) ray_dir(pl, ray, d[ray]);
  const double* vb = vobb + 3 * voff[b];
  const int* fb = faces + 3 * foff[b];
  const long long nf = foff[b + 1] - foff[b];
  double tbest[4] = {1e300, 1e300, 1e300, 1e300};
  for (long long f = blockIdx.x * 256 + tid; f < nf; f += (long long)gridDim.x * 256) {
    const double* A = vb + 3 * (size_t)fb[3 * f];
    const double* Bv = vb + 3 * (size_t)fb[3 * f + 1];
    const double* Cv = vb + 3 * (size_t)fb[3 * f + 2];
    const double e1[3] = {Bv[0] - A[0], Bv[1] - A[1], Bv[2] - A[2]};
    const double e2[3] = {Cv[0] - A[0], Cv[1] - A[1], Cv[2] - A[2]};
    const double tv[3] = {o[0] - A[0], o[1] - A[1], o[2] - A[2]};
    double qv[3];
    cross3(tv, e1, qv);
#pragma unroll
    for (int ray = 0; ray < 4; ++ray) {
      double pv[3];
      cross3(d[ray], e2, pv);
      const double det = dot3(e1, pv);
      if (!(fabs(det) > 1e-12)) continue;
      const double inv = 1.0 / det;
      const double uu = dot3(tv, pv) * inv;
      const double ww = dot3(d[ray], qv) * inv;
      const double t = dot3(e2, qv) * inv;
      if (uu >= 0 && ww >= 0 && uu + ww <= 1 && t > 1e-9 && t < tbest[ray]) tbest[ray] = t;
    }
  }
#pragma unroll
  for (int ray = 0; ray < 4; ++ray) {
    double t = tbest[ray];
    for (int off = 32; off > 0; off >>= 1) t = fmin(t, __shfl_down(t, off));
    if ((tid & 63) == 0 && t < 1e299) atomicMin(&tmin[4 * b + ray], (unsigned long long)__double_as_longlong(t));
  }
}
// the point of ray `ray` (0..3: +-normal, +-central) of humerus b from its nearest hit parameter; one lane per ray (k_tail, k_te.h)
__device__ inline void rays_point(const double* plane, const unsigned long long* tmin, double* axes_obb, int* err, int b, int ray) {
  const int g = 4 * b + ray;
  const double* pl = plane + 6 * b;
  double d[3];
  ray_dir(pl, ray, d);
  double* a = axes_obb + (size_t)g * 3;
  const unsigned long long e = tmin[g];
  if (e == ~0ull) { atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV); a[0] = a[1] = a[2] = 0.0; }
  else { const double t = __longlong_as_double((long long)e); for (int k = 0; k < 3; ++k) a[k] = pl[k] + d[k] * t; }
}

}  // namespace sh
