// k_groove.h -- bicipital groove on the device (reference src/shoulder/humerus/bicipital_groove.py).
//   k_groove_rows     :102-156  per-row savgol / find_peaks / features   (one lane per slice row)
//   k_groove_scale    :156      StandardScaler statistics                (one lane per feature)
//   k_groove_rfc      :174-185  random forest P(class 1) per peak        (one lane per peak)
//   k_groove_kde      :184-188  linear-kernel KDE argmax -> bg_theta     (one block per humerus)
//   k_groove_localmin :192-232  per-row local radius minimum -> points   (one lane per slice row)
//   k_groove_axis     :244-265  line fit (wave covariance + power iteration), B-4 orientation
// Buffers: groove.xraw [B][330*7][9] f64, groove.ptheta [B][330*7] f64, groove.npk [B][330] i32,
// groove.r0 [B][330][512] f64, groove.stats [B][18] f64 (mean, scale), groove.proba [B][330*7] f32,
// groove.bg_theta [B] f64, groove.local_idx [B][330] i32, groove.points_obb [B][330][3] f64.
#pragma once
#include "k_stages.h"

namespace sh {

#define SH_GSLOTS (SH_GROOVE_NROWS * SH_MAXPEAK)

__global__ void k_groove_rows(const double* __restrict__ itr_cs /*[B][600][2][512]*/, const double* __restrict__ prox_zs,
                              const double* __restrict__ canal_axis_ct, int row0,
                              double* __restrict__ xraw, double* __restrict__ ptheta, int* __restrict__ npk,
                              double* __restrict__ r0, int* __restrict__ err, int B) {
  // One wave per slice row; the row lives in LDS.  Same arithmetic as sh::groove_row_features
  // (sh_scalar.h, host-tested): NumPy's pairwise mean is reproduced with lanes as the 8 x 4 partial
  // accumulators, every filter tap sum / plateau walk / prominence scan is the sequential routine
  // applied per sample or per peak.
  // two row buffers serve the four stages (LDS per wave decides how many rows a CU works on at once): r, then the filtered
  // row in A; r - mean (negated), then the rolled filtered row in B
  __shared__ double s_bufA[SH_MPROX], s_bufB[SH_MPROX];
  double* const s_r = s_bufA; double* const s_filt = s_bufA;
  double* const s_neg = s_bufB; double* const s_roll = s_bufB;
  __shared__ int s_cand[SH_PEAK_CAP];
  __shared__ Peak s_pk[SH_PEAK_CAP];
  __shared__ int s_pass[SH_PEAK_CAP];
  __shared__ int s_ncand;
  const int M = SH_MPROX;
  const int gid = blockIdx.x, lane = threadIdx.x;
  const int b = gid / SH_GROOVE_NROWS, i = gid % SH_GROOVE_NROWS;
  const double* row = itr_cs + ((size_t)b * SH_NPROX + row0 + i) * 2 * M;
  const double* zs = prox_zs + (size_t)b * SH_NPROX + row0;
  for (int k = lane; k < M; k += 64) s_r[k] = row[M + k];
  if (lane == 0) s_ncand = 0;
  __syncthreads();
  // np.mean(r): pairwise_sum(512) = (leaf0 + leaf1) + (leaf2 + leaf3), leaf = 8 strided accumulators
  double acc = 0.0;
  if (lane < 32) {
    const int leaf = lane >> 3, k = lane & 7;
    acc = s_r[128 * leaf + k];
    for (int q = 1; q < 16; ++q) acc += s_r[128 * leaf + 8 * q + k];
  }
  acc = acc + __shfl_down(acc, 1);      // (r0+r1) at k=0, (r2+r3) at k=2, ...
  acc = acc + __shfl_down(acc, 2);      // ((r0+r1)+(r2+r3)) at k=0, ((r4+r5)+(r6+r7)) at k=4
  acc = acc + __shfl_down(acc, 4);      // leaf sum at k=0
  acc = acc + __shfl_down(acc, 8);      // leaf0+leaf1 at lane 0, leaf2+leaf3 at lane 16
  acc = acc + __shfl_down(acc, 16);
  const double mean = __shfl(acc, 0) / (double)M;
  for (int k = lane; k < M; k += 64) s_neg[k] = -1.0 * (s_r[k] - mean);
  __syncthreads();
  for (int k = 5 + lane; k < M - 5; k += 64) s_filt[k] = savgol10_1_at(s_neg, k);
  if (lane == 0) savgol10_1_edges(s_neg, M, s_filt);
  __syncthreads();
  // first argmin of the filtered row
  double bv = 1e300;
  int bi = 0x7fffffff;
  for (int k = lane; k < M; k += 64) if (s_filt[k] < bv) { bv = s_filt[k]; bi = k; }
  for (int off = 32; off > 0; off >>= 1) {
    double ov = __shfl_down(bv, off);
    int oi = __shfl_down(bi, off);
    if (ov < bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
  }
  const int amin = __shfl(bi, 0);
  double* r0row = r0 + (size_t)gid * M;
  for (int k = lane; k < M; k += 64) r0row[k] = -s_neg[k];      // polar_0 radius = r - mean(r)
  __syncthreads();      // (the rolled row takes s_neg's place)
  for (int k = lane; k < M; k += 64) { int src = k + amin; if (src >= M) src -= M; s_roll[k] = s_filt[src]; }
  __syncthreads();
  // local maxima: every rising edge is examined independently (equivalent to the sequential scan)
  for (int k = 1 + lane; k < M - 1; k += 64)
    if (s_roll[k - 1] < s_roll[k]) {
      int resume;
      int pk = local_maximum_at(s_roll, M, k, &resume);
      if (pk >= 0) { int s = atomicAdd(&s_ncand, 1); if (s < SH_PEAK_CAP) s_cand[s] = pk; }
    }
  __syncthreads();
  int nc = s_ncand < SH_PEAK_CAP ? s_ncand : SH_PEAK_CAP;
  if (lane == 0)      // ascending index
    for (int a = 1; a < nc; ++a) { int v = s_cand[a]; int q = a - 1; while (q >= 0 && s_cand[q] > v) { s_cand[q + 1] = s_cand[q]; --q; } s_cand[q + 1] = v; }
  __syncthreads();
  static_assert(SH_MPROX == 512, "8 samples per lane");
  // sh::peak_eval for every candidate.  Its prominence walks (left and right until a higher sample, up to the whole row
  // for the highest peak, one dependent LDS read per step) are done by the whole wave per candidate instead: every lane
  // holds 8 consecutive samples; nearest higher sample, minimum of the stretch up to it and the occurrence of that minimum
  // closest to the peak are wave reductions -- the same values and indices as the sequential walk.
  __shared__ int s_lb[SH_PEAK_CAP], s_rb[SH_PEAK_CAP];
  __shared__ double s_lmin[SH_PEAK_CAP], s_rmin[SH_PEAK_CAP];
  {
    double xr[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) xr[q] = s_roll[8 * lane + q];
    for (int cnd = 0; cnd < nc; ++cnd) {
      const int pk = s_cand[cnd];
      const double v = s_roll[pk];
      int L = -1, R = M;
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int k = 8 * lane + q; if (k < pk && xr[q] > v) L = k; }
#pragma unroll
      for (int q = 7; q >= 0; --q) { const int k = 8 * lane + q; if (k > pk && xr[q] > v) R = k; }
      for (int off = 32; off > 0; off >>= 1) { L = max(L, __shfl_xor(L, off)); R = min(R, __shfl_xor(R, off)); }
      double lm = v, rm = v;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const int k = 8 * lane + q;
        if (k > L && k <= pk) lm = fmin(lm, xr[q]);
        if (k >= pk && k < R) rm = fmin(rm, xr[q]);
      }
      for (int off = 32; off > 0; off >>= 1) { lm = fmin(lm, __shfl_xor(lm, off)); rm = fmin(rm, __shfl_xor(rm, off)); }
      int lb = -1, rb = M;
#pragma unroll
      for (int q = 0; q < 8; ++q) { const int k = 8 * lane + q; if (k > L && k <= pk && xr[q] == lm) lb = k; }
#pragma unroll
      for (int q = 7; q >= 0; --q) { const int k = 8 * lane + q; if (k >= pk && k < R && xr[q] == rm) rb = k; }
      for (int off = 32; off > 0; off >>= 1) { lb = max(lb, __shfl_xor(lb, off)); rb = min(rb, __shfl_xor(rb, off)); }
      if (lane == 0) { s_lb[cnd] = lb; s_rb[cnd] = rb; s_lmin[cnd] = lm; s_rmin[cnd] = rm; }
    }
  }
  __syncthreads();
  if (lane < nc) {      // heights, prominence threshold, widths at half prominence (short walks): as in sh::peak_eval
    const double* x = s_roll;
    const int pk = s_cand[lane], lb = s_lb[lane], rb = s_rb[lane];
    const double left_min = s_lmin[lane], right_min = s_rmin[lane];
    bool ok = x[pk] >= -10.0;
    const double prom = x[pk] - (left_min > right_min ? left_min : right_min);
    ok = ok && prom >= 0.6;
    if (ok) {
      const double h = x[pk] - prom * 0.5;
      int k = pk;
      while (lb < k && h < x[k]) --k;
      double lip = (double)k;
      if (x[k] < h) lip += (h - x[k]) / (x[k + 1] - x[k]);
      k = pk;
      while (k < rb && h < x[k]) ++k;
      double rip = (double)k;
      if (x[k] < h) rip -= (h - x[k]) / (x[k - 1] - x[k]);
      const double w = rip - lip;
      ok = w >= 0.1;
      if (ok) { s_pk[lane].idx = pk; s_pk[lane].prominence = prom; s_pk[lane].width = w; s_pk[lane].width_height = h; }
    }
    s_pass[lane] = ok ? 1 : 0;
  }
  __syncthreads();
  // Second half of the row (sh::groove_features_from_peaks, same arithmetic per element) spread over the wave: the
  // 7 x 7 wrapped angle differences (atan2 / sin / cos each) take one lane per pair instead of 49 serial evaluations.
  __shared__ int s_np;
  __shared__ double s_th[SH_MAXPEAK], s_d[SH_MAXPEAK][SH_MAXPEAK];
  __shared__ int s_pidx[SH_MAXPEAK];
  if (lane == 0) {
    int np_ = 0;
    for (int a = 0; a < nc; ++a) if (s_pass[a]) s_pk[np_++] = s_pk[a];
    if (np_ > SH_MAXPEAK) {      // keep the 7 most prominent (B-5: ascending index order among the kept)
      bool keep[SH_PEAK_CAP];
      for (int k = 0; k < np_; ++k) keep[k] = false;
      for (int q = 0; q < SH_MAXPEAK; ++q) {
        int bq = -1;
        for (int k = 0; k < np_; ++k)
          if (!keep[k] && (bq < 0 || s_pk[k].prominence > s_pk[bq].prominence)) bq = k;
        keep[bq] = true;
      }
      int w = 0;
      for (int k = 0; k < np_; ++k) if (keep[k]) s_pk[w++] = s_pk[k];
      np_ = SH_MAXPEAK;
    }
    s_np = np_;
  }
  __syncthreads();
  const int np_ = s_np;
  if (lane < np_) {
    const int idx = (s_pk[lane].idx + amin) % M;     // (peaks - rmin) % interp_num with rmin = -amin
    s_pidx[lane] = idx;
    s_th[lane] = row[idx];
  }
  // MinMaxScaler over the cut zs (bicipital_groove.py:89): X*scale_ + min_
  double zmax = -1e300, zmin = 1e300;
  for (int k = lane; k < SH_GROOVE_NROWS; k += 64) { zmax = fmax(zmax, zs[k]); zmin = fmin(zmin, zs[k]); }
  for (int off = 32; off > 0; off >>= 1) { zmax = fmax(zmax, __shfl_xor(zmax, off)); zmin = fmin(zmin, __shfl_xor(zmin, off)); }
  __syncthreads();
  {
    const int k = lane >> 3, j = lane & 7;
    if (k < np_ && j < np_) s_d[k][j] = wrapped_abs_diff(s_th[k], s_th[j]);
  }
  __syncthreads();
  if (lane < np_) {
    const int k = lane;
    double near = 0.0, next = 0.0;
    if (np_ > 1) {
      // sorted wrapped distances to all peaks, dropping those that round to 0.00 (:46)
      double a[SH_MAXPEAK];
      int na = 0;
      for (int j = 0; j < np_; ++j) {
        double d = s_d[k][j];
        if (rint(d * 100.0) / 100.0 != 0.0) a[na++] = d;     // np.round(angs, 2) != 0
      }
      for (int p = 1; p < na; ++p) { double v = a[p]; int q = p - 1; while (q >= 0 && a[q] > v) { a[q + 1] = a[q]; --q; } a[q + 1] = v; }
      near = na > 0 ? a[0] : nan("");
      if (np_ > 2) next = na > 1 ? a[1] : nan("");
    }
    double rng = zmax - zmin;
    if (rng == 0.0) rng = 1.0;
    const double sc = 1.0 / rng;
    const double z_scaled = zs[i] * sc + (0.0 - zmin * sc);
    const double* ax = canal_axis_ct + 6 * b;
    double cu[3] = {ax[0] - ax[3], ax[1] - ax[4], ax[2] - ax[5]};
    const double n = norm3(cu);
    cu[0] /= n; cu[1] /= n; cu[2] /= n;
    const double z = zs[i], th = s_th[k];
    const double rad = row[M + s_pidx[k]];
    const double px = rad * cos(th), py = rad * sin(th);
    const double dx = px - cu[0] * z, dy = py - cu[1] * z;
    double* X = xraw + ((size_t)b * SH_GSLOTS + (size_t)i * SH_MAXPEAK + k) * 9;
    X[0] = rad; X[1] = near; X[2] = next; X[3] = z_scaled; X[4] = s_pk[k].prominence; X[5] = s_pk[k].width;
    X[6] = s_pk[k].width_height; X[7] = sqrt(dx * dx + dy * dy); X[8] = (double)np_ / 7.0;
    ptheta[(size_t)b * SH_GSLOTS + (size_t)i * SH_MAXPEAK + k] = th;
    // a NaN feature = the reference's IndexError (all other peaks within 0.005 rad)
    for (int q = 0; q < 9; ++q) if (X[q] != X[q]) atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV);
  }
  if (lane == 0) npk[gid] = np_;
  (void)B;
}

// sklearn StandardScaler: mean over rows, population variance, scale = sqrt(var) (1 if ~0)
__global__ void __launch_bounds__(256)
k_groove_scale(const double* __restrict__ xraw, const int* __restrict__ npk, double* __restrict__ stats, int B,
               int* __restrict__ slots_out /*[B][SH_GSLOTS]: the occupied peak slots of a humerus in row order*/, int* __restrict__ nslot_out /*[B]*/,
               float* __restrict__ proba /*[B][SH_GSLOTS]: -1 in the slots that hold no peak*/) {
  // The sums run over the peaks in row order, one after the other (the order the oracle's column reduction uses), so
  // they stay sequential per feature; what is parallel is the staging: the valid rows are compacted into LDS first
  // (five, then four feature columns), the nine lanes then add from LDS instead of waiting on one global load per term.
  __shared__ double col[5][SH_GSLOTS];
  __shared__ int slot[SH_GSLOTS];
  __shared__ int P_s;
  const int b = blockIdx.x, tid = threadIdx.x;
  const double* X = xraw + (size_t)b * SH_GSLOTS * 9;
  const int* np_ = npk + (size_t)b * SH_GROOVE_NROWS;
  // slot table in row order: every row finds its first position from the counts of the rows before it (one lane walking
  // 330 dependent global loads was a third of the kernel)
  __shared__ int s_cnt[SH_GROOVE_NROWS];
  for (int i = tid; i < SH_GROOVE_NROWS; i += 256) s_cnt[i] = np_[i];
  __syncthreads();
  for (int i = tid; i < SH_GROOVE_NROWS; i += 256) {
    int off = 0;
    for (int j = 0; j < i; ++j) off += s_cnt[j];
    const int n = s_cnt[i];
    for (int k = 0; k < n; ++k) slot[off + k] = i * SH_MAXPEAK + k;
    if (i == SH_GROOVE_NROWS - 1) P_s = off + n;
  }
  __syncthreads();
  const int P = P_s;
  // the forest walks the occupied slots only (k_groove_rfc): a lane per SLOT left 60 % of its lanes without a peak -- beside a UNet pass,
  // where the chain has 32 CUs, the kernel is bound by how many waves are resident, not by one wave's latency
  for (int p = tid; p < P; p += 256) slots_out[(size_t)b * SH_GSLOTS + p] = slot[p];
  if (tid == 0) nslot_out[b] = P;
  for (int s = tid; s < SH_GSLOTS; s += 256)
    if (s % SH_MAXPEAK >= s_cnt[s / SH_MAXPEAK]) proba[(size_t)b * SH_GSLOTS + s] = -1.0f;
  for (int f0 = 0; f0 < 9; f0 += 5) {
    const int nf = f0 + 5 <= 9 ? 5 : 9 - f0;
    for (int q = tid; q < P * nf; q += 256) { const int p = q / nf, f = q - p * nf; col[f][p] = X[(size_t)slot[p] * 9 + f0 + f]; }
    __syncthreads();
    if (tid < nf) {
      const double* c = col[tid];
      // (the order of the additions is the oracle's; the operands are fetched eight at a time so that the chain of adds does
      //  not wait for an LDS round trip per term)
      double s = 0.0;
      int p = 0;
      for (; p + 8 <= P; p += 8) {
        double t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = c[p + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t8[u];
      }
      for (; p < P; ++p) s += c[p];
      const double mean = P ? s / (double)P : 0.0;
      double v = 0.0;
      p = 0;
      for (; p + 8 <= P; p += 8) {
        double t8[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t8[u] = c[p + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) { const double d = t8[u] - mean; v += d * d; }
      }
      for (; p < P; ++p) { double d = c[p] - mean; v += d * d; }
      const double var = P ? v / (double)P : 0.0;
      // sklearn StandardScaler (1.6): a feature is "constant" -- scale 1 -- when its variance is within the error bound of the
      // two-pass algorithm, var <= n eps var + (n mean eps)^2 (`_is_constant_feature`), not only when it is exactly zero.
      // A feature that is constant in exact arithmetic (the cut humerus has one) comes out of a similarity copy with a
      // standard deviation of ~1e-13: dividing by that turned the column into +-1 noise and moved bg_theta (randomized
      // sweep of the proximal path: 8 of 24 copies).
      const double eps_ = 2.220446049250313e-16, nn = (double)P;
      const double ub = nn * eps_ * var + (nn * mean * eps_) * (nn * mean * eps_);
      double scale = var <= ub ? 1.0 : sqrt(var);
      stats[(size_t)b * 18 + f0 + tid] = mean;
      stats[(size_t)b * 18 + 9 + f0 + tid] = scale;
    }
    __syncthreads();
  }
}

// Forest tables re-packed for the walk: one 16-byte node {feature, threshold bits (leaf: leaf weight bits), true, false} per
// load instead of four dependent 4-byte loads (done per run: the parameter block may have been re-broadcast)
__global__ void k_rfc_pack(const int* __restrict__ feat, const float* __restrict__ thr, const int* __restrict__ ti, const int* __restrict__ fi,
                           const float* __restrict__ lw, int4* __restrict__ nodes, int N) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N) return;
  const bool leaf = ti[i] < 0;
  nodes[i] = make_int4(feat[i], __float_as_int(leaf ? lw[i] : thr[i]), ti[i], fi[i]);
}

// sh::rfc_proba1 (onnxruntime TreeEnsembleClassifier walk, bicipital_groove.py:178-181) on the packed nodes; the nine
// scaled features of a lane sit in LDS so that the per-node feature pick is one ds_read
__global__ void __launch_bounds__(64)
k_groove_rfc(const double* __restrict__ xraw, const int* __restrict__ slots /*[B][SH_GSLOTS]*/, const int* __restrict__ nslot /*[B]*/, const double* __restrict__ stats,
             const int4* __restrict__ nodes, const int* __restrict__ roots, int n_trees,
             double* __restrict__ xs /*[B][slots][9] scaled*/, float* __restrict__ proba, int B) {
  __shared__ double sx[9][64];
  const int tid = threadIdx.x;
  // workgroup = 64 consecutive entries of one humerus' list of occupied slots (k_groove_scale); the workgroups beyond a list end at once
  constexpr int WPB = (SH_GSLOTS + 63) / 64;
  const int b = blockIdx.x / WPB, p = (blockIdx.x - b * WPB) * 64 + tid;
  if (b >= B || p >= nslot[b]) return;
  const int gid = b * SH_GSLOTS + slots[(size_t)b * SH_GSLOTS + p];
  for (int f = 0; f < 9; ++f) {
    const double v = (xraw[(size_t)gid * 9 + f] - stats[(size_t)b * 18 + f]) / stats[(size_t)b * 18 + 9 + f];
    xs[(size_t)gid * 9 + f] = v;
    sx[f][tid] = v;
  }
  double acc = 0.0;
  for (int t = 0; t < n_trees; ++t) {
    int4 nd = nodes[roots[t]];
    while (nd.z >= 0) nd = nodes[(sx[nd.x][tid] <= (double)__int_as_float(nd.y)) ? nd.z : nd.w];
    acc += (double)__int_as_float(nd.y);
  }
  proba[gid] = (float)acc;
}

// KernelDensity(kernel="linear", bandwidth=1): rho(t) ~ sum_i max(0, 1 - |t - theta_i|) over the
// peaks with P > 0.4; bg_theta = argmax over linspace(-pi, pi, 1024).
// Canonical rule B-8: rho is piecewise linear, and wherever as many selected peaks lie within the bandwidth on one side of t
// as on the other it is exactly FLAT -- a maximum on such a stretch spans every grid point between two neighbouring peaks, and
// which of them `np.argmax` returns in the reference is decided by the rounding noise of sklearn's tree-ordered log-sum-exp
// (found by a randomized sweep: one similarity copy in ~140 had a plateau 11 grid points wide).  Grid points within 1e-9
// (relative) of the maximum count as tied and the LOWEST index wins; off a plateau neighbouring grid values differ by >= 2e-5.
#define SH_KDE_TIE 1e-9
// (a device function of k_groove_tail: 256 lanes of the humerus' workgroup; returns bg_theta to every lane)
__device__ inline double groove_kde_wg(const double* __restrict__ ptheta, const float* __restrict__ proba, double* __restrict__ bg_theta,
                                       int* __restrict__ err, int b) {
  __shared__ double sel[SH_GSLOTS];
  __shared__ double dens[1024];
  __shared__ int nsel;
  __shared__ double wv[4];
  __shared__ int wi[4];
  __shared__ double s_bg;
  int tid = threadIdx.x;
  // the selected peaks in slot order (ordered compaction by ballots; one lane walking 2 310 dependent global loads was most
  // of the kernel)
  __shared__ int s_wc[4];
  if (tid == 0) nsel = 0;
  __syncthreads();
  for (int s0 = 0; s0 < SH_GSLOTS; s0 += 256) {
    const int s = s0 + tid;
    const bool keep = s < SH_GSLOTS && proba[(size_t)b * SH_GSLOTS + s] > 0.4f;
    const double val = keep ? ptheta[(size_t)b * SH_GSLOTS + s] : 0.0;
    const unsigned long long m = __ballot(keep);
    if ((tid & 63) == 0) s_wc[tid >> 6] = __popcll(m);
    __syncthreads();
    int pos = nsel + __popcll(m & ((1ull << (tid & 63)) - 1ull));
    for (int w = 0; w < (tid >> 6); ++w) pos += s_wc[w];
    if (keep) sel[pos] = val;
    __syncthreads();
    if (tid == 0) nsel += s_wc[0] + s_wc[1] + s_wc[2] + s_wc[3];
    __syncthreads();
  }
  const int n = nsel;
  if (n == 0) { if (tid == 0) { atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV); bg_theta[b] = 0.0; } return 0.0; }      // (uniform: every lane leaves)
  double best = -1.0;
  for (int j = tid; j < 1024; j += blockDim.x) {
    double t = linspace_at(-1.0 * M_PI, M_PI, 1024, j);
    double acc = 0.0;
    for (int i = 0; i < n; ++i) {
      double d = 1.0 - fabs(t - sel[i]);
      if (d > 0.0) acc += d;
    }
    dens[j] = acc;
    best = fmax(best, acc);
  }
  for (int off = 32; off > 0; off >>= 1) best = fmax(best, __shfl_down(best, off));
  if ((tid & 63) == 0) wv[tid >> 6] = best;
  __syncthreads();
  best = wv[0];
  for (int w = 1; w < (int)(blockDim.x >> 6); ++w) best = fmax(best, wv[w]);
  const double thr = best * (1.0 - SH_KDE_TIE);
  int bi = 0x7fffffff;
  for (int j = tid; j < 1024; j += blockDim.x)
    if (dens[j] >= thr) bi = min(bi, j);
  for (int off = 32; off > 0; off >>= 1) bi = min(bi, __shfl_down(bi, off));
  if ((tid & 63) == 0) wi[tid >> 6] = bi;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) bi = min(bi, wi[w]);
    s_bg = linspace_at(-1.0 * M_PI, M_PI, 1024, bi);
    bg_theta[b] = s_bg;
  }
  __syncthreads();
  return s_bg;
}

// row i of humerus b (one lane)
__device__ inline void groove_localmin_row(const double* __restrict__ itr_cs, const double* __restrict__ r0, const double* __restrict__ prox_zs,
                                           const double* __restrict__ prox_centroids, double bg, int row0, double deg_window,
                                           int* __restrict__ local_idx, double* pts_obb, int b, int i) {
  const int gid = b * SH_GROOVE_NROWS + i;
  const double* row = itr_cs + ((size_t)b * SH_NPROX + row0 + i) * 2 * SH_MPROX;
  int ivar = (int)rint(deg_window / (360.0 / (double)SH_MPROX));
  if (ivar < 1) ivar = 1;
  int loc = groove_local_min(row, r0 + (size_t)gid * SH_MPROX, SH_MPROX, bg, ivar);
  local_idx[gid] = loc;
  int k = loc < 0 ? loc + SH_MPROX : loc;      // python negative index
  k = k < 0 ? 0 : (k >= SH_MPROX ? SH_MPROX - 1 : k);      // never off the row (sh_set_params keeps the window within half a turn; the reference raises IndexError beyond)
  double t = row[k], r = row[SH_MPROX + k];
  const double* c = prox_centroids + 2 * ((size_t)b * SH_NPROX + row0 + i);
  double* p = pts_obb + (size_t)gid * 3;
  p[0] = r * cos(t) + c[0];
  p[1] = r * sin(t) + c[1];
  p[2] = prox_zs[(size_t)b * SH_NPROX + row0 + i] + 0.0;
}

// (one wave)
__device__ inline void groove_axis_wave(const double* pts_obb, const double* __restrict__ T_obb, double* __restrict__ axis_ct,
                                        double* __restrict__ pts_ct, int b) {
  int lane = threadIdx.x & 63;
  const double* P = pts_obb + (size_t)b * SH_GROOVE_NROWS * 3;
  double mean[3], d[3];
  wave_line_fit(P, SH_GROOVE_NROWS, 3, mean, d);
  double zmin = 1e300, zmax = -1e300;
  for (int i = lane; i < SH_GROOVE_NROWS; i += 64) { zmin = fmin(zmin, P[3 * i + 2]); zmax = fmax(zmax, P[3 * i + 2]); }
  for (int off = 32; off > 0; off >>= 1) { zmin = fmin(zmin, __shfl_down(zmin, off)); zmax = fmax(zmax, __shfl_down(zmax, off)); }
  zmin = __shfl(zmin, 0); zmax = __shfl(zmax, 0);
  double Ti[16];
  inv_transform(T_obb + 16 * b, Ti);
  for (int i = lane; i < SH_GROOVE_NROWS; i += 64) xform_pt(Ti, P[3 * i], P[3 * i + 1], P[3 * i + 2], pts_ct + ((size_t)b * SH_GROOVE_NROWS + i) * 3);
  if (lane == 0) {
    if (d[2] < 0) { d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }      // B-4
    double h = (zmax - zmin) / 2.0;
    xform_pt(Ti, mean[0] + d[0] * h, mean[1] + d[1] * h, mean[2] + d[2] * h, axis_ct + 6 * b);
    xform_pt(Ti, mean[0] - d[0] * h, mean[1] - d[1] * h, mean[2] - d[2] * h, axis_ct + 6 * b + 3);
  }
}

// bicipital_groove.py:184-265 behind the forest, per humerus, as ONE launch (were three: k_groove_kde, k_groove_localmin,
// k_groove_axis): KDE argmax -> bg_theta, the radius minimum of every row near it -> groove points, line fit -> axis.
__global__ void __launch_bounds__(256)
k_groove_tail(const double* __restrict__ ptheta, const float* __restrict__ proba, double* __restrict__ bg_theta, int* __restrict__ err,
              const double* __restrict__ itr_cs, const double* __restrict__ r0, const double* __restrict__ prox_zs, const double* __restrict__ prox_centroids,
              int row0, double deg_window, int* __restrict__ local_idx, double* pts_obb, const double* __restrict__ T_obb,
              double* __restrict__ axis_ct, double* __restrict__ pts_ct) {
  const int b = blockIdx.x, tid = threadIdx.x;
  const double bg = groove_kde_wg(ptheta, proba, bg_theta, err, b);
  for (int i = tid; i < SH_GROOVE_NROWS; i += 256) groove_localmin_row(itr_cs, r0, prox_zs, prox_centroids, bg, row0, deg_window, local_idx, pts_obb, b, i);
  __syncthreads();      // (the points of all rows, written by this workgroup, are read by its first wave)
  if (tid < 64) groove_axis_wave(pts_obb, T_obb, axis_ct, pts_ct, b);
}

}  // namespace sh
