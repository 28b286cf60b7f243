// sh_scalar.h -- the small sequential stages of the path, one call per humerus / per slice row.
// Written once as SH_HD functions: on the product path they run inside device kernels (one
// lane per item, data in HBM); tests instantiate the same source on the host
// (tests/hostcheck) to compare against the oracle without a GPU.
#pragma once
#include "sh_common.h"

namespace sh {

// ======================================================================================
// K10  ruptures.KernelCPD(kernel="rbf").fit(x).predict(n_bkps=1)   surgical_neck.py:31-33
// gamma = 1/median(pairwise sq. distances); K = exp(-clip(gamma d^2, 1e-2, 1e2));
// t* = first argmin_{t in [2, n-2]} c(0,t)+c(t,n),  c(a,b) = sum K_ii - sum_{ij} K_ij/(b-a).
// ======================================================================================
#define SH_CPD_MAXN 64
SH_HD double kth_smallest(double* a, int n, int k) {  // quickselect, destroys a
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    double p = a[(lo + hi) >> 1];
    int i = lo, j = hi;
    while (i <= j) {
      while (a[i] < p) ++i;
      while (a[j] > p) --j;
      if (i <= j) { double t = a[i]; a[i] = a[j]; a[j] = t; ++i; --j; }
    }
    if (k <= j) hi = j; else if (k >= i) lo = i; else break;
  }
  return a[k];
}

// gamma by the median heuristic; pd: scratch >= n*(n-1)/2 doubles
SH_HD double cpd_gamma(const double* x, int n, double* pd) {
  int np_ = n * (n - 1) / 2;
  int c = 0;
  for (int i = 0; i < n; ++i)
    for (int j = i + 1; j < n; ++j) { double d = x[i] - x[j]; pd[c++] = d * d; }
  double med;
  if (np_ & 1) med = kth_smallest(pd, np_, np_ / 2);
  else { double a = kth_smallest(pd, np_, np_ / 2 - 1); double b = kth_smallest(pd, np_, np_ / 2); med = (a + b) / 2.0; }
  return (med == 0.0) ? 1.0 : 1.0 / med;
}
SH_HD double cpd_kernel(double xi, double xj, double gamma) {
  double d = xi - xj;
  double v = d * d * gamma;
  v = v < 1e-2 ? 1e-2 : (v > 1e2 ? 1e2 : v);
  return exp(-v);
}
// c(0,t) + c(t,n) for the Gram matrix K (n x n)
SH_HD double cpd_cost(const double* K, int n, int t) {
  double d0 = 0, s0 = 0, d1 = 0, s1 = 0;
  for (int i = 0; i < t; ++i) { d0 += K[i * n + i]; for (int j = 0; j < t; ++j) s0 += K[i * n + j]; }
  for (int i = t; i < n; ++i) { d1 += K[i * n + i]; for (int j = t; j < n; ++j) s1 += K[i * n + j]; }
  return (d0 - s0 / (double)t) + (d1 - s1 / (double)(n - t));
}
// scratch: >= n*(n-1)/2 + n*n doubles
SH_HD int cpd_one_bkp(const double* x, int n, double* scratch) {
  const int min_size = 2;
  double* K = scratch + n * (n - 1) / 2;
  double gamma = cpd_gamma(x, n, scratch);
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < n; ++j) K[i * n + j] = cpd_kernel(x[i], x[j], gamma);
  int best_t = -1;
  double best = 1e300;
  for (int t = min_size; t <= n - min_size; ++t) {
    double cst = cpd_cost(K, n, t);
    if (cst < best) { best = cst; best_t = t; }
  }
  return best_t;
}

// ======================================================================================
// ProxObb canal range   mesh.py:181-190
// grad = np.gradient(savgol_filter(area, 3, 1)); canal = longest run (first of the longest) of consecutive sections
// with grad < 10.  savgol(3, 1): interior = mean of 3 (scipy's least-squares taps are 1/3 + 4e-17; tolerance-level),
// ends = the line fitted to the first / last three samples evaluated at the end ((5 y0 + 2 y1 - y2) / 6).
// tmp: 2 n doubles.  Returns 0 and lo = hi = 0 when no section qualifies.
// ======================================================================================
SH_HD int prox_canal_range(const double* area, int n, double* tmp, int* lo, int* hi) {
  double* sm = tmp;
  double* g = tmp + n;
  for (int i = 1; i + 1 < n; ++i) sm[i] = ((area[i - 1] + area[i]) + area[i + 1]) / 3.0;
  sm[0] = ((5.0 * area[0] + 2.0 * area[1]) - area[2]) / 6.0;
  sm[n - 1] = ((5.0 * area[n - 1] + 2.0 * area[n - 2]) - area[n - 3]) / 6.0;
  g[0] = sm[1] - sm[0];
  g[n - 1] = sm[n - 1] - sm[n - 2];
  for (int i = 1; i + 1 < n; ++i) g[i] = (sm[i + 1] - sm[i - 1]) / 2.0;
  int best_a = 0, best_len = 0, a = -1;
  for (int i = 0; i <= n; ++i) {
    const bool in = i < n && g[i] < 10.0;
    if (in && a < 0) a = i;
    if (!in && a >= 0) { if (i - a > best_len) { best_len = i - a; best_a = a; } a = -1; }
  }
  *lo = best_a; *hi = best_len > 0 ? best_a + best_len - 1 : best_a;
  return best_len;
}

// ======================================================================================
// K4  circle_fit.least_squares_circle residual   mesh.py:102
// minimise sum (R_i - mean R)^2 over the centre from the barycentre (Levenberg-Marquardt);
// returns residu = sum (R_i - mean R)^2 at the optimum.
// ======================================================================================
SH_HD double circle_residual_at(const double* xy, int n, double cx, double cy, double* g, double* H) {
  double sr = 0, sux = 0, suy = 0;
  for (int i = 0; i < n; ++i) {
    double dx = xy[2 * i] - cx, dy = xy[2 * i + 1] - cy;
    double r = sqrt(dx * dx + dy * dy);
    sr += r; sux += dx / r; suy += dy / r;
  }
  double rm = sr / n, mux = sux / n, muy = suy / n;
  double f2 = 0;
  if (g) { g[0] = g[1] = 0; H[0] = H[1] = H[2] = 0; }
  for (int i = 0; i < n; ++i) {
    double dx = xy[2 * i] - cx, dy = xy[2 * i + 1] - cy;
    double r = sqrt(dx * dx + dy * dy);
    double f = r - rm;
    f2 += f * f;
    if (g) {  // J_i = d f_i / d c = -(u_i - mean u)
      double jx = -(dx / r - mux), jy = -(dy / r - muy);
      g[0] += jx * f; g[1] += jy * f;
      H[0] += jx * jx; H[1] += jx * jy; H[2] += jy * jy;
    }
  }
  return f2;
}

SH_HD double circle_fit_residual(const double* xy, int n, double* cx_out = nullptr, double* cy_out = nullptr) {
  double cx = 0, cy = 0;
  for (int i = 0; i < n; ++i) { cx += xy[2 * i]; cy += xy[2 * i + 1]; }
  cx /= n; cy /= n;
  double lam = 1e-3, g[2], H[3];
  double f2 = circle_residual_at(xy, n, cx, cy, g, H);
  for (int it = 0; it < 200; ++it) {
    double a = H[0] * (1 + lam), b = H[1], d = H[2] * (1 + lam);
    double det = a * d - b * b;
    if (det == 0) break;
    double sx = -(d * g[0] - b * g[1]) / det, sy = -(-b * g[0] + a * g[1]) / det;
    double g2[2], H2[3];
    double f2n = circle_residual_at(xy, n, cx + sx, cy + sy, g2, H2);
    if (f2n <= f2) {
      cx += sx; cy += sy;
      bool done = (fabs(sx) + fabs(sy)) < 1e-13 * (1.0 + fabs(cx) + fabs(cy));
      f2 = f2n; g[0] = g2[0]; g[1] = g2[1]; H[0] = H2[0]; H[1] = H2[1]; H[2] = H2[2];
      lam *= 0.2;
      if (done) break;
    } else {
      lam *= 10.0;
      if (lam > 1e12) break;
    }
  }
  if (cx_out) { *cx_out = cx; *cy_out = cy; }
  return f2;
}

// ======================================================================================
// K12/K13  scipy.signal.savgol_filter(x, 10, 1) and find_peaks(height=-10, prominence=0.6,
// width=0.1)   bicipital_groove.py:107-118
// ======================================================================================
// savgol(10,1,mode="interp"): a 10-tap box mean over x[i-4 .. i+5] for i in [5, n-5) (scipy's
// lstsq-derived coefficients are 0.1 to within 3e-17 and differ in the last bit between
// LAPACK builds, so the canonical taps here are exactly 0.1: |y - scipy| <= 4e-15*|x|);
// the first / last 5 samples are a least-squares line through the first / last 10 samples
// (scipy _fit_edges_polyfit).
SH_HD double savgol10_1_at(const double* x, int i) {     // interior sample, 5 <= i < n-5
  double s = 0.0;
  for (int j = 0; j < 10; ++j) s += 0.1 * x[i - 4 + j];
  return s;
}
SH_HD void savgol10_1_edges(const double* x, int n, double* y);
SH_HD void savgol10_1(const double* x, int n, double* y) {
  for (int i = 5; i < n - 5; ++i) y[i] = savgol10_1_at(x, i);
  savgol10_1_edges(x, n, y);
}
SH_HD void savgol10_1_edges(const double* x, int n, double* y) {
  // edges: line fit a + b*t over t = 0..9
  for (int side = 0; side < 2; ++side) {
    const double* p = side == 0 ? x : x + (n - 10);
    double sy = 0, sty = 0;
    for (int t = 0; t < 10; ++t) { sy += p[t]; sty += t * p[t]; }
    // sum t = 45, sum t^2 = 285, n = 10: b = (10*sty - 45*sy)/(10*285 - 45*45), a = (sy - 45 b)/10
    double b = (10.0 * sty - 45.0 * sy) / 825.0;
    double a = (sy - 45.0 * b) / 10.0;
    if (side == 0) for (int t = 0; t < 5; ++t) y[t] = a + b * t;
    else for (int t = 5; t < 10; ++t) y[n - 10 + t] = a + b * t;
  }
}

struct Peak {
  int idx;
  double prominence, width, width_height;
};

// scipy _local_maxima_1d for one candidate start i (x[i-1] < x[i] is the caller's test): follows the
// plateau; returns the peak index (plateau midpoint) or -1, and the index the scan resumes from.
SH_HD int local_maximum_at(const double* x, int n, int i, int* resume) {
  const int i_max = n - 1;
  int ia = i + 1;
  while (ia < i_max && x[ia] == x[i]) ++ia;
  *resume = i;
  if (x[ia] < x[i]) { *resume = ia; return (i + (ia - 1)) / 2; }
  return -1;
}

// height / prominence (wlen=-1) / width (rel_height 0.5) filters of scipy.signal.find_peaks for one
// local maximum pk; true if it passes all three.
SH_HD bool peak_eval(const double* x, int n, int pk, double hmin, double pmin, double wmin, Peak* out) {
  if (!(x[pk] >= hmin)) return false;
  double left_min = x[pk], right_min = x[pk];
  int lb = pk, rb = pk, k = pk;
  while (0 <= k && x[k] <= x[pk]) { if (x[k] < left_min) { left_min = x[k]; lb = k; } --k; }
  k = pk;
  while (k <= n - 1 && x[k] <= x[pk]) { if (x[k] < right_min) { right_min = x[k]; rb = k; } ++k; }
  double prom = x[pk] - (left_min > right_min ? left_min : right_min);
  if (!(prom >= pmin)) return false;
  double h = x[pk] - prom * 0.5;
  k = pk;
  while (lb < k && h < x[k]) --k;
  double lip = (double)k;
  if (x[k] < h) lip += (h - x[k]) / (x[k + 1] - x[k]);
  k = pk;
  while (k < rb && h < x[k]) ++k;
  double rip = (double)k;
  if (x[k] < h) rip -= (h - x[k]) / (x[k - 1] - x[k]);
  double w = rip - lip;
  if (!(w >= wmin)) return false;
  out->idx = pk; out->prominence = prom; out->width = w; out->width_height = h;
  return true;
}

// Returns number of peaks written (all that pass the filters, ascending index), cap = capacity.
SH_HD int find_peaks_hpw(const double* x, int n, double hmin, double pmin, double wmin, Peak* out, int cap) {
  int np_ = 0;
  int i = 1, i_max = n - 1;
  while (i < i_max) {
    if (x[i - 1] < x[i]) {
      int resume;
      int pk = local_maximum_at(x, n, i, &resume);
      i = resume;
      if (pk >= 0) {
        Peak p;
        if (peak_eval(x, n, pk, hmin, pmin, wmin, &p)) {
          if (np_ < cap) out[np_] = p;
          ++np_;
        }
      }
    }
    ++i;
  }
  return np_;
}

// ======================================================================================
// K14  per-row peak features   bicipital_groove.py:102-156
// polar row: theta[M], r[M] = itr_centered_start (theta rolled to argmin, r about the slice
// AABB centre).  Writes up to 7 rows of raw features X_raw[.,9] and peak thetas.
// Feature order (:144-154): radius, nearest, next_nearest, z_scaled, prominence, width,
// width_height, canal_dist, n_peaks/7.
// scratch: 3*M doubles.
// ======================================================================================
#define SH_PEAK_CAP 64
SH_HD int groove_features_from_peaks(const double* theta, const double* r, int M, double z, double z_scaled, const double* canal_u,
                                     Peak* pk, int np_, int amin, double* Xraw, double* peak_theta, int* peak_idx);
SH_HD double wrapped_abs_diff(double v, double a) { return fabs(atan2(sin(v - a), cos(v - a))); }

SH_HD int groove_row_features(const double* theta, const double* r, int M, double z, double z_scaled,
                              const double* canal_u /*3: unit(axis0-axis1), CT*/, double* scratch,
                              double* Xraw /*7x9*/, double* peak_theta /*7*/, int* peak_idx /*7*/) {
  double* neg0 = scratch;        // -(r - mean r)
  double* filt = scratch + M;    // savgol
  double* roll = scratch + 2 * M;
  // np.mean over M=512: NumPy pairwise summation (blocks of 128, 8 accumulators)
  double mean;
  {
    double tot = 0.0;
    // pairwise_sum for n = 512: split 256/256 -> 128/128 each; leaf (<=128) uses 8 partial sums
    double leaf[8];
    int nleaf = 0;
    for (int base = 0; base < M; base += 128) {
      int len = (M - base) < 128 ? (M - base) : 128;
      double a8[8];
      const double* p = r + base;
      double res;
      if (len < 8) { res = 0.0; for (int k = 0; k < len; ++k) res += p[k]; }
      else {
        for (int k = 0; k < 8; ++k) a8[k] = p[k];
        int k8;
        for (k8 = 8; k8 < len - (len % 8); k8 += 8)
          for (int k = 0; k < 8; ++k) a8[k] += p[k8 + k];
        res = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
        for (; k8 < len; ++k8) res += p[k8];
      }
      leaf[nleaf++] = res;
    }
    // combine leaves pairwise (valid for M = 512: ((l0+l1)+(l2+l3)); generic fallback sequential)
    if (nleaf == 4) tot = (leaf[0] + leaf[1]) + (leaf[2] + leaf[3]);
    else { tot = 0.0; for (int k = 0; k < nleaf; ++k) tot += leaf[k]; }
    mean = tot / (double)M;
  }
  for (int k = 0; k < M; ++k) neg0[k] = -1.0 * (r[k] - mean);
  savgol10_1(neg0, M, filt);
  int amin = 0;
  for (int k = 1; k < M; ++k) if (filt[k] < filt[amin]) amin = k;
  // np.roll(radius, -amin): roll[k] = filt[(k + amin) % M]
  for (int k = 0; k < M; ++k) roll[k] = filt[(k + amin) % M];
  Peak pk[SH_PEAK_CAP];
  int np_ = find_peaks_hpw(roll, M, -10.0, 0.6, 0.1, pk, SH_PEAK_CAP);
  if (np_ > SH_PEAK_CAP) np_ = SH_PEAK_CAP;
  return groove_features_from_peaks(theta, r, M, z, z_scaled, canal_u, pk, np_, amin, Xraw, peak_theta, peak_idx);
}

// Second half of the per-row work (bicipital_groove.py:121-156): pk = peaks of the rolled, filtered
// row that passed find_peaks (ascending index, np_ <= SH_PEAK_CAP; modified in place), amin = roll.
SH_HD int groove_features_from_peaks(const double* theta, const double* r, int M, double z, double z_scaled, const double* canal_u,
                                     Peak* pk, int np_, int amin, double* Xraw /*7x9*/, double* peak_theta /*7*/, int* peak_idx /*7*/) {
  // keep the 7 most prominent (B-5: ascending index order among the kept)
  if (np_ > SH_MAXPEAK) {
    bool keep[SH_PEAK_CAP];
    for (int k = 0; k < np_; ++k) keep[k] = false;
    for (int s = 0; s < SH_MAXPEAK; ++s) {
      int b = -1;
      for (int k = 0; k < np_; ++k)
        if (!keep[k] && (b < 0 || pk[k].prominence > pk[b].prominence)) b = k;
      keep[b] = true;
    }
    int w = 0;
    for (int k = 0; k < np_; ++k) if (keep[k]) pk[w++] = pk[k];
    np_ = SH_MAXPEAK;
  }
  double th[SH_MAXPEAK];
  for (int k = 0; k < np_; ++k) {
    int idx = (pk[k].idx + amin) % M;     // (peaks - rmin) % interp_num with rmin = -amin
    peak_idx[k] = idx;
    th[k] = theta[idx];
    peak_theta[k] = th[k];
  }
  for (int k = 0; k < np_; ++k) {
    double near = 0.0, next = 0.0;
    if (np_ > 1) {
      // sorted wrapped distances to all peaks, dropping those that round to 0.00 (:46)
      double a[SH_MAXPEAK];
      int na = 0;
      for (int j = 0; j < np_; ++j) {
        double d = wrapped_abs_diff(th[k], th[j]);
        if (rint(d * 100.0) / 100.0 != 0.0) a[na++] = d;     // np.round(angs, 2) != 0
      }
      for (int p = 1; p < na; ++p) { double v = a[p]; int q = p - 1; while (q >= 0 && a[q] > v) { a[q + 1] = a[q]; --q; } a[q + 1] = v; }
      near = na > 0 ? a[0] : nan("");
      if (np_ > 2) next = na > 1 ? a[1] : nan("");
    }
    double rad = r[peak_idx[k]];
    double px = rad * cos(th[k]), py = rad * sin(th[k]);
    double dx = px - canal_u[0] * z, dy = py - canal_u[1] * z;
    double* X = Xraw + k * 9;
    X[0] = rad; X[1] = near; X[2] = next; X[3] = z_scaled; X[4] = pk[k].prominence; X[5] = pk[k].width;
    X[6] = pk[k].width_height; X[7] = sqrt(dx * dx + dy * dy); X[8] = (double)np_ / 7.0;
  }
  return np_;
}

// K15  onnxruntime TreeEnsembleClassifier walk (bicipital_groove.py:178-181): P(class 1) as f32.
SH_HD float rfc_proba1(const int32_t* feat, const float* thr, const int32_t* ti, const int32_t* fi,
                       const float* leafw, const int32_t* roots, int n_trees, const double* x) {
  double s = 0.0;
  for (int t = 0; t < n_trees; ++t) {
    int c = roots[t];
    while (ti[c] >= 0) c = (x[feat[c]] <= (double)thr[c]) ? ti[c] : fi[c];
    s += (double)leafw[c];
  }
  return (float)s;
}

// K17  local radius minimum near bg_theta for one row (bicipital_groove.py:199-229).
// theta/r0 = row of polar_0 (theta, r - mean r).  Returns bg_i_local (may be negative: the
// reference then indexes from the end of the row).
SH_HD int groove_local_min(const double* theta, const double* r0, int M, double bg_theta, int ivar) {
  int esti = searchsorted_left(theta, M, bg_theta);
  if (esti == M) esti = M - 1;
  int best = 0;
  double bv = 0;
  bool first = true;
  int pos = 0;
  if (ivar > esti) {
    // polar_0[:, (esti-ivar):] (negative start = tail) ++ polar_0[:, :(esti+ivar)]
    int start = M + (esti - ivar);
    if (start < 0) start = 0;      // a Python slice start below -M clamps to the beginning of the row
    for (int k = start; k < M; ++k, ++pos) if (first || r0[k] < bv) { bv = r0[k]; best = pos; first = false; }
    for (int k = 0; k < esti + ivar && k < M; ++k, ++pos) if (first || r0[k] < bv) { bv = r0[k]; best = pos; first = false; }
  } else {
    int hi = esti + ivar < M ? esti + ivar : M;
    for (int k = esti - ivar; k < hi; ++k, ++pos) if (first || r0[k] < bv) { bv = r0[k]; best = pos; first = false; }
  }
  return best + (esti - ivar);
}

// ======================================================================================
// K22  LsqEllipse().fit(xy).as_parameters() centre   anatomic_neck.py:139-144
// (Halir & Flusser).  S = the 6x6 scatter of [x^2, xy, y^2, x, y, 1] (upper triangle used).
// ======================================================================================
SH_HD bool inv3(const double* A, double* Ai) {
  double c00 = A[4] * A[8] - A[5] * A[7], c01 = A[5] * A[6] - A[3] * A[8], c02 = A[3] * A[7] - A[4] * A[6];
  double det = A[0] * c00 + A[1] * c01 + A[2] * c02;
  if (det == 0.0) return false;
  double id = 1.0 / det;
  Ai[0] = c00 * id; Ai[1] = (A[2] * A[7] - A[1] * A[8]) * id; Ai[2] = (A[1] * A[5] - A[2] * A[4]) * id;
  Ai[3] = c01 * id; Ai[4] = (A[0] * A[8] - A[2] * A[6]) * id; Ai[5] = (A[2] * A[3] - A[0] * A[5]) * id;
  Ai[6] = c02 * id; Ai[7] = (A[1] * A[6] - A[0] * A[7]) * id; Ai[8] = (A[0] * A[4] - A[1] * A[3]) * id;
  return true;
}

// real eigenvalues of a real 3x3 (assumed to have 3 real eigenvalues, as Halir-Flusser's M has)
SH_HD int eigvals3_real(const double* M, double* ev) {
  double tr = M[0] + M[4] + M[8];
  double c1 = (M[0] * M[4] - M[1] * M[3]) + (M[0] * M[8] - M[2] * M[6]) + (M[4] * M[8] - M[5] * M[7]);
  double det = M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
  // l^3 - tr l^2 + c1 l - det = 0 ; substitute l = t + tr/3
  double a = tr / 3.0;
  double p = c1 - tr * tr / 3.0;
  double q = -2.0 * a * a * a + a * c1 - det;   // t^3 + p t + q = 0
  int n = 0;
  if (p >= 0) {  // one real root
    double t = 0;  // Newton from 0 works since monotone
    double sq = sqrt(q * q / 4.0 + p * p * p / 27.0);
    t = cbrt(-q / 2.0 + sq) + cbrt(-q / 2.0 - sq);
    ev[n++] = t + a;
  } else {
    double m = 2.0 * sqrt(-p / 3.0);
    double arg = 3.0 * q / (p * m);
    if (arg > 1) arg = 1; if (arg < -1) arg = -1;
    double th = acos(arg) / 3.0;
    for (int k = 0; k < 3; ++k) ev[n++] = m * cos(th - 2.0 * M_PI * k / 3.0) + a;
  }
  // Newton polish on the characteristic polynomial
  for (int k = 0; k < n; ++k) {
    double l = ev[k];
    for (int it = 0; it < 8; ++it) {
      double f = ((l - tr) * l + c1) * l - det;
      double df = (3.0 * l - 2.0 * tr) * l + c1;
      if (df == 0) break;
      double dl = f / df;
      l -= dl;
      if (fabs(dl) <= 1e-16 * fabs(l)) break;
    }
    ev[k] = l;
  }
  return n;
}

SH_HD bool ellipse_center_from_scatter(const double* S /*6x6 row-major, symmetric*/, double* cx, double* cy) {
  double S1[9], S2[9], S3[9], S3i[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { S1[i * 3 + j] = S[i * 6 + j]; S2[i * 3 + j] = S[i * 6 + 3 + j]; S3[i * 3 + j] = S[(3 + i) * 6 + 3 + j]; }
  if (!inv3(S3, S3i)) return false;
  // T = S3^-1 S2^T ; A = S1 - S2 T ; M = C1^-1 A with C1^-1 = [[0,0,.5],[0,-1,0],[.5,0,0]]
  double T[9], A[9], Mx[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += S3i[i * 3 + k] * S2[j * 3 + k]; T[i * 3 + j] = s; }
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) { double s = 0; for (int k = 0; k < 3; ++k) s += S2[i * 3 + k] * T[k * 3 + j]; A[i * 3 + j] = S1[i * 3 + j] - s; }
  for (int j = 0; j < 3; ++j) { Mx[j] = 0.5 * A[6 + j]; Mx[3 + j] = -A[3 + j]; Mx[6 + j] = 0.5 * A[j]; }
  double ev[3];
  int ne = eigvals3_real(Mx, ev);
  for (int e = 0; e < ne; ++e) {
    // eigenvector = cross product of two rows of (M - l I), pick the largest
    double R[9];
    for (int i = 0; i < 9; ++i) R[i] = Mx[i];
    R[0] -= ev[e]; R[4] -= ev[e]; R[8] -= ev[e];
    double v[3], best[3] = {0, 0, 0}, bn = -1;
    for (int a = 0; a < 3; ++a)
      for (int b = a + 1; b < 3; ++b) {
        cross3(R + 3 * a, R + 3 * b, v);
        double nn = dot3(v, v);
        if (nn > bn) { bn = nn; best[0] = v[0]; best[1] = v[1]; best[2] = v[2]; }
      }
    if (bn <= 0) continue;
    double cond = 4.0 * best[0] * best[2] - best[1] * best[1];
    if (cond > 0) {
      // a2 = -S3^-1 S2^T a1 = -T a1
      double a2[3];
      for (int i = 0; i < 3; ++i) a2[i] = -(T[i * 3] * best[0] + T[i * 3 + 1] * best[1] + T[i * 3 + 2] * best[2]);
      double a = best[0], b = best[1] / 2.0, c = best[2], d = a2[0] / 2.0, f = a2[1] / 2.0;
      double den = b * b - a * c;
      *cx = (c * d - b * f) / den;
      *cy = (a * f - b * d) / den;
      return true;
    }
  }
  return false;
}

// orthonormal basis of the plane with unit normal n (oracle fits.plane_basis)
SH_HD void plane_basis(const double* n, double* u, double* v) {
  int k = 0;
  if (fabs(n[1]) < fabs(n[k])) k = 1;
  if (fabs(n[2]) < fabs(n[k])) k = 2;
  double e[3] = {0, 0, 0};
  e[k] = 1.0;
  cross3(n, e, u);
  double l = norm3(u);
  for (int i = 0; i < 3; ++i) u[i] /= l;
  cross3(n, u, v);
}

// ======================================================================================
// K24/K25  trans-epicondylar helpers   epicondyle.py:33-81, utils.py:36-97
// ======================================================================================
// Convex hull of a simple closed polygon given as an open CCW vertex list (Melkman, O(n)).
// hull: capacity >= n+1 ints (indices into xy), returns count (CCW, no repeated end).
SH_HD double orient2(const double* a, const double* b, const double* c) {
  return (b[0] - a[0]) * (c[1] - a[1]) - (b[1] - a[1]) * (c[0] - a[0]);
}

// Andrew monotone chain on a copy of the indices (insertion sort by (x,y)); robust for any
// point set, n <= SH_MAXSEG.  idx: scratch n ints; hull: out capacity 2n.
SH_HD int convex_hull_2d(const double* xy, int n, int* idx, int* hull) {
  for (int i = 0; i < n; ++i) idx[i] = i;
  // shell sort by (x, y)
  for (int gap = n / 2; gap > 0; gap /= 2)
    for (int i = gap; i < n; ++i) {
      int t = idx[i];
      double tx = xy[2 * t], ty = xy[2 * t + 1];
      int j = i;
      while (j >= gap) {
        int u = idx[j - gap];
        double ux = xy[2 * u], uy = xy[2 * u + 1];
        if (ux > tx || (ux == tx && uy > ty)) { idx[j] = u; j -= gap; } else break;
      }
      idx[j] = t;
    }
  int k = 0;
  for (int i = 0; i < n; ++i) {
    while (k >= 2 && orient2(xy + 2 * hull[k - 2], xy + 2 * hull[k - 1], xy + 2 * idx[i]) <= 0) --k;
    hull[k++] = idx[i];
  }
  int lo = k + 1;
  for (int i = n - 2; i >= 0; --i) {
    while (k >= lo && orient2(xy + 2 * hull[k - 2], xy + 2 * hull[k - 1], xy + 2 * idx[i]) <= 0) --k;
    hull[k++] = idx[i];
  }
  return k - 1;
}

// Convex hull of a SIMPLE polygon given as an open vertex list in boundary order (Melkman's
// O(n) deque algorithm; collinear points are dropped like the monotone chain above does).
// dq: scratch 2n+2 ints; returns the hull size, hull vertices (CCW) in hull[0..).  Falls back to
// the sort-based hull when the first three vertices are collinear.
SH_HD int convex_hull_simple_polygon(const double* xy, int n, int* dq, int* hull) {
  if (n < 3) { for (int i = 0; i < n; ++i) hull[i] = i; return n; }
  double o = orient2(xy, xy + 2, xy + 4);
  if (o == 0.0) return convex_hull_2d(xy, n, dq, hull);
  int bot = n - 2, top = bot + 3;
  dq[bot] = dq[top] = 2;
  if (o > 0) { dq[bot + 1] = 0; dq[bot + 2] = 1; } else { dq[bot + 1] = 1; dq[bot + 2] = 0; }
  for (int i = 3; i < n; ++i) {
    const double* v = xy + 2 * i;
    if (orient2(xy + 2 * dq[bot], xy + 2 * dq[bot + 1], v) > 0 && orient2(xy + 2 * dq[top - 1], xy + 2 * dq[top], v) > 0) continue;
    while (top - bot >= 2 && orient2(xy + 2 * dq[bot], xy + 2 * dq[bot + 1], v) <= 0) ++bot;
    dq[--bot] = i;
    while (top - bot >= 2 && orient2(xy + 2 * dq[top - 1], xy + 2 * dq[top], v) <= 0) --top;
    dq[++top] = i;
  }
  int nh = top - bot;
  for (int k = 0; k < nh; ++k) hull[k] = dq[bot + k];
  return nh;
}

struct Rect2 {
  double cx, cy, mx, my, L, W, area;
};

// minimum-area rectangle over hull edges, first minimum in hull order (oracle te.min_area_rect)
SH_HD bool min_area_rect(const double* xy, const int* hull, int nh, Rect2* out) {
  bool have = false;
  for (int i = 0; i < nh; ++i) {
    const double* p = xy + 2 * hull[i];
    const double* q = xy + 2 * hull[(i + 1) % nh];
    double ex = q[0] - p[0], ey = q[1] - p[1];
    double ln = hypot(ex, ey);
    if (ln == 0) continue;
    ex /= ln; ey /= ln;
    double nx = -ey, ny = ex;
    double amin = 1e300, amax = -1e300, bmin = 1e300, bmax = -1e300;
    for (int k = 0; k < nh; ++k) {
      const double* h = xy + 2 * hull[k];
      double a = h[0] * ex + h[1] * ey, b = h[0] * nx + h[1] * ny;
      amin = a < amin ? a : amin; amax = a > amax ? a : amax;
      bmin = b < bmin ? b : bmin; bmax = b > bmax ? b : bmax;
    }
    double ea = amax - amin, eb = bmax - bmin, area = ea * eb;
    if (!have || area < out->area) {
      have = true;
      double ca = 0.5 * (amax + amin), cb = 0.5 * (bmax + bmin);
      out->cx = ex * ca + nx * cb; out->cy = ey * ca + ny * cb; out->area = area;
      if (ea >= eb) { out->mx = ex; out->my = ey; out->L = ea; out->W = eb; }
      else { out->mx = nx; out->my = ny; out->L = eb; out->W = ea; }
    }
  }
  return have;
}

// Area centroid of an open vertex list about its first vertex (oracle te._poly_centroid).
SH_HD void poly_centroid(const double* p, int n, double* cx, double* cy, double* area) {
  double a2 = 0, sx = 0, sy = 0;
  for (int i = 0; i < n; ++i) {
    int j = (i + 1) % n;
    double x = p[2 * i] - p[0], y = p[2 * i + 1] - p[1], xn = p[2 * j] - p[0], yn = p[2 * j + 1] - p[1];
    double cr = x * yn - xn * y;
    a2 += cr; sx += (x + xn) * cr; sy += (y + yn) * cr;
  }
  double a = a2 / 2.0;
  *area = a;
  *cx = p[0] + sx / (6.0 * a);
  *cy = p[1] + sy / (6.0 * a);
}

// Pieces of a simple CCW ring (open list, n points) inside {(p-c).m > w0}: centroids appended to
// cents (2 doubles each); returns the number of pieces.  scratch: >= (2n + 8*maxchains) doubles.
#define SH_TE_MAXCH 32
SH_HD int clip_halfplane_pieces(const double* pts, int n, double cx, double cy, double mx, double my, double w0,
                                double* cents, int cap, double* scratch) {
  double* poly = scratch;  // up to 2*(n+2*chains) doubles
  int start = -1;
  bool any_in = false, all_in = true;
  for (int i = 0; i < n; ++i) {
    bool in = ((pts[2 * i] - cx) * mx + (pts[2 * i + 1] - cy) * my - w0) > 0;
    any_in |= in; all_in &= in;
  }
  if (!any_in) return 0;
  if (all_in) { double a; if (cap > 0) poly_centroid(pts, n, cents, cents + 1, &a); return 1; }
  for (int i = 0; i < n && start < 0; ++i) {
    int j = (i + 1) % n;
    bool ii = ((pts[2 * i] - cx) * mx + (pts[2 * i + 1] - cy) * my - w0) > 0;
    bool jj = ((pts[2 * j] - cx) * mx + (pts[2 * j + 1] - cy) * my - w0) > 0;
    if (!ii && jj) start = i;
  }
  // chains: [begin, end) into poly plus s_in/s_out along the clip line
  int cb[SH_TE_MAXCH], ce[SH_TE_MAXCH];
  double sin_[SH_TE_MAXCH], sout[SH_TE_MAXCH];
  int nc = 0, np_ = 0;
  bool open = false;
  double px = -my, py = mx;
  for (int k = 0; k < n; ++k) {
    int i = (start + k) % n, j = (start + k + 1) % n;
    double fi = (pts[2 * i] - cx) * mx + (pts[2 * i + 1] - cy) * my - w0;
    double fj = (pts[2 * j] - cx) * mx + (pts[2 * j + 1] - cy) * my - w0;
    bool ii = fi > 0, jj = fj > 0;
    if (ii != jj) {
      double t = fi / (fi - fj);
      double x = pts[2 * i] + t * (pts[2 * j] - pts[2 * i]), y = pts[2 * i + 1] + t * (pts[2 * j + 1] - pts[2 * i + 1]);
      double s = (x - cx) * px + (y - cy) * py;
      if (jj) {
        if (nc >= SH_TE_MAXCH) return -1;
        cb[nc] = np_; sin_[nc] = s; open = true;
        poly[2 * np_] = x; poly[2 * np_ + 1] = y; ++np_;
        poly[2 * np_] = pts[2 * j]; poly[2 * np_ + 1] = pts[2 * j + 1]; ++np_;
      } else if (open) {
        poly[2 * np_] = x; poly[2 * np_ + 1] = y; ++np_;
        sout[nc] = s; ce[nc] = np_; ++nc; open = false;
      }
    } else if (jj && open) {
      poly[2 * np_] = pts[2 * j]; poly[2 * np_ + 1] = pts[2 * j + 1]; ++np_;
    }
  }
  // pair the crossings along the line: sort 2*nc events by s; (0,1),(2,3),...
  int ev_chain[2 * SH_TE_MAXCH], ev_out[2 * SH_TE_MAXCH];
  double ev_s[2 * SH_TE_MAXCH];
  int ne = 0;
  for (int c = 0; c < nc; ++c) { ev_s[ne] = sin_[c]; ev_chain[ne] = c; ev_out[ne] = 0; ++ne; ev_s[ne] = sout[c]; ev_chain[ne] = c; ev_out[ne] = 1; ++ne; }
  for (int a = 1; a < ne; ++a) {
    double s = ev_s[a]; int c = ev_chain[a], o = ev_out[a]; int b = a - 1;
    while (b >= 0 && ev_s[b] > s) { ev_s[b + 1] = ev_s[b]; ev_chain[b + 1] = ev_chain[b]; ev_out[b + 1] = ev_out[b]; --b; }
    ev_s[b + 1] = s; ev_chain[b + 1] = c; ev_out[b + 1] = o;
  }
  int next_after_out[SH_TE_MAXCH];
  for (int a = 0; a + 1 < ne; a += 2) {
    if (ev_out[a]) next_after_out[ev_chain[a]] = ev_chain[a + 1];
    if (ev_out[a + 1]) next_after_out[ev_chain[a + 1]] = ev_chain[a];
  }
  bool used[SH_TE_MAXCH];
  for (int c = 0; c < nc; ++c) used[c] = false;
  double* comp = poly + 2 * np_;
  int npieces = 0;
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
  return npieces;
}

}  // namespace sh
