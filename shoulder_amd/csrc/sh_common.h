// sh_common.h -- small fp64 helpers shared by device kernels and host-side stages.
// Everything here is plain sequential arithmetic written so that, compiled with
// -ffp-contract=off, it performs the same IEEE operations in the same order as the NumPy
// expressions of the reference it restates (file:line given per function).
#pragma once
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define SH_HD __host__ __device__ inline
#else
#define SH_HD inline
#endif

#define SH_UNET_MAXBASE 256  // first-level channels the UNet kernels are sized for (first-conv weight tile in LDS)
#define SH_MAXSEG 1024      // capacity: crossing segments per (mesh, plane)
#define SH_NFULL 200        // slice.py:213
#define SH_NDIST 200        // slice.py:260
#define SH_NPROX 600        // slice.py:236 "must not change needed for anp cnn"
#define SH_NPSCAN 100       // mesh.py:153 ProxObb area scan
#define SH_MPROX 512        // slice.py:237
#define SH_ANP_ROWS 512     // rows 88..599 (anatomic_neck.py:34)
#define SH_ANP_ROW0 88
#define SH_GROOVE_ROW0 150
#define SH_GROOVE_NROWS 330
#define SH_MAXPEAK 7        // bicipital_groove.py:122
#define SH_SECTION_TOL 1e-8 // trimesh tol.merge used by intersections.mesh_plane
// Per-humerus status word written by the kernels.  A capacity overflow is recorded unconditionally (atomicExch), a geometry
// failure only where nothing is recorded yet (atomicCAS from 0): a truncated slice also fails to close its contours, and the
// status should name the cause, not the consequence.
#define SH_ERR_CAPACITY_DEV (-4)
#define SH_ERR_GEOMETRY_DEV (-5)
#define SH_ERR_HULL_DEV (-7)      // internal: the device hull gave this humerus up (k_hull.h); sh_collect re-does that humerus with the host quickhull

namespace sh {

// utils.transform_pts (utils.py:172-188): (T * [p;1])[:3]; T row-major 4x4.
// np.matmul of a (4,4) by (4,n) accumulates k = 0..3 in order.
SH_HD void xform_pt(const double* T, double x, double y, double z, double* o) {
  o[0] = ((T[0] * x + T[1] * y) + T[2] * z) + T[3];
  o[1] = ((T[4] * x + T[5] * y) + T[6] * z) + T[7];
  o[2] = ((T[8] * x + T[9] * y) + T[10] * z) + T[11];
}

SH_HD void mat4_mul(const double* A, const double* B, double* C) {
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      double s = 0.0;
      for (int k = 0; k < 4; ++k) s += A[i * 4 + k] * B[k * 4 + j];
      C[i * 4 + j] = s;
    }
}

SH_HD void mat4_identity(double* A) {
  for (int i = 0; i < 16; ++i) A[i] = (i % 5 == 0) ? 1.0 : 0.0;
}

// General 4x4 inverse by Gauss-Jordan with partial pivoting (stands in for np.linalg.inv).
SH_HD bool mat4_inv(const double* A, double* Ai) {
  double m[4][8];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      m[i][j] = A[i * 4 + j];
      m[i][4 + j] = (i == j) ? 1.0 : 0.0;
    }
  for (int c = 0; c < 4; ++c) {
    int p = c;
    for (int r = c + 1; r < 4; ++r)
      if (fabs(m[r][c]) > fabs(m[p][c])) p = r;
    if (m[p][c] == 0.0) return false;
    if (p != c)
      for (int j = 0; j < 8; ++j) { double t = m[c][j]; m[c][j] = m[p][j]; m[p][j] = t; }
    double d = 1.0 / m[c][c];
    for (int j = 0; j < 8; ++j) m[c][j] *= d;
    for (int r = 0; r < 4; ++r)
      if (r != c) {
        double f = m[r][c];
        if (f != 0.0)
          for (int j = 0; j < 8; ++j) m[r][j] -= f * m[c][j];
      }
  }
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) Ai[i * 4 + j] = m[i][4 + j];
  return true;
}

// utils.inv_transform (utils.py:227-256): inv([R|0]) * inv([I|t]).
SH_HD bool inv_transform(const double* T, double* out) {
  double rot[16], tr[16], ri[16], ti[16];
  for (int i = 0; i < 16; ++i) rot[i] = T[i];
  rot[3] = rot[7] = rot[11] = 0.0;
  rot[15] = 1.0;
  mat4_identity(tr);
  tr[3] = T[3]; tr[7] = T[7]; tr[11] = T[11];
  if (!mat4_inv(rot, ri) || !mat4_inv(tr, ti)) return false;
  mat4_mul(ri, ti, out);
  return true;
}

SH_HD double norm3(const double* v) { return sqrt((v[0] * v[0] + v[1] * v[1]) + v[2] * v[2]); }
SH_HD void cross3(const double* a, const double* b, double* c) {
  c[0] = a[1] * b[2] - a[2] * b[1];
  c[1] = a[2] * b[0] - a[0] * b[2];
  c[2] = a[0] * b[1] - a[1] * b[0];
}
SH_HD double dot3(const double* a, const double* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }

SH_HD double det3_of4(const double* T) {
  // determinant of a 4x4 whose last row is (0,0,0,1) = det of the upper-left 3x3
  return T[0] * (T[5] * T[10] - T[6] * T[9]) - T[1] * (T[4] * T[10] - T[6] * T[8]) +
         T[2] * (T[4] * T[9] - T[5] * T[8]);
}

// utils.construct_csys (utils.py:289-318). vz, vy: two xyz points each. out: CT -> csys.
SH_HD bool construct_csys(const double* vz, const double* vy, double* out) {
  double pos[3], zh[3], xh[3], yh[3];
  for (int k = 0; k < 3; ++k) pos[k] = (vz[k] + vz[3 + k]) / 2.0;
  for (int k = 0; k < 3; ++k) { zh[k] = vz[k] - vz[3 + k]; xh[k] = vy[k] - vy[3 + k]; }
  double n = norm3(zh);
  for (int k = 0; k < 3; ++k) zh[k] /= n;
  n = norm3(xh);
  for (int k = 0; k < 3; ++k) xh[k] /= n;
  cross3(xh, zh, yh);
  n = norm3(yh);
  for (int k = 0; k < 3; ++k) yh[k] /= n;
  cross3(yh, zh, xh);
  n = norm3(xh);
  for (int k = 0; k < 3; ++k) xh[k] /= n;
  double T[16];
  for (int k = 0; k < 3; ++k) { T[k * 4 + 0] = xh[k]; T[k * 4 + 1] = yh[k]; T[k * 4 + 2] = zh[k]; T[k * 4 + 3] = pos[k]; }
  T[12] = T[13] = T[14] = 0.0; T[15] = 1.0;
  if (rint(det3_of4(T)) == -1.0)
    for (int k = 0; k < 4; ++k) T[k * 4 + 0] *= -1.0;
  return inv_transform(T, out);
}

// numpy.linspace(start, stop, num)[k]  (num > 1, endpoint=True): k*step + start, last = stop.
SH_HD double linspace_at(double start, double stop, int num, int k) {
  if (k == num - 1) return stop;
  double step = (stop - start) / (double)(num - 1);
  return (double)k * step + start;
}

// slice.py:157-164 with return_odd=False: [int((1-c1)*n), int((1-c0)*n))
SH_HD void cutoff_range(int n, double c0, double c1, int* a, int* b) {
  *a = (int)((1.0 - c1) * (double)n);
  *b = (int)((1.0 - c0) * (double)n);
}

// np.searchsorted(a, v, side='left') on an array that need not be sorted: NumPy's plain
// binary search (npy_binsearch left): while (lo < hi) { mid = lo + ((hi-lo)>>1); a[mid] < v ? lo=mid+1 : hi=mid }
SH_HD int searchsorted_left(const double* a, int n, double v, int stride = 1) {
  int lo = 0, hi = n;
  while (lo < hi) {
    int mid = lo + ((hi - lo) >> 1);
    if (a[(size_t)mid * stride] < v) lo = mid + 1; else hi = mid;
  }
  return lo;
}

// np.interp(x, xp, fp) for one x (compiled_base.c arr_interp): j = largest index with
// xp[j] <= x; left/right clamp; slope*(x-xp[j]) + fp[j].
SH_HD double interp1(double x, const double* xp, const double* fp, int n, int sxp = 1, int sfp = 1) {
  if (x < xp[0]) return fp[0];                       // left
  if (!(x < xp[(size_t)(n - 1) * sxp])) return fp[(size_t)(n - 1) * sfp];  // x >= last (right / exact end)
  int lo = 0, hi = n - 1;  // invariant xp[lo] <= x < xp[hi]
  while (hi - lo > 1) {
    int mid = (lo + hi) >> 1;
    if (x >= xp[(size_t)mid * sxp]) lo = mid; else hi = mid;
  }
  double x0 = xp[(size_t)lo * sxp], f0 = fp[(size_t)lo * sfp];
  if (x0 == x) return f0;
  double slope = (fp[(size_t)(lo + 1) * sfp] - f0) / (xp[(size_t)(lo + 1) * sxp] - x0);
  return slope * (x - x0) + f0;
}

// NumPy's np.interp search (compiled_base.c binary_search_with_guess): identical to a plain
// search on sorted xp, and reproduces NumPy's answer on the not-quite-sorted theta rows of
// anatomic_neck.py:43-44 because the previous index is carried as the guess.
SH_HD int np_search_with_guess(double key, const double* arr, int len, int guess) {
  int imin = 0, imax = len;
  if (key > arr[len - 1]) return len;
  else if (key < arr[0]) return -1;
  if (len <= 4) { int i; for (i = 1; i < len && key >= arr[i]; ++i) {} return i - 1; }
  if (guess > len - 3) guess = len - 3;
  if (guess < 1) guess = 1;
  if (key < arr[guess]) {
    if (key < arr[guess - 1]) {
      imax = guess - 1;
      if (guess > 8 && key >= arr[guess - 8]) imin = guess - 8;
    } else return guess - 1;
  } else {
    if (key < arr[guess + 1]) return guess;
    else if (key < arr[guess + 2]) return guess + 1;
    else {
      imin = guess + 2;
      if (guess < len - 8 - 1 && key < arr[guess + 8]) imax = guess + 8;
    }
  }
  while (imin < imax) {
    int imid = imin + ((imax - imin) >> 1);
    if (key >= arr[imid]) imin = imid + 1; else imax = imid;
  }
  return imin - 1;
}

// One np.interp evaluation with the carried guess *j (arr_interp loop body).
SH_HD double np_interp_step(double x, const double* xp, const double* fp, int n, int* j) {
  *j = np_search_with_guess(x, xp, n, *j);
  if (*j == -1) return fp[0];
  if (*j == n) return fp[n - 1];
  if (*j == n - 1) return fp[*j];
  if (xp[*j] == x) return fp[*j];
  double slope = (fp[*j + 1] - fp[*j]) / (xp[*j + 1] - xp[*j]);
  double r = slope * (x - xp[*j]) + fp[*j];
  if (r != r) {  // NaN: NumPy retries from the right neighbour
    r = slope * (x - xp[*j + 1]) + fp[*j + 1];
    if (r != r && fp[*j] == fp[*j + 1]) r = fp[*j];
  }
  return r;
}

// ---- symmetric 3x3 eigen-decomposition (cyclic Jacobi, fp64) ---------------------------
// w ascending, V columns = eigenvectors (row-major V[r*3+c]).
SH_HD void eig_sym3(const double* Ain, double* w, double* V) {
  double A[9];
  for (int i = 0; i < 9; ++i) { A[i] = Ain[i]; V[i] = (i % 4 == 0) ? 1.0 : 0.0; }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = fabs(A[1]) + fabs(A[2]) + fabs(A[5]);
    double dg = fabs(A[0]) + fabs(A[4]) + fabs(A[8]);
    if (off <= 1e-300 || off <= 1e-22 * dg) break;
    for (int p = 0; p < 2; ++p)
      for (int q = p + 1; q < 3; ++q) {
        double apq = A[p * 3 + q];
        if (apq == 0.0) continue;
        double theta = (A[q * 3 + q] - A[p * 3 + p]) / (2.0 * apq);
        double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 3; ++k) {
          double akp = A[k * 3 + p], akq = A[k * 3 + q];
          A[k * 3 + p] = c * akp - s * akq;
          A[k * 3 + q] = s * akp + c * akq;
        }
        for (int k = 0; k < 3; ++k) {
          double apk = A[p * 3 + k], aqk = A[q * 3 + k];
          A[p * 3 + k] = c * apk - s * aqk;
          A[q * 3 + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < 3; ++k) {
          double vkp = V[k * 3 + p], vkq = V[k * 3 + q];
          V[k * 3 + p] = c * vkp - s * vkq;
          V[k * 3 + q] = s * vkp + c * vkq;
        }
      }
  }
  w[0] = A[0]; w[1] = A[4]; w[2] = A[8];
  for (int i = 0; i < 2; ++i)
    for (int j = i + 1; j < 3; ++j)
      if (w[j] < w[i]) {
        double t = w[i]; w[i] = w[j]; w[j] = t;
        for (int k = 0; k < 3; ++k) { double u = V[k * 3 + i]; V[k * 3 + i] = V[k * 3 + j]; V[k * 3 + j] = u; }
      }
}

// Dominant eigenvector of a symmetric PSD 3x3 by power iteration on C (north-star "PCA with
// power iteration"); converged to |delta| < 1e-15, seeded by the largest column, then
// polished by one Jacobi solve when the spectrum is too flat for fast convergence.
SH_HD void dominant_eigvec3(const double* C, double* v) {
  int best = 0;
  double bn = -1.0;
  for (int c = 0; c < 3; ++c) {
    double n = C[c] * C[c] + C[3 + c] * C[3 + c] + C[6 + c] * C[6 + c];
    if (n > bn) { bn = n; best = c; }
  }
  v[0] = C[best]; v[1] = C[3 + best]; v[2] = C[6 + best];
  double n = norm3(v);
  if (n == 0.0) { v[0] = 1; v[1] = 0; v[2] = 0; return; }
  for (int k = 0; k < 3; ++k) v[k] /= n;
  bool ok = false;
  for (int it = 0; it < 200; ++it) {
    double u[3] = {dot3(C, v), dot3(C + 3, v), dot3(C + 6, v)};
    n = norm3(u);
    if (n == 0.0) break;
    for (int k = 0; k < 3; ++k) u[k] /= n;
    double d = fabs(u[0] - v[0]) + fabs(u[1] - v[1]) + fabs(u[2] - v[2]);
    for (int k = 0; k < 3; ++k) v[k] = u[k];
    if (d < 1e-16) { ok = true; break; }
  }
  if (!ok) {
    double w[3], V[9];
    eig_sym3(C, w, V);
    v[0] = V[2]; v[1] = V[5]; v[2] = V[8];
  }
}

}  // namespace sh
