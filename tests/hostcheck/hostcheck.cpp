// Test shim (NOT product): instantiates the SH_HD routines of shoulder_amd/csrc/sh_scalar.h
// on the host so CPU tests can compare the exact source the GPU runs against the oracle.
#include "../../shoulder_amd/csrc/sh_scalar.h"
#include <vector>
using namespace sh;
extern "C" {
int hc_cpd(const double* x, int n) { std::vector<double> s(n * n + n * n); return cpd_one_bkp(x, n, s.data()); }
double hc_circle(const double* xy, int n, double* c) { return circle_fit_residual(xy, n, c, c + 1); }
void hc_savgol(const double* x, int n, double* y) { savgol10_1(x, n, y); }
int hc_prox_canal_range(const double* a, int n, int* lo, int* hi) { std::vector<double> t(2 * n); return prox_canal_range(a, n, t.data(), lo, hi); }
int hc_find_peaks(const double* x, int n, double h, double p, double w, int* idx, double* prom, double* wid, double* wh, int cap) {
  std::vector<Peak> pk(cap);
  int k = find_peaks_hpw(x, n, h, p, w, pk.data(), cap);
  for (int i = 0; i < k && i < cap; ++i) { idx[i] = pk[i].idx; prom[i] = pk[i].prominence; wid[i] = pk[i].width; wh[i] = pk[i].width_height; }
  return k;
}
int hc_groove_row(const double* th, const double* r, int M, double z, double zs, const double* cu, double* X, double* pth, int* pidx) {
  std::vector<double> s(3 * M);
  return groove_row_features(th, r, M, z, zs, cu, s.data(), X, pth, pidx);
}
float hc_rfc(const int32_t* feat, const float* thr, const int32_t* ti, const int32_t* fi, const float* lw, const int32_t* roots, int nt, const double* x) {
  return rfc_proba1(feat, thr, ti, fi, lw, roots, nt, x);
}
int hc_local_min(const double* th, const double* r0, int M, double bg, int ivar) { return groove_local_min(th, r0, M, bg, ivar); }
int hc_ellipse(const double* S, double* c) { return ellipse_center_from_scatter(S, c, c + 1) ? 0 : -1; }
int hc_mrr(const double* xy, int n, double* out7) {
  std::vector<int> idx(n), hull(2 * n + 2);
  int nh = convex_hull_2d(xy, n, idx.data(), hull.data());
  Rect2 r;
  if (!min_area_rect(xy, hull.data(), nh, &r)) return -1;
  out7[0] = r.cx; out7[1] = r.cy; out7[2] = r.mx; out7[3] = r.my; out7[4] = r.L; out7[5] = r.W; out7[6] = r.area;
  return nh;
}
int hc_mrr_ring(const double* xy, int n, double* out7) {
  std::vector<int> dq(2 * n + 8), hull(2 * n + 8);
  int nh = convex_hull_simple_polygon(xy, n, dq.data(), hull.data());
  Rect2 r;
  if (!min_area_rect(xy, hull.data(), nh, &r)) return -1;
  out7[0] = r.cx; out7[1] = r.cy; out7[2] = r.mx; out7[3] = r.my; out7[4] = r.L; out7[5] = r.W; out7[6] = r.area;
  return nh;
}
int hc_clip(const double* pts, int n, double cx, double cy, double mx, double my, double w0, double* cents, int cap) {
  std::vector<double> s(4 * n + 64 * SH_TE_MAXCH);
  return clip_halfplane_pieces(pts, n, cx, cy, mx, my, w0, cents, cap, s.data());
}
int hc_construct_csys(const double* vz, const double* vy, double* out) { return construct_csys(vz, vy, out) ? 0 : -1; }
int hc_inv_transform(const double* T, double* out) { return inv_transform(T, out) ? 0 : -1; }
void hc_eig_sym3(const double* A, double* w, double* V) { eig_sym3(A, w, V); }
void hc_dominant(const double* C, double* v) { dominant_eigvec3(C, v); }
double hc_interp(double x, const double* xp, const double* fp, int n) { return interp1(x, xp, fp, n); }
double hc_linspace(double a, double b, int n, int k) { return linspace_at(a, b, n, k); }
}

#include "../../shoulder_amd/csrc/sh_hull.h"
extern "C" int hc_hull_eps(const double* pts, int n, double eps_rel, int* nv, int* nf) {
  shhull::Hull H;
  if (!shhull::convex_hull_eps(pts, n, H, eps_rel)) return -shhull::hull_fail_reason();
  *nv = (int)H.vert_ids.size(); *nf = (int)H.tris.size() / 3;
  return 0;
}
extern "C" int hc_hull(const double* pts, int n, int* vert_ids, int cap_v, int* tris, int cap_f, int* n_edges) {
  shhull::Hull H;
  if (!shhull::convex_hull(pts, n, H)) return -1;
  int nv = (int)H.vert_ids.size(), nf = (int)H.tris.size() / 3;
  if (nv > cap_v || nf > cap_f) return -2;
  for (int i = 0; i < nv; ++i) vert_ids[i] = H.vert_ids[i];
  for (int i = 0; i < 3 * nf; ++i) tris[i] = H.tris[i];
  *n_edges = (int)H.edges.size() / 4;
  return nv * 100000 + nf;
}

#include "hull_rounds_ref.h"
// Round-based quickhull (reference restatement of the device kernel): returns the number of faces (>= 0) or -fail code.
// tris: point indices, 3 per face; info: {n hull vertices, rounds, insertions}.
extern "C" int hc_hull_rounds(const float* pts, int n, int K, int* tris, int cap_f, int* info) {
  hullref::Out o;
  int rc = hullref::hull_rounds(pts, n, o, K);
  info[0] = (int)o.vert_ids.size(); info[1] = o.rounds; info[2] = o.insertions;
  if (rc) return -rc;
  int nf = (int)o.tris.size() / 3;
  if (nf > cap_f) return -99;
  for (int i = 0; i < 3 * nf; ++i) tris[i] = o.tris[i];
  return nf;
}
// the host quickhull's faces as point indices (for set comparison with the round-based one)
extern "C" int hc_hull_tris_pts(const double* pts, int n, int* tris, int cap_f) {
  shhull::Hull H;
  if (!shhull::convex_hull(pts, n, H)) return -1;
  int nf = (int)H.tris.size() / 3;
  if (nf > cap_f) return -2;
  for (int i = 0; i < 3 * nf; ++i) tris[i] = H.vert_ids[H.tris[i]];
  return nf;
}
