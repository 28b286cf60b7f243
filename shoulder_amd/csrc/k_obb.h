// k_obb.h -- oriented bounding box + head-end detection on the device
// (reference src/shoulder/humerus/mesh.py:63-125; trimesh `oriented_bounds` semantics restated in
// oracle/obb.py, canonical rule B-3).
//   k_obb_candidates  one workgroup per (hull face, humerus): project the hull on the face plane,
//                     find the silhouette edges (= edges of the 2-D hull of the projection), and
//                     for each take the enclosing rectangle -> min area x height = candidate volume
//   k_obb_pick        argmin volume -> axes ordered by ascending extent, signs fixed by vertex 0,
//                     box centred at the origin  -> T_pre (CT -> raw OBB)
//   k_obb_end_points  sections at 0.95*zmin / 0.95*zmax (mesh.py:91-99), crossing points only
//   k_obb_ends        circle-fit residual of both ends (mesh.py:102) -> flip (mesh.py:105-124)
// Hull record per humerus (host quickhull, sh_hull.h): hv [HV][3] f64 CT coords, normals [HF][3],
// edges [HE][4] = (va, vb, face f, face g).
#pragma once
#include "k_te.h"

namespace sh {

#define SH_HV 4096
#define SH_HF 8192
#define SH_HE 12288
#define SH_ENDCAP 1024

__device__ inline void obb_basis(const double* n, double* u, double* v) { plane_basis(n, u, v); }

__global__ void __launch_bounds__(256)
k_obb_candidates(const double* __restrict__ hv, const int* __restrict__ nv_, const double* __restrict__ normals, const int* __restrict__ nf_,
                 const int* __restrict__ edges, const int* __restrict__ ne_, double* __restrict__ cand_vol, int* __restrict__ cand_edge, int nvmax) {
  extern __shared__ double sm[];          // pu[nvmax], pv[nvmax]
  __shared__ int sil[2048];
  __shared__ int silv[2048];      // start vertex of the edge directed along its front face's winding
  __shared__ unsigned char front[SH_HF];
  __shared__ int nsil;
  __shared__ double red[8];
  __shared__ double b_area[4];
  __shared__ int b_edge[4];
  const int b = blockIdx.y, f = blockIdx.x, tid = threadIdx.x;
  const int nv = nv_[b], nf = nf_[b], ne = ne_[b];
  if (f >= nf) return;
  double* pu = sm;
  double* pv = sm + nvmax;
  const double* N = normals + ((size_t)b * SH_HF + f) * 3;
  double n[3] = {N[0], N[1], N[2]}, u[3], v[3];
  obb_basis(n, u, v);
  const double* P = hv + (size_t)b * SH_HV * 3;
  double hmin = 1e300, hmax = -1e300;
  for (int i = tid; i < nv; i += 256) {
    const double* p = P + 3 * i;
    pu[i] = dot3(p, u); pv[i] = dot3(p, v);
    double h = dot3(p, n);
    hmin = fmin(hmin, h); hmax = fmax(hmax, h);
  }
  for (int off = 32; off > 0; off >>= 1) { hmin = fmin(hmin, __shfl_down(hmin, off)); hmax = fmax(hmax, __shfl_down(hmax, off)); }
  if (tid == 0) nsil = 0;
  if ((tid & 63) == 0) { red[tid >> 6] = hmin; red[4 + (tid >> 6)] = hmax; }
  __syncthreads();
  // front/back flag of every hull face for this direction: one coalesced pass over the normals
  const int* E = edges + (size_t)b * SH_HE * 4;
  const double* NN = normals + (size_t)b * SH_HF * 3;
  for (int f2 = tid; f2 < nf; f2 += 256) front[f2] = dot3(NN + 3 * f2, n) > 0 ? 1 : 0;
  __syncthreads();
  // silhouette edges: the two incident faces see the direction n from opposite sides
  for (int e = tid; e < ne; e += 256) {
    const int4 ed = *(const int4*)(E + 4 * e);
    const int f1 = front[ed.z], f2 = front[ed.w];
    // (edge stored with face f's winding: va -> vb; directed along the FRONT face every silhouette vertex
    //  is the start of exactly one edge, so the start vertices enumerate the 2-D hull once)
    if (f1 != f2) { int s = atomicAdd(&nsil, 1); if (s < 2048) { sil[s] = e; silv[s] = f1 ? ed.x : ed.y; } }
  }
  __syncthreads();
  const int ns = nsil < 2048 ? nsil : 2048;
  double best = 1e300;
  int be = 0x7fffffff;
  for (int s = tid; s < ns; s += 256) {
    int e = sil[s];
    int a = E[4 * e], c = E[4 * e + 1];
    double ex = pu[c] - pu[a], ey = pv[c] - pv[a];
    double l = sqrt(ex * ex + ey * ey);
    if (l == 0.0) continue;
    ex /= l; ey /= l;
    // the end points of the silhouette edges are exactly the vertices of the projection's 2-D hull,
    // so the rectangle extents over them equal the extents over every projected hull vertex
    double amin = 1e300, amax = -1e300, bmin = 1e300, bmax = -1e300;
    for (int s2 = 0; s2 < ns; ++s2) {
      const int i = silv[s2];
      double x = pu[i], y = pv[i];
      double pa = x * ex + y * ey, pb = y * ex - x * ey;
      amin = fmin(amin, pa); amax = fmax(amax, pa); bmin = fmin(bmin, pb); bmax = fmax(bmax, pb);
    }
    double area = (amax - amin) * (bmax - bmin);
    if (area < best || (area == best && e < be)) { best = area; be = e; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    double ob = __shfl_down(best, off);
    int oe = __shfl_down(be, off);
    if (ob < best || (ob == best && oe < be)) { best = ob; be = oe; }
  }
  if ((tid & 63) == 0) { b_area[tid >> 6] = best; b_edge[tid >> 6] = be; }
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < 4; ++w) if (b_area[w] < best || (b_area[w] == best && b_edge[w] < be)) { best = b_area[w]; be = b_edge[w]; }
    double lo = fmin(fmin(red[0], red[1]), fmin(red[2], red[3])), hi = fmax(fmax(red[4], red[5]), fmax(red[6], red[7]));
    cand_vol[(size_t)b * SH_HF + f] = best * (hi - lo);
    cand_edge[(size_t)b * SH_HF + f] = be;
  }
}

// one workgroup (256) per humerus
__global__ void __launch_bounds__(256)
k_obb_pick(const double* __restrict__ hv, const double* __restrict__ normals, const int* __restrict__ nf_, const int* __restrict__ edges,
           const double* __restrict__ cand_vol, const int* __restrict__ cand_edge, const float* __restrict__ verts,
           const long long* __restrict__ voff, double* __restrict__ T_pre, double* __restrict__ zb_pre, int* __restrict__ err) {
  __shared__ double wv[4];
  __shared__ int wi[4];
  __shared__ double R[9];
  __shared__ double mm[6 * 4];
  const int b = blockIdx.x, tid = threadIdx.x;
  const int nf = nf_[b];
  double best = 1e300;
  int bf = 0x7fffffff;
  for (int f = tid; f < nf; f += 256) {
    double v = cand_vol[(size_t)b * SH_HF + f];
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
    if (bf == 0x7fffffff || !(best < 1e299)) { atomicExch(&err[b], SH_ERR_GEOMETRY_DEV); bf = 0; }
    const double* N = normals + ((size_t)b * SH_HF + bf) * 3;
    double n[3] = {N[0], N[1], N[2]}, u[3], v[3];
    obb_basis(n, u, v);
    int e = cand_edge[(size_t)b * SH_HF + bf];
    const int* E = edges + ((size_t)b * SH_HE + (e < 0 || e >= SH_HE ? 0 : e)) * 4;
    const double* pa = hv + ((size_t)b * SH_HV + E[0]) * 3;
    const double* pc = hv + ((size_t)b * SH_HV + E[1]) * 3;
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
                                 double* __restrict__ endpts /*[B][2][ENDCAP][2]*/, int* __restrict__ endcnt /*[B][2]*/) {
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
      if (slot < SH_ENDCAP) {
        double* p = endpts + (((size_t)b * 2 + e) * SH_ENDCAP + slot) * 2;
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
           double* __restrict__ resid /*[B][2]*/, double* __restrict__ T_obb, int* __restrict__ flipped, int* __restrict__ err, int B) {
  __shared__ double res[2];
  const int b = blockIdx.x, e = threadIdx.x >> 6, lane = threadIdx.x & 63;
  int n = endcnt[2 * b + e];
  double r;
  if (n > SH_ENDCAP) { if (lane == 0) atomicExch(&err[b], SH_ERR_CAPACITY_DEV); n = SH_ENDCAP; }
  if (n < 3) { if (lane == 0) atomicExch(&err[b], SH_ERR_GEOMETRY_DEV); r = 1e300; }
  else r = wave_circle_fit_residual(endpts + ((size_t)b * 2 + e) * SH_ENDCAP * 2, n);
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
