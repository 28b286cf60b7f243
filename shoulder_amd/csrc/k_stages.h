// k_stages.h -- per-mesh stages around the slice layer: synthetic batch, affine apply, surgical
// neck, canal.  Small fp64 work; one wave (or one lane) per humerus.
#pragma once
#include "k_slices.h"

namespace sh {

// BASELINE config 3/4: mesh i = T[i] * template (float64 arithmetic, stored float32 like an STL)
__global__ void k_synth_batch(const float* __restrict__ tv, const int* __restrict__ tf, long long V, long long F,
                              const double* __restrict__ T, float* __restrict__ verts, int* __restrict__ faces) {
  int b = blockIdx.y;
  const double* Tb = T + 16 * b;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < V; i += (long long)gridDim.x * blockDim.x) {
    double o[3];
    xform_pt(Tb, (double)tv[3 * i], (double)tv[3 * i + 1], (double)tv[3 * i + 2], o);
    float* q = verts + 3 * (V * b + i);
    q[0] = (float)o[0]; q[1] = (float)o[1]; q[2] = (float)o[2];
  }
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < 3 * F; i += (long long)gridDim.x * blockDim.x)
    faces[3 * F * b + i] = tf[i];
}

// utils.transform_pts (utils.py:172-188), B point sets, float64 xyz
__global__ void k_affine_f64(const double* __restrict__ T, const double* __restrict__ in, double* __restrict__ out,
                             const long long* __restrict__ off) {
  int b = blockIdx.y;
  const double* Tb = T + 16 * b;
  long long o0 = off[b], n = off[b + 1] - o0;
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const double* p = in + 3 * (o0 + i);
    double o[3];
    xform_pt(Tb, p[0], p[1], p[2], o);
    double* q = out + 3 * (o0 + i);
    q[0] = o[0]; q[1] = o[1]; q[2] = o[2];
  }
}

// Trimesh.apply_transform (bone.py:155) on float32 vertices -> float64
__global__ void k_affine_f32in(const double* __restrict__ T, const float* __restrict__ in, double* __restrict__ out, long long n) {
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    double o[3];
    xform_pt(T, (double)in[3 * i], (double)in[3 * i + 1], (double)in[3 * i + 2], o);
    out[3 * i] = o[0]; out[3 * i + 1] = o[1]; out[3 * i + 2] = o[2];
  }
}

// One general plane section of one mesh: unique crossing points in CT coordinates
// (`mesh_ct.section(plane_origin, plane_normal).vertices`, anatomic_neck.py:161-165).
// d = n.(v - o); sign rule and edge orientation as in k_slice_emit; one point per crossed edge.
__global__ void k_section_points(const float* __restrict__ verts, const int* __restrict__ faces, long long nf,
                                 const double* __restrict__ plane /*origin xyz, unit normal xyz*/, double* __restrict__ out, int cap,
                                 int* __restrict__ count) {
  const double ox = plane[0], oy = plane[1], oz = plane[2], nx = plane[3], ny = plane[4], nz = plane[5];
  for (long long fi = blockIdx.x * (long long)blockDim.x + threadIdx.x; fi < nf; fi += (long long)gridDim.x * blockDim.x) {
    int id[3] = {faces[3 * fi], faces[3 * fi + 1], faces[3 * fi + 2]};
    double P[3][3], d[3];
    int s[3];
    for (int k = 0; k < 3; ++k) {
      for (int q = 0; q < 3; ++q) P[k][q] = (double)verts[3 * (size_t)id[k] + q];
      d[k] = ((P[k][0] - ox) * nx + (P[k][1] - oy) * ny) + (P[k][2] - oz) * nz;
      s[k] = d[k] < -SH_SECTION_TOL ? -1 : 1;
    }
    if (s[0] == s[1] && s[1] == s[2]) continue;
    int dn = 0;
    for (int j = 0; j < 3; ++j) if (s[j] == 1 && s[(j + 1) % 3] == -1) dn = j;
    int a = dn, c = (dn + 1) % 3;
    int l = id[a] < id[c] ? a : c, h = id[a] < id[c] ? c : a;
    double t = d[l] / (d[l] - d[h]);
    int slot = atomicAdd(count, 1);
    if (slot < cap)
      for (int q = 0; q < 3; ++q) out[3 * (size_t)slot + q] = P[l][q] + t * (P[h][q] - P[l][q]);
  }
}

// k-th smallest (0-based) of n NON-NEGATIVE doubles by one 64-lane workgroup: radix select on the bit patterns (which order
// like the values), one byte per pass -- the same value any selection algorithm returns (sh::kth_smallest on the host).
// hist: 256 counters + 2 words in LDS.
__device__ inline double wave_kth_smallest_nonneg(const double* v, int n, int k, unsigned* hist) {
  const int lane = threadIdx.x & 63;
  unsigned long long prefix = 0, mask = 0;
  for (int shift = 56; shift >= 0; shift -= 8) {
    for (int i = lane; i < 256; i += 64) hist[i] = 0;
    __syncthreads();
    for (int i = lane; i < n; i += 64) {
      const unsigned long long u = (unsigned long long)__double_as_longlong(v[i]);
      if ((u & mask) == prefix) atomicAdd(&hist[(u >> shift) & 255], 1u);
    }
    __syncthreads();
    const unsigned c0 = hist[4 * lane], c1 = hist[4 * lane + 1], c2 = hist[4 * lane + 2], c3 = hist[4 * lane + 3];
    const unsigned tot = c0 + c1 + c2 + c3;
    unsigned incl = tot;
    for (int off = 1; off < 64; off <<= 1) { unsigned o = __shfl_up(incl, off); if (lane >= off) incl += o; }
    const unsigned excl = incl - tot;
    if ((unsigned)k >= excl && (unsigned)k < incl) {      // exactly one lane owns the bin that holds rank k
      unsigned r = (unsigned)k - excl, bin = 4 * lane;
      if (r >= c0) { r -= c0; ++bin; if (r >= c1) { r -= c1; ++bin; if (r >= c2) { r -= c2; ++bin; } } }
      hist[256] = bin; hist[257] = r;
    }
    __syncthreads();
    prefix |= (unsigned long long)hist[256] << shift;
    mask |= 0xffull << shift;
    k = (int)hist[257];
    __syncthreads();
  }
  return __longlong_as_double((long long)prefix);
}

// surgical_neck.py:22-34: KernelCPD on areas1((0.70,0.99)) -> neck_z = zs_cut[bkp]
// One workgroup (64 lanes) per humerus: gamma = 1 / median of the pairwise squared distances (sh::cpd_gamma) by a
// radix select over all lanes, all lanes fill the Gram matrix, each lane evaluates the cost of some breakpoints with
// the same arithmetic as sh::cpd_one_bkp, first minimum wins.
template <bool BIG>      // BIG: the cut is longer than SH_CPD_MAXN samples, the Gram matrix lives in global scratch (proximal humeri)
__global__ void __launch_bounds__(64)
k_neck(const double* __restrict__ areas, const double* __restrict__ zs, double* __restrict__ neck_z, int* __restrict__ neck_index, int B,
       double c0, double c1, double* __restrict__ gscratch /*[B][SH_NFULL^2]*/) {
  __shared__ double Ks[BIG ? 1 : SH_CPD_MAXN * SH_CPD_MAXN];
  __shared__ unsigned hist[258];
  int b = blockIdx.x, lane = threadIdx.x;
  int a, e;
  cutoff_range(SH_NFULL, c0, c1, &a, &e);      // (0.70, 0.99) for a whole humerus, (0.2, 0.99) for a proximal one (surgical_neck.py:25-28)
  int n = e - a;
  if (!BIG && n > SH_CPD_MAXN) n = SH_CPD_MAXN;
  double* K;
  if (BIG) K = gscratch + (size_t)b * SH_NFULL * SH_NFULL; else K = Ks;
  const double* x = areas + (size_t)b * SH_NFULL + a;
  // pairwise squared distances in K (free until the Gram fill): row i holds j > i
  for (int i = 0; i < n; ++i) {
    const int base = i * n - i * (i + 1) / 2 - (i + 1);      // index of (i, j) = base + j
    for (int j = i + 1 + lane; j < n; j += 64) { double d = x[i] - x[j]; K[base + j] = d * d; }
  }
  __syncthreads();
  const int np_ = n * (n - 1) / 2;
  double med;
  __threadfence_block();
  if (np_ & 1) med = wave_kth_smallest_nonneg(K, np_, np_ / 2, hist);
  else { double lo = wave_kth_smallest_nonneg(K, np_, np_ / 2 - 1, hist); double hi = wave_kth_smallest_nonneg(K, np_, np_ / 2, hist); med = (lo + hi) / 2.0; }
  const double gamma = (med == 0.0) ? 1.0 : 1.0 / med;
  __syncthreads();
  for (int q = lane; q < n * n; q += 64) K[q] = cpd_kernel(x[q / n], x[q % n], gamma);
  __syncthreads();
  double best = 1e300;
  int bt = 0x7fffffff;
  for (int t = 2 + lane; t <= n - 2; t += 64) {
    double cst = cpd_cost(K, n, t);
    if (cst < best) { best = cst; bt = t; }
  }
  for (int off = 32; off > 0; off >>= 1) {
    double ob = __shfl_down(best, off);
    int ot = __shfl_down(bt, off);
    if (ob < best || (ob == best && ot < bt)) { best = ob; bt = ot; }
  }
  if (lane == 0) {
    neck_index[b] = bt;
    neck_z[b] = zs[(size_t)b * SH_NFULL + a + bt];
  }
}

// mesh.py:163-192 (ProxObb): head = largest section area -> flip so that it is at +z; canal range from the (reversed if
// flipped) area profile; T_obb = flip * T_pre.  One lane per humerus.
__global__ void k_prox_obb(const double* __restrict__ area_scan /*[B][SH_NPSCAN]*/, const double* __restrict__ scan_zs, const double* __restrict__ T_pre,
                           double* __restrict__ T_obb, int* __restrict__ flipped, double* __restrict__ cutoff /*[B][2]*/, int* __restrict__ cutoff_idx /*[B][2]*/,
                           int* __restrict__ err, int B) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double* A = area_scan + (size_t)b * SH_NPSCAN;
  int k = 0;
  for (int j = 1; j < SH_NPSCAN; ++j) if (A[j] > A[k]) k = j;      // np.argmax: first maximum
  const bool flip = scan_zs[(size_t)b * SH_NPSCAN + k] < 0;         // humeral_head_z < 0
  double ar[SH_NPSCAN], tmp[2 * SH_NPSCAN];
  for (int j = 0; j < SH_NPSCAN; ++j) ar[j] = flip ? A[SH_NPSCAN - 1 - j] : A[j];      // z_area[::-1]
  int lo, hi;
  if (prox_canal_range(ar, SH_NPSCAN, tmp, &lo, &hi) < 1) atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV);
  cutoff_idx[2 * b] = lo; cutoff_idx[2 * b + 1] = hi;
  cutoff[2 * b] = (double)lo / (double)SH_NPSCAN;
  cutoff[2 * b + 1] = (double)hi / (double)SH_NPSCAN;
  flipped[b] = flip ? 1 : 0;
  const double* Tp = T_pre + 16 * b;
  double* To = T_obb + 16 * b;
  for (int q = 0; q < 16; ++q) To[q] = Tp[q];
  if (flip) for (int q = 0; q < 4; ++q) { To[q] = -Tp[q]; To[8 + q] = -Tp[8 + q]; }     // diag(-1,1,-1,1) * T_pre
}

__device__ inline double wave_sum(double v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off);
  return __shfl(v, 0);
}

// Mean + covariance of n xyz points by one wave (lane-strided loads, shuffle reduction), then
// the dominant direction by power iteration (K11 `Line.best_fit`, canal.py:66).
__device__ inline void wave_line_fit(const double* p, int n, int stride, double* mean, double* dir) {
  int lane = threadIdx.x & 63;
  double s[3] = {0, 0, 0};
  for (int i = lane; i < n; i += 64) { s[0] += p[(size_t)i * stride]; s[1] += p[(size_t)i * stride + 1]; s[2] += p[(size_t)i * stride + 2]; }
  for (int k = 0; k < 3; ++k) mean[k] = wave_sum(s[k]) / (double)n;
  double c[6] = {0, 0, 0, 0, 0, 0};
  for (int i = lane; i < n; i += 64) {
    double x = p[(size_t)i * stride] - mean[0], y = p[(size_t)i * stride + 1] - mean[1], z = p[(size_t)i * stride + 2] - mean[2];
    c[0] += x * x; c[1] += x * y; c[2] += x * z; c[3] += y * y; c[4] += y * z; c[5] += z * z;
  }
  for (int k = 0; k < 6; ++k) c[k] = wave_sum(c[k]);
  double C[9] = {c[0], c[1], c[2], c[1], c[3], c[4], c[2], c[4], c[5]};
  dominant_eigvec3(C, dir);   // every lane computes the same 3x3 problem
}

// canal.py:19-85
#define SH_CANAL_MAXPTS SH_NFULL
__global__ void k_canal(const double* __restrict__ centroids, const double* __restrict__ zs, const double* __restrict__ zb,
                        const double* __restrict__ T_obb, double c0, double c1, const double* __restrict__ cut /*nullable [B][2]: ProxObb.cutoff_pcts (canal.py:33-38)*/,
                        double* __restrict__ pts_obb, double* __restrict__ axis_obb, double* __restrict__ axis_ct, int* __restrict__ err) {
  int b = blockIdx.x, lane = threadIdx.x;
  if (cut) { c0 = cut[2 * b]; c1 = cut[2 * b + 1]; }
  int a, e;
  cutoff_range(SH_NFULL, c0, c1, &a, &e);
  int n = e - a;
  if (n > SH_CANAL_MAXPTS) n = SH_CANAL_MAXPTS;
  if (n < 2) { if (lane == 0) { atomicCAS(&err[b], 0, SH_ERR_GEOMETRY_DEV); for (int k = 0; k < 6; ++k) { axis_obb[6 * b + k] = 0; axis_ct[6 * b + k] = 0; } } return; }
  double* P = pts_obb + (size_t)b * SH_CANAL_MAXPTS * 3;
  for (int i = lane; i < n; i += 64) {
    P[3 * i] = centroids[2 * ((size_t)b * SH_NFULL + a + i)];
    P[3 * i + 1] = centroids[2 * ((size_t)b * SH_NFULL + a + i) + 1];
    P[3 * i + 2] = zs[(size_t)b * SH_NFULL + a + i];
  }
  __syncthreads();
  double mean[3], d[3];
  wave_line_fit(P, n, 3, mean, d);
  if (lane == 0) {
    if (d[2] < 0) { d[0] = -d[0]; d[1] = -d[1]; d[2] = -d[2]; }
    double z_length = fabs(zb[2 * b]) + fabs(zb[2 * b + 1]);
    double half = (z_length * ((c0 + c1) / 2.0)) / 2.0;
    double* A = axis_obb + 6 * b;
    for (int k = 0; k < 3; ++k) { A[k] = mean[k] + d[k] * half; A[3 + k] = mean[k] - d[k] * half; }
    double Ti[16];
    inv_transform(T_obb + 16 * b, Ti);
    xform_pt(Ti, A[0], A[1], A[2], axis_ct + 6 * b);
    xform_pt(Ti, A[3], A[4], A[5], axis_ct + 6 * b + 3);
  }
}

}  // namespace sh
