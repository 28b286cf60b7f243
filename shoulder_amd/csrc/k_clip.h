// k_clip.h -- cut a triangle mesh with planes and keep the positive side (SURVEY 8(f) rank 4:
// `HumeralHeadOsteotomy.resect_mesh`, reference src/shoulder/arthroplasty.py:80-87 ->
// `trimesh.Trimesh.slice_plane(origin, normal)` = trimesh.intersections.slice_faces_plane + the vertex merge of the
// Trimesh constructor; restated in oracle/clip.py, which states the canonical ordering rule).
// One mesh, P planes (grid.y = plane): a sweep of resection planes over one humerus is one pass.
//   k_clip_sign     per vertex: side of the plane (|dot| <= 1e-8 -> on the plane)
//   k_clip_class    per face: dropped / kept whole / cut to a quad (two vertices kept) / cut to a triangle (one kept);
//                   one workgroup per plane scans the classes -> position of every face inside its class, class totals
//   k_clip_emit     kept faces, the two triangles of every quad, the triangle of every cut-to-triangle face, with the
//                   two new crossing points of every cut face (float64, operation order of the numpy code);
//                   the cut edge (new0, new1) of every cut face -> the section polyline comes for free
//   k_clip_mark     which pre-merge vertices a face uses
//   k_clip_hash     key = round(v * 1e8) as 3 x int64 (trimesh merges at 8 decimals), open-addressing insert
//   k_clip_rank / k_clip_out   first referenced vertex of a key wins; renumber by a block scan; write out
#pragma once
#include "sh_common.h"
#include "k_stl.h"

namespace sh {

#define SH_CLIP_TOL 1e-8      // trimesh.constants.tol.merge

struct ClipCounts { int n_in, n_quad, n_tri, n_pre, n_verts, n_faces, n_edges, pad; };      // per plane

__global__ void k_clip_sign(const double* __restrict__ verts, int nv, const double* __restrict__ planes /* P x (origin, normal) */,
                            signed char* __restrict__ sign /* P x nv */) {
  const int p = blockIdx.y;
  const double* pl = planes + 6 * p;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += gridDim.x * blockDim.x) {
    const double dx = verts[3 * i] - pl[0], dy = verts[3 * i + 1] - pl[1], dz = verts[3 * i + 2] - pl[2];
    const double d = (dx * pl[3] + dy * pl[4]) + dz * pl[5];
    sign[(size_t)p * nv + i] = d < -SH_CLIP_TOL ? 1 : (d > SH_CLIP_TOL ? -1 : 0);      // -1 = kept side (slice_faces_plane's convention)
  }
}

// class of a face from its three signs: 0 dropped, 1 kept, 2 quad, 3 triangle, 4 = lies in the plane (decided by its normal)
__device__ inline int clip_class(int s0, int s1, int s2) {
  const int sum = s0 + s1 + s2, asum = abs(s0) + abs(s1) + abs(s2);
  if (asum == 0) return 4;
  if (asum >= 2 && abs(sum) <= 1) return sum < 0 ? 2 : 3;
  return sum == -asum ? 1 : 0;
}

__global__ void __launch_bounds__(SH_STL_SCAN_THREADS)
k_clip_class(const double* __restrict__ verts, const int* __restrict__ faces, int nf, int nv, const double* __restrict__ planes,
             const signed char* __restrict__ sign, unsigned char* __restrict__ cls /* P x nf */, int* __restrict__ fpos /* P x nf */,
             ClipCounts* __restrict__ counts) {
  __shared__ int s_wave[SH_STL_SCAN_THREADS / 64];
  const int p = blockIdx.x, tid = threadIdx.x;
  const signed char* S = sign + (size_t)p * nv;
  const double* pl = planes + 6 * p;
  unsigned char* C = cls + (size_t)p * nf;
  int* FP = fpos + (size_t)p * nf;
  const int per = (nf + SH_STL_SCAN_THREADS - 1) / SH_STL_SCAN_THREADS;
  const int a = min(nf, tid * per), e = min(nf, a + per);
  int c1 = 0, c2 = 0, c3 = 0;
  for (int f = a; f < e; ++f) {
    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
    int k = clip_class(S[i0], S[i1], S[i2]);
    if (k == 4) {      // in the plane: kept iff not degenerate and facing away from the plane normal
      const double* A = verts + 3 * (size_t)i0; const double* B = verts + 3 * (size_t)i1; const double* Cc = verts + 3 * (size_t)i2;
      const double ux = B[0] - A[0], uy = B[1] - A[1], uz = B[2] - A[2], vx = Cc[0] - A[0], vy = Cc[1] - A[1], vz = Cc[2] - A[2];
      const double nx = uy * vz - uz * vy, ny = uz * vx - ux * vz, nz = ux * vy - uy * vx;
      const double nn = sqrt((nx * nx + ny * ny) + nz * nz);
      k = (nn > 1e-13 && ((nx / nn) * pl[3] + (ny / nn) * pl[4]) + (nz / nn) * pl[5] < 0.0) ? 1 : 0;
    }
    C[f] = (unsigned char)k;
    c1 += k == 1; c2 += k == 2; c3 += k == 3;
  }
  int t1, t2, t3;
  int p1 = stl_block_scan(c1, s_wave, &t1);
  int p2 = stl_block_scan(c2, s_wave, &t2);
  int p3 = stl_block_scan(c3, s_wave, &t3);
  for (int f = a; f < e; ++f) {
    const int k = C[f];
    FP[f] = k == 1 ? p1++ : k == 2 ? p2++ : k == 3 ? p3++ : -1;
  }
  if (tid == 0) {
    ClipCounts& o = counts[p];
    o.n_in = t1; o.n_quad = t2; o.n_tri = t3; o.n_pre = nv + 2 * t2 + 2 * t3; o.n_verts = 0; o.n_faces = t1 + 2 * t2 + t3; o.n_edges = t2 + t3; o.pad = 0;
  }
}

// crossing point of edge j (o[j] -> o[(j+1)%3]) with the plane: slice_faces_plane's  dist = num / denom;  point = dist * d + o
__device__ inline void clip_cross(const double* O /* 3 x 3 */, int j, const double* pl, double* out) {
  const double* o = O + 3 * j; const double* o1 = O + 3 * ((j + 1) % 3);
  const double dx = o1[0] - o[0], dy = o1[1] - o[1], dz = o1[2] - o[2];
  const double num = ((pl[0] - o[0]) * pl[3] + (pl[1] - o[1]) * pl[4]) + (pl[2] - o[2]) * pl[5];
  double den = (dx * pl[3] + dy * pl[4]) + dz * pl[5];
  if (den == 0.0) den = 1e-12;
  const double dist = num / den;
  out[0] = dist * dx + o[0]; out[1] = dist * dy + o[1]; out[2] = dist * dz + o[2];
}

// pre-merge numbering: original vertices [0, nv), quad points nv + 2q + {0,1}, triangle points nv + 2 n_quad + 2t + {0,1}
__global__ void k_clip_emit(const double* __restrict__ verts, const int* __restrict__ faces, int nf, int nv, const double* __restrict__ planes,
                            const signed char* __restrict__ sign, const unsigned char* __restrict__ cls, const int* __restrict__ fpos,
                            const ClipCounts* __restrict__ counts, const long long* __restrict__ new_off /* P+1 */, const long long* __restrict__ face_off,
                            const long long* __restrict__ edge_off, double* __restrict__ new_pts, int* __restrict__ pre_faces, int* __restrict__ pre_edges) {
  const int p = blockIdx.y;
  const ClipCounts cn = counts[p];
  const double* pl = planes + 6 * p;
  const signed char* S = sign + (size_t)p * nv;
  double* NP = new_pts + 3 * new_off[p];
  int* F = pre_faces + 3 * face_off[p];
  int* E = pre_edges + 2 * edge_off[p];
  for (int f = blockIdx.x * blockDim.x + threadIdx.x; f < nf; f += gridDim.x * blockDim.x) {
    const int k = cls[(size_t)p * nf + f];
    if (k == 0) continue;
    const int pos = fpos[(size_t)p * nf + f];
    const int id[3] = {faces[3 * f], faces[3 * f + 1], faces[3 * f + 2]};
    if (k == 1) { F[3 * pos] = id[0]; F[3 * pos + 1] = id[1]; F[3 * pos + 2] = id[2]; continue; }
    double O[9];
    for (int j = 0; j < 3; ++j) for (int d = 0; d < 3; ++d) O[3 * j + d] = verts[3 * (size_t)id[j] + d];
    if (k == 2) {      // one vertex cut away
      const int qi = S[id[0]] == 1 ? 0 : (S[id[1]] == 1 ? 1 : 2);
      const int n0 = nv + 2 * pos, n1 = n0 + 1;
      clip_cross(O, (qi + 2) % 3, pl, NP + 3 * (size_t)(2 * pos));
      clip_cross(O, qi, pl, NP + 3 * (size_t)(2 * pos + 1));
      const int a = id[(qi + 1) % 3], b = id[(qi + 2) % 3];
      int* f0 = F + 3 * (size_t)(cn.n_in + pos); int* f1 = F + 3 * (size_t)(cn.n_in + cn.n_quad + pos);      // triangulate_quads: [0,1,2] block, then [2,3,0] block
      f0[0] = a; f0[1] = b; f0[2] = n0;
      f1[0] = n0; f1[1] = n1; f1[2] = a;
      E[2 * pos] = n0; E[2 * pos + 1] = n1;
    } else {           // one vertex kept
      const int ti = S[id[0]] == -1 ? 0 : (S[id[1]] == -1 ? 1 : 2);
      const int n0 = nv + 2 * cn.n_quad + 2 * pos, n1 = n0 + 1;
      clip_cross(O, ti, pl, NP + 3 * (size_t)(2 * cn.n_quad + 2 * pos));
      clip_cross(O, (ti + 2) % 3, pl, NP + 3 * (size_t)(2 * cn.n_quad + 2 * pos + 1));
      int* f0 = F + 3 * (size_t)(cn.n_in + 2 * cn.n_quad + pos);
      f0[0] = id[ti]; f0[1] = n0; f0[2] = n1;
      E[2 * (cn.n_quad + pos)] = n0; E[2 * (cn.n_quad + pos) + 1] = n1;
    }
  }
}

__device__ inline const double* clip_pre_vertex(const double* verts, const double* NP, int nv, int i) {
  return i < nv ? verts + 3 * (size_t)i : NP + 3 * (size_t)(i - nv);
}

// referenced[i] = 1 for every pre-merge vertex a face uses
__global__ void k_clip_mark(const int* __restrict__ pre_faces, const long long* __restrict__ face_off, const long long* __restrict__ pre_off,
                            unsigned char* __restrict__ referenced) {
  const int p = blockIdx.y;
  const long long n = 3 * (face_off[p + 1] - face_off[p]);
  const int* F = pre_faces + 3 * face_off[p];
  unsigned char* R = referenced + pre_off[p];
  for (long long i = blockIdx.x * (long long)blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) R[F[i]] = 1;
}

__device__ inline unsigned clip_hash(long long x, long long y, long long z) {
  unsigned long long k = (unsigned long long)x * 0x9E3779B97F4A7C15ull ^ ((unsigned long long)y * 0xC2B2AE3D27D4EB4Full) ^ ((unsigned long long)z * 0x165667B19E3779F9ull);
  k ^= k >> 29; k *= 0xBF58476D1CE4E5B9ull; k ^= k >> 32;
  return (unsigned)k;
}

// keys + hash insert of every referenced pre-merge vertex; table entry = (owner, smallest referenced index with this key)
__global__ void k_clip_hash(const double* __restrict__ verts, int nv, const double* __restrict__ new_pts, const long long* __restrict__ new_off,
                            const long long* __restrict__ pre_off, const unsigned char* __restrict__ referenced, long long* __restrict__ keys,
                            int2* __restrict__ table, const long long* __restrict__ tab_off, int* __restrict__ slot_of) {
  const int p = blockIdx.y;
  const int n = (int)(pre_off[p + 1] - pre_off[p]);
  const int tsize = (int)(tab_off[p + 1] - tab_off[p]);
  const double* NP = new_pts + 3 * new_off[p];
  long long* K = keys + 3 * pre_off[p];
  int2* T = table + tab_off[p];
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    if (!referenced[pre_off[p] + i]) continue;
    const double* v = clip_pre_vertex(verts, NP, nv, i);
    const long long x = (long long)rint(v[0] * 1e8), y = (long long)rint(v[1] * 1e8), z = (long long)rint(v[2] * 1e8);
    K[3 * (size_t)i] = x; K[3 * (size_t)i + 1] = y; K[3 * (size_t)i + 2] = z;
    __threadfence();      // the key is visible before the slot can name this vertex as its owner
    unsigned h = clip_hash(x, y, z) & (unsigned)(tsize - 1);
    for (;;) {
      const int o = atomicCAS(&T[h].x, -1, i);
      bool same = o == -1;
      if (!same) {
        const volatile long long* Ko = K + 3 * (size_t)o;
        same = Ko[0] == x && Ko[1] == y && Ko[2] == z;
      }
      if (same) { atomicMin(&T[h].y, i); slot_of[pre_off[p] + i] = (int)h; break; }
      h = (h + 1) & (unsigned)(tsize - 1);
    }
  }
}

// new id of every referenced pre-merge vertex: rank of its representative among the representatives (by pre-merge index)
__global__ void __launch_bounds__(SH_STL_SCAN_THREADS)
k_clip_rank(const long long* __restrict__ pre_off, const unsigned char* __restrict__ referenced, const int2* __restrict__ table,
            const long long* __restrict__ tab_off, const int* __restrict__ slot_of, int* __restrict__ vid, ClipCounts* __restrict__ counts) {
  __shared__ int s_wave[SH_STL_SCAN_THREADS / 64];
  const int p = blockIdx.x, tid = threadIdx.x;
  const long long b0 = pre_off[p];
  const int n = (int)(pre_off[p + 1] - b0);
  const int2* T = table + tab_off[p];
  const int per = (n + SH_STL_SCAN_THREADS - 1) / SH_STL_SCAN_THREADS;
  const int a = min(n, tid * per), e = min(n, a + per);
  int cnt = 0;
  for (int i = a; i < e; ++i) cnt += (referenced[b0 + i] && T[slot_of[b0 + i]].y == i) ? 1 : 0;
  int V;
  int pos = stl_block_scan(cnt, s_wave, &V);
  for (int i = a; i < e; ++i) if (referenced[b0 + i] && T[slot_of[b0 + i]].y == i) vid[b0 + i] = pos++;
  __syncthreads();
  __threadfence_block();
  for (int i = a; i < e; ++i)
    if (referenced[b0 + i]) { const int r = T[slot_of[b0 + i]].y; if (r != i) vid[b0 + i] = vid[b0 + r]; }
  if (tid == 0) counts[p].n_verts = V;
}

__global__ void k_clip_out(const double* __restrict__ verts, int nv, const double* __restrict__ new_pts, const long long* __restrict__ new_off,
                           const long long* __restrict__ pre_off, const unsigned char* __restrict__ referenced, const int2* __restrict__ table,
                           const long long* __restrict__ tab_off, const int* __restrict__ slot_of, const int* __restrict__ vid,
                           const int* __restrict__ pre_faces, const long long* __restrict__ face_off, const int* __restrict__ pre_edges,
                           const long long* __restrict__ edge_off, double* __restrict__ out_verts, int cap_v, int* __restrict__ out_faces, int cap_f,
                           int* __restrict__ out_edges, int cap_e) {
  const int p = blockIdx.y;
  const long long b0 = pre_off[p];
  const int n = (int)(pre_off[p + 1] - b0);
  const double* NP = new_pts + 3 * new_off[p];
  const int2* T = table + tab_off[p];
  const int* VID = vid + b0;
  const int gtid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
  for (int i = gtid; i < n; i += gsz)
    if (referenced[b0 + i] && T[slot_of[b0 + i]].y == i) {
      const double* v = clip_pre_vertex(verts, NP, nv, i);
      double* o = out_verts + 3 * ((size_t)p * cap_v + VID[i]);
      o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
    }
  const long long nfc = 3 * (face_off[p + 1] - face_off[p]);
  const int* F = pre_faces + 3 * face_off[p];
  for (long long i = gtid; i < nfc; i += gsz) out_faces[3 * (size_t)p * cap_f + i] = VID[F[i]];
  if (out_edges) {
    const long long nec = 2 * (edge_off[p + 1] - edge_off[p]);
    const int* E = pre_edges + 2 * edge_off[p];
    for (long long i = gtid; i < nec; i += gsz) out_edges[2 * (size_t)p * cap_e + i] = VID[E[i]];
  }
}

}  // namespace sh
