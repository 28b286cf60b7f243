// sh_hull.h -- 3-D convex hull (quickhull with conflict lists), host side of SH_STAGE_OBB.
//
// The reference reaches the hull through trimesh -> scipy -> qhull inside `Trimesh.apply_obb()`
// (src/shoulder/humerus/mesh.py:82).  The hull is the one irregular, pointer-chasing step of the
// path; it runs on the host (one mesh per worker thread) and hands the device a compact record:
// hull vertices, triangles with unit normals and the edge list used for silhouette tests
// (k_obb.h).  Points closer than eps = 1e-10 * bbox diagonal to the current hull count as inside.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <vector>

namespace shhull {

struct Face {
  int v[3];
  int nb[3];          // neighbour across edge v[i] -> v[(i+1)%3]
  double n[3], d;     // unit normal, plane offset: n.p = d
  double area2 = 0.0; // |cross product| = twice the area (a sliver's normal is noise)
  int head = -1;      // conflict list (points outside this face), linked through Builder::next
  int far_pt = -1;    // farthest point of the list and its distance
  double far_d = 0.0;
  bool alive = true;
  int mark = 0;
};

struct Hull {
  std::vector<int> vert_ids;         // indices into the input points
  std::vector<int> tris;             // 3 per face, indices into vert_ids
  std::vector<double> normals;       // 3 per face, outward unit
  std::vector<int> edges;            // 4 per edge: va, vb (hull vertex ids), face f, face g
};

struct Builder {
  const double* P;
  int n;
  double eps;
  std::vector<Face> F;
  std::vector<int> next;             // intrusive conflict lists: no per-face allocations

  double dist(const Face& f, int p) const { return f.n[0] * P[3 * p] + f.n[1] * P[3 * p + 1] + f.n[2] * P[3 * p + 2] - f.d; }

  void push(int f, int p, double d) {
    Face& fc = F[f];
    next[p] = fc.head;
    fc.head = p;
    if (fc.far_pt < 0 || d > fc.far_d) { fc.far_pt = p; fc.far_d = d; }
  }

  bool set_plane(Face& f) {
    const double* a = P + 3 * f.v[0]; const double* b = P + 3 * f.v[1]; const double* c = P + 3 * f.v[2];
    double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
    double nx = u[1] * w[2] - u[2] * w[1], ny = u[2] * w[0] - u[0] * w[2], nz = u[0] * w[1] - u[1] * w[0];
    double l = std::sqrt(nx * nx + ny * ny + nz * nz);
    f.area2 = l;
    if (l == 0.0) { f.n[0] = f.n[1] = 0; f.n[2] = 1; f.d = a[2]; return false; }
    f.n[0] = nx / l; f.n[1] = ny / l; f.n[2] = nz / l;
    f.d = f.n[0] * a[0] + f.n[1] * a[1] + f.n[2] * a[2];
    return true;
  }

  int add_face(int a, int b, int c) {
    Face f;
    f.v[0] = a; f.v[1] = b; f.v[2] = c;
    f.nb[0] = f.nb[1] = f.nb[2] = -1;
    set_plane(f);
    F.push_back(f);
    return (int)F.size() - 1;
  }
};

// pts: n x 3 doubles (any frame).  Returns false for degenerate (flat) input.
inline int& hull_fail_reason() { static thread_local int r = 0; return r; }
inline bool fail_(int why) { hull_fail_reason() = why; return false; }

inline bool convex_hull_eps(const double* pts_in, int n, Hull& H, double eps_rel) {
  if (n < 4) return fail_(1);
  // centre the cloud (smaller magnitudes -> smaller plane-distance rounding)
  std::vector<double> Pc(3 * (size_t)n);
  double c[3] = {0, 0, 0}, lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) { double v = pts_in[3 * i + k]; c[k] += v; lo[k] = std::min(lo[k], v); hi[k] = std::max(hi[k], v); }
  for (int k = 0; k < 3; ++k) c[k] /= n;
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) Pc[3 * (size_t)i + k] = pts_in[3 * i + k] - c[k];
  Builder B;
  B.P = Pc.data(); B.n = n;
  B.next.assign(n, -1);
  B.F.reserve(8192);
  double diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
  B.eps = eps_rel * diag;
  const double* P = B.P;
  // initial simplex: extreme pair along x, farthest from that line, farthest from that plane
  int i0 = 0, i1 = 0;
  for (int i = 1; i < n; ++i) { if (P[3 * i] < P[3 * i0]) i0 = i; if (P[3 * i] > P[3 * i1]) i1 = i; }
  if (i0 == i1) return fail_(2);
  auto sub = [&](int a, int b, double* o) { for (int k = 0; k < 3; ++k) o[k] = P[3 * a + k] - P[3 * b + k]; };
  double e[3]; sub(i1, i0, e);
  int i2 = -1; double best = 0;
  for (int i = 0; i < n; ++i) {
    double w[3]; sub(i, i0, w);
    double cx = e[1] * w[2] - e[2] * w[1], cy = e[2] * w[0] - e[0] * w[2], cz = e[0] * w[1] - e[1] * w[0];
    double a2 = cx * cx + cy * cy + cz * cz;
    if (a2 > best) { best = a2; i2 = i; }
  }
  if (i2 < 0) return fail_(3);
  Face tmp; tmp.v[0] = i0; tmp.v[1] = i1; tmp.v[2] = i2; B.set_plane(tmp);
  int i3 = -1; best = 0;
  for (int i = 0; i < n; ++i) { double d = std::fabs(B.dist(tmp, i)); if (d > best) { best = d; i3 = i; } }
  if (i3 < 0 || best <= B.eps) return fail_(4);
  if (B.dist(tmp, i3) > 0) std::swap(i1, i2);        // make (i0,i1,i2) face away from i3
  int f0 = B.add_face(i0, i1, i2), f1 = B.add_face(i0, i3, i1), f2 = B.add_face(i1, i3, i2), f3 = B.add_face(i2, i3, i0);
  auto link = [&](int f, int ei, int g) { B.F[f].nb[ei] = g; };
  // f0: (i0,i1),(i1,i2),(i2,i0) ; f1: (i0,i3),(i3,i1),(i1,i0) ; f2: (i1,i3),(i3,i2),(i2,i1) ; f3: (i2,i3),(i3,i0),(i0,i2)
  link(f0, 0, f1); link(f0, 1, f2); link(f0, 2, f3);
  link(f1, 0, f3); link(f1, 1, f2); link(f1, 2, f0);
  link(f2, 0, f1); link(f2, 1, f3); link(f2, 2, f0);
  link(f3, 0, f2); link(f3, 1, f1); link(f3, 2, f0);
  for (int p = 0; p < n; ++p) {
    if (p == i0 || p == i1 || p == i2 || p == i3) continue;
    for (int f = 0; f < 4; ++f) {
      double d = B.dist(B.F[f], p);
      if (d > B.eps) { B.push(f, p, d); break; }
    }
  }
  std::vector<int> stack, visible, horizon_f, horizon_e, pending, starts, ends, dropped, seen_start(n, 0), seen_end(n, 0);
  for (int f = 0; f < 4; ++f) if (B.F[f].head >= 0) pending.push_back(f);
  int stamp = 0, vstamp = 0;
  // visible set of p by flood fill from f (faces farther than `thr` in front of p), horizon = its boundary edges.
  // false: an adjacency slot was empty.
  auto flood = [&](int f, int p, double thr) -> bool {
    ++stamp;
    visible.clear(); horizon_f.clear(); horizon_e.clear();
    stack.clear(); stack.push_back(f); B.F[f].mark = stamp;
    while (!stack.empty()) {
      int g = stack.back(); stack.pop_back();
      visible.push_back(g);
      for (int ei = 0; ei < 3; ++ei) {
        int h = B.F[g].nb[ei];
        if (h < 0 || h >= (int)B.F.size() || !B.F[h].alive) return fail_(5);
        if (B.F[h].mark == stamp) continue;
        if (B.dist(B.F[h], p) > thr) { B.F[h].mark = stamp; stack.push_back(h); }
        else { horizon_f.push_back(g); horizon_e.push_back(ei); }
      }
    }
    return true;
  };
  // In exact arithmetic the horizon of an outside point is one simple closed loop.  With a tolerance, nearly coplanar
  // neighbours can pinch it (a vertex starts two horizon edges) or open it; the new faces could then not be linked.
  auto horizon_is_simple_loop = [&]() -> bool {
    const size_t m = horizon_f.size();
    if (m < 3) return fail_(6);
    ++vstamp;
    for (size_t k = 0; k < m; ++k) {
      const Face& g = B.F[horizon_f[k]];
      const int a = g.v[horizon_e[k]], b = g.v[(horizon_e[k] + 1) % 3];
      if (seen_start[a] == vstamp || seen_end[b] == vstamp) return fail_(7);
      seen_start[a] = vstamp; seen_end[b] = vstamp;
    }
    for (size_t k = 0; k < m; ++k) {      // every end is some edge's start
      const Face& g = B.F[horizon_f[k]];
      if (seen_start[g.v[(horizon_e[k] + 1) % 3]] != vstamp) return fail_(8);
    }
    // one loop, not several: walk it
    std::vector<int>& nxt = stack;      // (free here) start vertex -> index of the edge starting there, via a small map
    nxt.assign(m, -1);
    // m is small (tens): quadratic walk
    size_t steps = 0; int cur = 0;
    do {
      const Face& g = B.F[horizon_f[cur]];
      const int b = g.v[(horizon_e[cur] + 1) % 3];
      int found = -1;
      for (size_t k = 0; k < m; ++k) if (B.F[horizon_f[k]].v[horizon_e[k]] == b) { found = (int)k; break; }
      if (found < 0) return fail_(9);
      cur = found;
    } while (++steps < m && cur != 0);
    return cur == 0 && steps == m;
  };
  auto unlink_point = [&](int f, int p) {      // remove p from the conflict list of f, refresh the farthest point
    Face& fc = B.F[f];
    int prev = -1, q = fc.head;
    while (q >= 0 && q != p) { prev = q; q = B.next[q]; }
    if (q == p) { if (prev < 0) fc.head = B.next[p]; else B.next[prev] = B.next[p]; }
    fc.far_pt = -1; fc.far_d = 0.0;
    for (q = fc.head; q >= 0; q = B.next[q]) { const double d = B.dist(fc, q); if (fc.far_pt < 0 || d > fc.far_d) { fc.far_pt = q; fc.far_d = d; } }
  };
  for (int round = 0; round < 16; ++round) {
  if (round > 0) {
    // points whose horizon could not be made simple earlier get another chance on the hull as it is now: each goes to the
    // conflict list of the face it is farthest in front of
    std::vector<int> retry;
    retry.swap(dropped);
    bool any = false;
    for (int q : retry) {
      int bf = -1; double bd = B.eps;
      for (size_t fi = 0; fi < B.F.size(); ++fi)
        if (B.F[fi].alive) { const double d = B.dist(B.F[fi], q); if (d > bd) { bd = d; bf = (int)fi; } }
      if (bf >= 0) { B.push(bf, q, bd); pending.push_back(bf); any = true; }
    }
    if (!any) break;
  }
  while (!pending.empty()) {
    int f = pending.back(); pending.pop_back();
    if (!B.F[f].alive || B.F[f].head < 0) continue;
    const int p = B.F[f].far_pt;       // farthest point of this face (tracked while the list was built)
    if (p < 0) return fail_(10);
    // visible set by flood fill; if the tolerance pinches the horizon, widen the visible set once (faces that p sees
    // edge-on count as visible too); if that does not help either, leave p out (checked against the finished hull below)
    bool ok_h = flood(f, p, B.eps) && horizon_is_simple_loop();
    if (!ok_h) ok_h = flood(f, p, -100.0 * B.eps) && horizon_is_simple_loop();
    if (!ok_h) {
      unlink_point(f, p);
      dropped.push_back(p);
      if (B.F[f].head >= 0) pending.push_back(f);
      continue;
    }
    // new faces, one per horizon edge (a -> b as oriented in the visible face)
    const size_t first_new = B.F.size();
    starts.clear(); ends.clear();
    for (size_t k = 0; k < horizon_f.size(); ++k) {
      int g = horizon_f[k], ei = horizon_e[k];
      int a = B.F[g].v[ei], b = B.F[g].v[(ei + 1) % 3];
      int h = B.F[g].nb[ei];
      int nf = B.add_face(a, b, p);
      starts.push_back(a); ends.push_back(b);
      B.F[nf].nb[0] = h;                       // across (a,b): the non-visible face h; fix h's back pointer
      for (int q = 0; q < 3; ++q)
        if (B.F[h].nb[q] == g && B.F[h].v[q] == b && B.F[h].v[(q + 1) % 3] == a) B.F[h].nb[q] = nf;
    }
    const size_t nn = B.F.size() - first_new;
    for (size_t k = 0; k < nn; ++k) {
      // edge 1: b -> p neighbours the new face whose start == b ; edge 2: p -> a the one whose end == a
      int b = ends[k], a = starts[k];
      for (size_t m = 0; m < nn; ++m) {
        if (starts[m] == b) B.F[first_new + k].nb[1] = (int)(first_new + m);
        if (ends[m] == a) B.F[first_new + k].nb[2] = (int)(first_new + m);
      }
      if (B.F[first_new + k].nb[1] < 0 || B.F[first_new + k].nb[2] < 0) return fail_(11);      // (cannot happen on a simple loop)
    }
    // redistribute the conflict points of the visible faces
    for (int g : visible) {
      int q = B.F[g].head;
      while (q >= 0) {
        int nx = B.next[q];
        if (q != p)
          for (size_t k = 0; k < nn; ++k) {
            double d = B.dist(B.F[first_new + k], q);
            if (d > B.eps) { B.push((int)(first_new + k), q, d); break; }
          }
        q = nx;
      }
      B.F[g].head = -1;
      B.F[g].alive = false;
    }
    for (size_t k = 0; k < nn; ++k) if (B.F[first_new + k].head >= 0) pending.push_back((int)(first_new + k));
  }
  if (dropped.empty()) break;
  }
  // points left out because their horizon could not be made simple must be (almost) inside the finished hull
  if (!dropped.empty()) {
    const double allow = std::max(100.0 * B.eps, 1e-6 * diag);      // 3e-4 mm on a humerus: the noise of float32 midpoints is 3e-5
    for (int q : dropped)
      for (size_t fi = 0; fi < B.F.size(); ++fi)
        if (B.F[fi].alive && B.dist(B.F[fi], q) > allow) return fail_(12);
  }
  // local convexity of the finished surface (closed + consistently oriented + locally convex = convex): the vertex of every
  // neighbour that is not on the shared edge lies on or behind the face.  The tolerance retries above can otherwise leave a
  // folded surface on clouds with thousands of nearly coplanar points.
  {
    const double allow = std::max(100.0 * B.eps, 1e-6 * diag);
    for (size_t fi = 0; fi < B.F.size(); ++fi) {
      if (!B.F[fi].alive) continue;
      const bool sliver = B.F[fi].area2 < 1e-7 * diag * diag;      // no usable normal: only its adjacency is checked
      for (int k = 0; k < 3; ++k) {
        const int g = B.F[fi].nb[k];
        if (g < 0 || g >= (int)B.F.size() || !B.F[g].alive) return fail_(14);
        if (sliver) continue;
        for (int q = 0; q < 3; ++q) {
          const int w = B.F[g].v[q];
          if (w != B.F[fi].v[0] && w != B.F[fi].v[1] && w != B.F[fi].v[2] && B.dist(B.F[fi], w) > allow) return fail_(15);
        }
      }
    }
  }
  // every face, slivers included, must face away from an interior point (the centroid of the initial simplex): the raw
  // cross product keeps its sign where the normalised normal of a sliver is noise.  A fold shows up as inverted faces.
  {
    double ctr[3];
    for (int k = 0; k < 3; ++k) ctr[k] = 0.25 * (P[3 * i0 + k] + P[3 * i1 + k] + P[3 * i2 + k] + P[3 * i3 + k]);
    for (size_t fi = 0; fi < B.F.size(); ++fi) {
      if (!B.F[fi].alive) continue;
      const double* a = P + 3 * B.F[fi].v[0]; const double* b = P + 3 * B.F[fi].v[1]; const double* c2 = P + 3 * B.F[fi].v[2];
      const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w[3] = {c2[0] - a[0], c2[1] - a[1], c2[2] - a[2]};
      const double nx = u[1] * w[2] - u[2] * w[1], ny = u[2] * w[0] - u[0] * w[2], nz = u[0] * w[1] - u[1] * w[0];
      if (nx * (ctr[0] - a[0]) + ny * (ctr[1] - a[1]) + nz * (ctr[2] - a[2]) > 0.0) return fail_(16);
    }
  }
  // compact
  std::vector<int> vmap(n, -1), fmap(B.F.size(), -1);
  H.vert_ids.clear(); H.tris.clear(); H.normals.clear(); H.edges.clear();
  int nf = 0;
  for (size_t f = 0; f < B.F.size(); ++f) {
    if (!B.F[f].alive) continue;
    fmap[f] = nf++;
    for (int k = 0; k < 3; ++k) {
      int v = B.F[f].v[k];
      if (vmap[v] < 0) { vmap[v] = (int)H.vert_ids.size(); H.vert_ids.push_back(v); }
      H.tris.push_back(vmap[v]);
    }
    // the record's normal: from the triangle rotated to its smallest point index first, coordinates centred on the
    // bounding-box midpoint -- independent of the insertion history and of the summation order of a mean, so that the device
    // hull (k_hull.h), which finds the same triangles by another route, hands k_obb_candidates the same bits
    {
      int r = 0;
      for (int k = 1; k < 3; ++k) if (B.F[f].v[k] < B.F[f].v[r]) r = k;
      double q[3][3];
      for (int k = 0; k < 3; ++k)
        for (int a = 0; a < 3; ++a) q[k][a] = pts_in[3 * (size_t)B.F[f].v[(r + k) % 3] + a] - 0.5 * (lo[a] + hi[a]);
      const double u[3] = {q[1][0] - q[0][0], q[1][1] - q[0][1], q[1][2] - q[0][2]}, w[3] = {q[2][0] - q[0][0], q[2][1] - q[0][1], q[2][2] - q[0][2]};
      const double nx = u[1] * w[2] - u[2] * w[1], ny = u[2] * w[0] - u[0] * w[2], nz = u[0] * w[1] - u[1] * w[0];
      const double l = std::sqrt(nx * nx + ny * ny + nz * nz);
      if (l == 0.0) { H.normals.push_back(0.0); H.normals.push_back(0.0); H.normals.push_back(1.0); }
      else { H.normals.push_back(nx / l); H.normals.push_back(ny / l); H.normals.push_back(nz / l); }
    }
  }
  for (size_t f = 0; f < B.F.size(); ++f) {
    if (!B.F[f].alive) continue;
    for (int k = 0; k < 3; ++k) {
      int g = B.F[f].nb[k];
      if (g < 0 || !B.F[g].alive) return fail_(13);       // broken adjacency
      if ((int)f < g) { H.edges.push_back(vmap[B.F[f].v[k]]); H.edges.push_back(vmap[B.F[f].v[(k + 1) % 3]]); H.edges.push_back(fmap[f]); H.edges.push_back(fmap[g]); }
    }
  }
  return true;
}

// Points closer than eps = 1e-10 * bbox diagonal to the current hull count as inside.  Clouds with thousands of nearly
// coplanar points (a subdivided or remeshed surface: float32 midpoints sit ~1e-5 mm off their parent triangle) can defeat a
// tolerance-based quickhull: a point is judged visible from one face and not from its neighbour and the horizon is no simple
// loop.  convex_hull_eps() validates every horizon, retries with a widened visible set, re-inserts the points it had to leave
// out, and checks the finished surface (left-out points inside, local convexity); when that still fails the cloud is
// joggled -- every coordinate moved by a deterministic pseudo-random amount of at most 1e-7 of the diagonal (3e-5 mm on a
// humerus, the size of the float32 noise that caused the trouble), which puts the points in general position -- and the
// hull of the joggled cloud is reported on the ORIGINAL coordinates.  `how`: 0 plain, k > 0 = joggle attempt k.
inline double joggle_unit(unsigned long long i, unsigned long long seed) {      // splitmix64 -> (-1, 1)
  unsigned long long z = (i + 0x9E3779B97F4A7C15ull * (seed + 1));
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  z ^= z >> 31;
  return (double)(z >> 11) * (2.0 / 9007199254740992.0) - 1.0;
}

inline bool convex_hull(const double* pts_in, int n, Hull& H, int* how = nullptr) {
  if (convex_hull_eps(pts_in, n, H, 1e-10)) { if (how) *how = 0; return true; }
  if (n < 4) return false;
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) { lo[k] = std::min(lo[k], pts_in[3 * i + k]); hi[k] = std::max(hi[k], pts_in[3 * i + k]); }
  const double diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
  std::vector<double> J(3 * (size_t)n);
  for (int attempt = 1; attempt <= 3; ++attempt) {
    const double amp = 1e-7 * diag * attempt;
    for (size_t i = 0; i < 3 * (size_t)n; ++i) J[i] = pts_in[i] + amp * joggle_unit(i, (unsigned long long)attempt);
    if (convex_hull_eps(J.data(), n, H, 1e-10)) {
      // normals from the original coordinates of the triangles (the record the device receives holds original points)
      const size_t nf = H.tris.size() / 3;
      for (size_t f = 0; f < nf; ++f) {
        const double* a = pts_in + 3 * (size_t)H.vert_ids[H.tris[3 * f]]; const double* b = pts_in + 3 * (size_t)H.vert_ids[H.tris[3 * f + 1]];
        const double* c = pts_in + 3 * (size_t)H.vert_ids[H.tris[3 * f + 2]];
        const double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w[3] = {c[0] - a[0], c[1] - a[1], c[2] - a[2]};
        double nx = u[1] * w[2] - u[2] * w[1], ny = u[2] * w[0] - u[0] * w[2], nz = u[0] * w[1] - u[1] * w[0];
        const double l = std::sqrt(nx * nx + ny * ny + nz * nz);
        if (l > 0.0) { H.normals[3 * f] = nx / l; H.normals[3 * f + 1] = ny / l; H.normals[3 * f + 2] = nz / l; }
      }
      if (how) *how = attempt;
      return true;
    }
  }
  return false;
}

}  // namespace shhull
