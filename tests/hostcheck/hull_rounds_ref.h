// hull_rounds_ref.h -- sequential restatement (test infrastructure, NOT product) of the round-based device quickhull of
// shoulder_amd/csrc/k_hull.h: the same phases, the same deterministic rules, one statement per parallel step, so that the
// algorithm can be checked on the CPU (against qhull and the host quickhull) where a debugger exists.  The device kernel is
// a hand translation of this file; tests/test_host_scalar.py::test_round_hull_* pin both to the same hulls.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>
#include <unordered_map>
#include <cstdio>
#include <cstdlib>

namespace hullref {

struct Out {
  std::vector<int> vert_ids;      // hull vertices = input point indices, ascending
  std::vector<int> tris;          // 3 per face: input point indices, rotated so that the smallest comes first
  int rounds = 0, fail = 0, insertions = 0;
};

// pts: n x 3 float32.  Returns 0 or a positive failure code (the product then falls back to the host quickhull).
inline int hull_rounds(const float* pts, int n, Out& out, int K = 128, int VMAX = 512, int FCAP = 16384, double eps_rel = 1e-10) {
  out = Out();
  if (n < 4) return out.fail = 1;
  // centre = bounding-box midpoint (order independent, unlike a mean), eps = eps_rel * diagonal
  double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) { double v = pts[3 * i + k]; lo[k] = std::min(lo[k], v); hi[k] = std::max(hi[k], v); }
  double c[3];
  for (int k = 0; k < 3; ++k) c[k] = 0.5 * (lo[k] + hi[k]);
  const double diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
  const double eps = eps_rel * diag;
  std::vector<double> P(3 * (size_t)n);
  for (int i = 0; i < n; ++i)
    for (int k = 0; k < 3; ++k) P[3 * (size_t)i + k] = (double)pts[3 * i + k] - c[k];
  struct Face { int v[3]; double n[3], d; bool alive; unsigned long long apex; int kill; };
  std::vector<Face> F(FCAP);
  int nslots = 0;
  auto plane = [&](Face& f) {
    const double* a = &P[3 * (size_t)f.v[0]]; const double* b = &P[3 * (size_t)f.v[1]]; const double* cc = &P[3 * (size_t)f.v[2]];
    double u[3] = {b[0] - a[0], b[1] - a[1], b[2] - a[2]}, w[3] = {cc[0] - a[0], cc[1] - a[1], cc[2] - a[2]};
    double nx = u[1] * w[2] - u[2] * w[1], ny = u[2] * w[0] - u[0] * w[2], nz = u[0] * w[1] - u[1] * w[0];
    double l = std::sqrt(nx * nx + ny * ny + nz * nz);
    if (l == 0.0) { f.n[0] = f.n[1] = 0; f.n[2] = 1; f.d = a[2]; return; }
    f.n[0] = nx / l; f.n[1] = ny / l; f.n[2] = nz / l;
    f.d = f.n[0] * a[0] + f.n[1] * a[1] + f.n[2] * a[2];
  };
  auto dist = [&](const Face& f, int p) { return f.n[0] * P[3 * (size_t)p] + f.n[1] * P[3 * (size_t)p + 1] + f.n[2] * P[3 * (size_t)p + 2] - f.d; };
  auto key = [&](double d, int q) -> unsigned long long {      // (distance truncated to 43 bits, smaller index wins ties); d > 0
    unsigned long long b; std::memcpy(&b, &d, 8);
    return (b & ~0x1FFFFFull) | (unsigned long long)(0x1FFFFF - q);
  };
  // ---- initial simplex (first index wins every tie, as in sh_hull.h)
  int i0 = 0, i1 = 0;
  for (int i = 1; i < n; ++i) { if (P[3 * (size_t)i] < P[3 * (size_t)i0]) i0 = i; if (P[3 * (size_t)i] > P[3 * (size_t)i1]) i1 = i; }
  if (i0 == i1) return out.fail = 2;
  double e[3] = {P[3 * (size_t)i1] - P[3 * (size_t)i0], P[3 * (size_t)i1 + 1] - P[3 * (size_t)i0 + 1], P[3 * (size_t)i1 + 2] - P[3 * (size_t)i0 + 2]};
  int i2 = -1; double best = 0;
  for (int i = 0; i < n; ++i) {
    double w[3] = {P[3 * (size_t)i] - P[3 * (size_t)i0], P[3 * (size_t)i + 1] - P[3 * (size_t)i0 + 1], P[3 * (size_t)i + 2] - P[3 * (size_t)i0 + 2]};
    double cx = e[1] * w[2] - e[2] * w[1], cy = e[2] * w[0] - e[0] * w[2], cz = e[0] * w[1] - e[1] * w[0];
    double a2 = cx * cx + cy * cy + cz * cz;
    if (a2 > best) { best = a2; i2 = i; }
  }
  if (i2 < 0) return out.fail = 3;
  Face tmp; tmp.v[0] = i0; tmp.v[1] = i1; tmp.v[2] = i2; plane(tmp);
  int i3 = -1; best = 0;
  for (int i = 0; i < n; ++i) { double d = std::fabs(dist(tmp, i)); if (d > best) { best = d; i3 = i; } }
  if (i3 < 0 || best <= eps) return out.fail = 4;
  if (dist(tmp, i3) > 0) std::swap(i1, i2);
  const int init[4][3] = {{i0, i1, i2}, {i0, i3, i1}, {i1, i3, i2}, {i2, i3, i0}};
  for (int f = 0; f < 4; ++f) { Face& g = F[f]; g.v[0] = init[f][0]; g.v[1] = init[f][1]; g.v[2] = init[f][2]; plane(g); g.alive = true; g.apex = 0; g.kill = -1; }
  nslots = 4;
  std::vector<int> conf(n, -1);      // conflict face of a point; -1 inside, -2 hull vertex
  conf[i0] = conf[i1] = conf[i2] = conf[i3] = -2;
  for (int p = 0; p < n; ++p) {
    if (conf[p] == -2) continue;
    for (int f = 0; f < 4; ++f) { double d = dist(F[f], p); if (d > eps) { conf[p] = f; F[f].apex = std::max(F[f].apex, key(d, p)); break; } }
  }
  // ---- rounds
  auto hash32 = [](int x) { unsigned u = (unsigned)x * 2654435761u; u ^= u >> 15; u *= 2246822519u; u ^= u >> 13; return u; };
  std::unordered_map<long long, unsigned> eowner;      // undirected edge (min * n + max) -> smallest priority that claims it
  struct Cand { int face, pt, ok, nh, off; unsigned prio; std::vector<int> vis; std::vector<int> ha, hb; };
  std::vector<Cand> C;
  std::vector<int> freeStack;                           // dead slots of EARLIER rounds, popped from the back
  const int KC = K;                                     // candidates per round (device: 128)
  for (int round = 0; round < 100000; ++round) {
    // R0: candidates.  Every alive face that holds outside points offers its farthest one; priority = a hash of (the point's
    // index, the round) (spatially incoherent, so that chains of neighbours do not all lose to one another).  When there are more than
    // 64, those with priority <= 2^32 * 64 / ncand take part (about 64), plus the minimum; at most KC, in slot order.
    C.clear();
    int ncand = 0; unsigned minp = 0xFFFFFFFFu;
    for (int f = 0; f < nslots; ++f)
      if (F[f].alive && F[f].apex != 0) { ++ncand; minp = std::min(minp, hash32((0x1FFFFF - (int)(F[f].apex & 0x1FFFFF)) + round * 0x9E3779B)); }
    if (ncand == 0) break;
    out.rounds = round + 1;
    const unsigned T = ncand <= 64 ? 0xFFFFFFFFu : (unsigned)((64ull << 32) / (unsigned long long)ncand);
    for (int f = 0; f < nslots && (int)C.size() < KC; ++f) {
      if (!F[f].alive || F[f].apex == 0) continue;
      const int pt = 0x1FFFFF - (int)(F[f].apex & 0x1FFFFF);
      const unsigned h = hash32(pt + round * 0x9E3779B);      // (re-drawn every round: a face's apex stays the same until the face dies)
      if (h > T && h != minp) continue;
      Cand cd; cd.face = f; cd.pt = pt; cd.ok = 1; cd.nh = 0; cd.off = 0; cd.prio = (h & ~127u) | (unsigned)C.size();
      C.push_back(cd);
    }
    // R1: visible faces of every candidate (all alive slots, slot order); every undirected edge of a visible face is claimed
    // with the candidate's priority, the smallest claim wins
    eowner.clear();
    for (size_t ci = 0; ci < C.size(); ++ci) {
      Cand& cd = C[ci];
      for (int f = 0; f < nslots; ++f)
        if (F[f].alive && dist(F[f], cd.pt) > eps) cd.vis.push_back(f);
      if ((int)cd.vis.size() > VMAX) return out.fail = 20;
      if (cd.vis.empty()) return out.fail = 25;
      for (int f : cd.vis) for (int k = 0; k < 3; ++k) {
        const int a = F[f].v[k], b = F[f].v[(k + 1) % 3];
        const long long ek = (long long)std::min(a, b) * n + std::max(a, b);
        auto it = eowner.find(ek);
        if (it == eowner.end()) eowner[ek] = cd.prio; else it->second = std::min(it->second, cd.prio);
      }
    }
    // R2: a candidate goes ahead when every edge of its visible faces is its own: its visible region then shares no face with
    // another going candidate's and is not edge-adjacent to it (touching in a vertex is harmless: a cone's faces are only
    // reachable through its ring)
    for (size_t ci = 0; ci < C.size(); ++ci)
      for (int f : C[ci].vis) for (int k = 0; k < 3; ++k) {
        const int a = F[f].v[k], b = F[f].v[(k + 1) % 3];
        if (eowner[(long long)std::min(a, b) * n + std::max(a, b)] != C[ci].prio) C[ci].ok = 0;
      }
    // R3: horizon = directed edges of visible faces whose reverse is not an edge of a visible face; must be one simple loop
    for (size_t ci = 0; ci < C.size(); ++ci) {
      Cand& cd = C[ci];
      if (!cd.ok) continue;
      for (int f : cd.vis) for (int k = 0; k < 3; ++k) {
        const int a = F[f].v[k], b = F[f].v[(k + 1) % 3];
        bool twin = false;
        for (int g : cd.vis) for (int q = 0; q < 3; ++q) if (F[g].v[q] == b && F[g].v[(q + 1) % 3] == a) twin = true;
        if (!twin) { cd.ha.push_back(a); cd.hb.push_back(b); }
      }
      cd.nh = (int)cd.ha.size();
      if (cd.nh < 3) return out.fail = 21;
      int cur = 0, steps = 0;
      do {
        int found = -1, cnt = 0;
        for (int k = 0; k < cd.nh; ++k) if (cd.ha[k] == cd.hb[cur]) { if (found < 0) found = k; ++cnt; }
        if (cnt != 1) return out.fail = 22;
        cur = found;
      } while (++steps < cd.nh && cur != 0);
      if (cur != 0 || steps != cd.nh) return out.fail = 23;
    }
    if (getenv("HULLREF_TRACE")) { int nok = 0; for (auto& cd : C) nok += cd.ok; fprintf(stderr, "round %d ncand %d selected %zu ok %d nslots %d\n", round, ncand, C.size(), nok, nslots); }
    // R4: slots of the new faces, candidates in order: free slots of earlier rounds first (stack), then the end of the array
    std::vector<int> newslots;
    for (size_t ci = 0; ci < C.size(); ++ci) {
      Cand& cd = C[ci];
      if (!cd.ok) continue;
      cd.off = (int)newslots.size();
      for (int k = 0; k < cd.nh; ++k) {
        int sl;
        if (!freeStack.empty()) { sl = freeStack.back(); freeStack.pop_back(); }
        else { if (nslots >= FCAP) return out.fail = 24; sl = nslots++; }
        newslots.push_back(sl);
      }
    }
    // R5: kill the visible faces (their slots become free for LATER rounds), create the new ones
    for (size_t ci = 0; ci < C.size(); ++ci) {
      Cand& cd = C[ci];
      if (!cd.ok) continue;
      ++out.insertions;
      for (int f : cd.vis) { F[f].alive = false; F[f].kill = (int)ci; F[f].apex = 0; freeStack.push_back(f); }
    }
    for (size_t ci = 0; ci < C.size(); ++ci) {
      Cand& cd = C[ci];
      if (!cd.ok) continue;
      for (int k = 0; k < cd.nh; ++k) {
        Face& g = F[newslots[cd.off + k]];
        g.v[0] = cd.ha[k]; g.v[1] = cd.hb[k]; g.v[2] = cd.pt; plane(g); g.alive = true; g.apex = 0; g.kill = -1;
      }
    }
    // R6: points of the killed faces go to the first new face (creation order) of their killer that sees them
    for (int q = 0; q < n; ++q) {
      const int f = conf[q];
      if (f < 0 || F[f].kill < 0) continue;
      Cand& cd = C[F[f].kill];
      if (q == cd.pt) { conf[q] = -2; continue; }
      conf[q] = -1;
      for (int k = 0; k < cd.nh; ++k) {
        const int sl = newslots[cd.off + k];
        const double d = dist(F[sl], q);
        if (d > eps) { conf[q] = sl; F[sl].apex = std::max(F[sl].apex, key(d, q)); break; }
      }
    }
    for (size_t ci = 0; ci < C.size(); ++ci) if (C[ci].ok) for (int f : C[ci].vis) F[f].kill = -1;
  }
  // ---- emit: vertices ascending, faces in slot order with the smallest vertex first; closedness check
  std::vector<char> used(n, 0);
  std::vector<long long> dir;
  int nf = 0;
  for (int f = 0; f < nslots; ++f) {
    if (!F[f].alive) continue;
    ++nf;
    int r = 0;
    for (int k = 1; k < 3; ++k) if (F[f].v[k] < F[f].v[r]) r = k;
    for (int k = 0; k < 3; ++k) { const int v = F[f].v[(r + k) % 3]; out.tris.push_back(v); used[v] = 1; }
    for (int k = 0; k < 3; ++k) dir.push_back((long long)F[f].v[k] * n + F[f].v[(k + 1) % 3]);
  }
  std::sort(dir.begin(), dir.end());
  for (size_t i = 0; i < dir.size(); ++i) {
    if (i + 1 < dir.size() && dir[i] == dir[i + 1]) return out.fail = 30;
    const long long a = dir[i] / n, b = dir[i] % n;
    if (!std::binary_search(dir.begin(), dir.end(), b * n + a)) return out.fail = 31;
  }
  for (int v = 0; v < n; ++v) if (used[v]) out.vert_ids.push_back(v);
  if ((int)out.vert_ids.size() - (int)dir.size() / 2 + nf != 2) return out.fail = 32;
  return 0;
}

}  // namespace hullref
