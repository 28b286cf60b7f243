"""K3/K5: mesh / z-plane sections joined into closed loops (oracle; test infrastructure).

Restates what the reference obtains from trimesh 3.23.5 (absent from this image;
poetry.lock:4145) at
  `mesh.section(...)`            mesh.py:95-97, surgical_neck.py:37-39
  `mesh.section_multiplane(...)` slice.py:26-28
i.e. `trimesh.intersections.mesh_plane` followed by `load_path` (closed polylines,
`discrete` rings forced counter-clockwise).  Parity UNPINNED (third-party code);
canonical rules:
  * vertex/plane classification: d = z_v - z_plane, "below" iff d < -1e-8, else
    "above" (trimesh zeroes |d|<1e-8; treating on-plane vertices as above yields the
    same geometry to 1e-8 mm while keeping every cut a closed manifold curve);
  * a crossing point is computed once per mesh edge, from the lower to the higher
    vertex id: p = p_lo + d_lo/(d_lo-d_hi) * (p_hi-p_lo);
  * every loop is returned counter-clockwise (shoelace area > 0), closed
    (first == last) and starts at the crossing point on the mesh edge with the
    smallest (min vid, max vid) key (B-1).
"""
import numpy as np

TOL = 1e-8


class ZSlicer:
    """Sections of one mesh (OBB-frame vertices, fp64) by planes z = const."""

    def __init__(self, verts: np.ndarray, faces: np.ndarray):
        self.v = np.asarray(verts, dtype=np.float64)
        self.f = np.asarray(faces, dtype=np.int64)
        fz = self.v[:, 2][self.f]
        self.fzmin = fz.min(axis=1)
        self.fzmax = fz.max(axis=1)
        self.order = np.argsort(self.fzmin, kind="stable")
        self.fzmin_sorted = self.fzmin[self.order]
        self.nv = len(self.v)

    # -- segments -----------------------------------------------------------------
    def segments(self, z: float):
        """-> (start_key, end_key, start_pt, end_pt) for every crossing triangle.

        Segment direction: from the edge crossed downwards (+ -> -) to the edge
        crossed upwards (- -> +) in the triangle's cyclic order; for an outward
        oriented surface this walks the outer boundary counter-clockwise."""
        hi = np.searchsorted(self.fzmin_sorted, z + TOL, side="right")
        cand = self.order[:hi]
        cand = cand[self.fzmax[cand] >= z - TOL]
        f = self.f[cand]
        d = self.v[:, 2][f] - z
        s = np.where(d < -TOL, -1, 1)
        cross = s.min(axis=1) != s.max(axis=1)
        f, d, s = f[cross], d[cross], s[cross]
        n = len(f)
        if n == 0:
            z2 = np.zeros((0, 2))
            e = np.zeros(0, dtype=np.int64)
            return e, e, z2, z2
        s_to = np.roll(s, -1, axis=1)
        up = np.argmax((s == -1) & (s_to == 1), axis=1)
        dn = np.argmax((s == 1) & (s_to == -1), axis=1)
        r = np.arange(n)

        def edge(e):
            a, b = f[r, e], f[r, (e + 1) % 3]
            da, db = d[r, e], d[r, (e + 1) % 3]
            swap = a > b
            lo = np.where(swap, b, a)
            hi_ = np.where(swap, a, b)
            dlo = np.where(swap, db, da)
            dhi = np.where(swap, da, db)
            t = dlo / (dlo - dhi)
            plo = self.v[lo, :2]
            phi = self.v[hi_, :2]
            pt = plo + t[:, None] * (phi - plo)
            return lo * self.nv + hi_, pt

        sk, sp = edge(dn)
        ek, ep = edge(up)
        return sk, ek, sp, ep

    # -- loops --------------------------------------------------------------------
    def loops(self, z: float):
        """-> list of closed CCW rings, each (n+1,2) float64, canonical start."""
        sk, ek, sp, _ = self.segments(z)
        n = len(sk)
        if n == 0:
            return []
        idx_of_start = {int(k): i for i, k in enumerate(sk)}
        nxt = np.array([idx_of_start.get(int(k), -1) for k in ek], dtype=np.int64)
        seen = np.zeros(n, dtype=bool)
        rings = []
        for i0 in range(n):
            if seen[i0]:
                continue
            chain = []
            i = i0
            closed = False
            while i >= 0 and not seen[i]:
                seen[i] = True
                chain.append(i)
                i = nxt[i]
                if i == i0:
                    closed = True
                    break
            if not closed or len(chain) < 3:
                continue
            chain = np.array(chain)
            keys = sk[chain]
            k0 = int(np.argmin(keys))
            chain = np.r_[chain[k0:], chain[:k0]]
            pts = sp[chain]
            x, y = pts[:, 0], pts[:, 1]
            area2 = np.sum(x * np.roll(y, -1) - np.roll(x, -1) * y)
            if area2 < 0:
                pts = np.r_[pts[:1], pts[1:][::-1]]
            rings.append(np.r_[pts, pts[:1]])
        return rings

    def points(self, z: float):
        """Unique crossing points of one section (`Path.vertices`), (n,2)."""
        _, _, sp, _ = self.segments(z)
        return sp


def ring_area(ring: np.ndarray) -> float:
    """Shoelace area of a closed ring (first == last); positive for CCW."""
    x, y = ring[:-1, 0], ring[:-1, 1]
    xn, yn = ring[1:, 0], ring[1:, 1]
    return 0.5 * float(np.sum(x * yn - xn * y))
