"""Plane cut of a triangle mesh (oracle; test infrastructure only).

Restates `trimesh.Trimesh.slice_plane(plane_origin, plane_normal)` with cap=False as the reference calls it at
`src/shoulder/arthroplasty.py:80-87` (HumeralHeadOsteotomy.resect_mesh).  trimesh is a third-party dependency absent
from this image (pyproject pins `trimesh`), so this follows the published algorithm of
`trimesh.intersections.slice_faces_plane` (sign of every vertex with tolerance tol.merge = 1e-8; faces kept whole, cut
to a quad = two triangles, or cut to one triangle; crossing point = o + (num / denom) * d per edge) followed by what the
`Trimesh(vertices, faces)` constructor does with process=True: vertices whose coordinates agree after
`round(v * 1e8)` are merged and unreferenced ones dropped.  PARITY UNPINNED against trimesh itself (not importable
here); pinned by size-independent properties in tests/test_oracle_clip.py (area / volume additivity, both halves
re-assemble the surface, every cut edge lies in the plane, the section is closed).

Canonical ordering rule (trimesh's own vertex order comes from sorting row hashes, an implementation detail):
faces in trimesh's order (kept faces, first triangles of the quads, second triangles of the quads, triangles);
merged vertices numbered by the smallest pre-merge index that is referenced, where pre-merge indices are
[original vertices | 2 points per quad face | 2 points per triangle face].
"""
import numpy as np

TOL_MERGE = 1e-8


def _dot3(a, n):
    return (a[..., 0] * n[0] + a[..., 1] * n[1]) + a[..., 2] * n[2]


def slice_faces_plane(vertices, faces, origin, normal):
    """-> (pre-merge vertices, faces, cut edges) in the pre-merge numbering."""
    v = np.asarray(vertices, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)
    o = np.asarray(origin, dtype=np.float64)
    n = np.asarray(normal, dtype=np.float64)
    dots = _dot3(v - o, n)
    signs = np.zeros(len(v), dtype=np.int8)
    signs[dots < -TOL_MERGE] = 1
    signs[dots > TOL_MERGE] = -1
    sf = signs[f]
    ssum = sf.sum(axis=1, dtype=np.int64)
    asum = np.abs(sf).sum(axis=1, dtype=np.int64)
    onedge = (asum >= 2) & (np.abs(ssum) <= 1)
    inside = ssum == -asum
    on_plane = asum == 0
    if on_plane.any():
        tri = v[f[on_plane]]
        cr = np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0])
        nn = np.sqrt((cr[:, 0] ** 2 + cr[:, 1] ** 2) + cr[:, 2] ** 2)
        valid = nn > 1e-13
        unit = cr / np.where(valid, nn, 1.0)[:, None]
        inside[on_plane] = valid & (_dot3(unit, n) < 0.0)
    quad = onedge & (ssum < 0)
    tri_ = onedge & (ssum >= 0)
    new_faces = [f[inside]]
    nv = len(v)

    def crossing(F, j):
        a, b = v[F[np.arange(len(F)), j]], v[F[np.arange(len(F)), (j + 1) % 3]]
        d = b - a
        num = _dot3(o - a, n)
        den = _dot3(d, n)
        den = np.where(den == 0.0, 1e-12, den)
        return (num / den)[:, None] * d + a

    fq, sq = f[quad], sf[quad]
    nq = len(fq)
    qi = np.argmax(sq == 1, axis=1) if nq else np.zeros(0, dtype=np.int64)
    r = np.arange(nq)
    p0, p1 = crossing(fq, (qi + 2) % 3), crossing(fq, qi)
    quad_pts = np.stack([p0, p1], axis=1).reshape(-1, 3)
    n0 = nv + 2 * r
    a_, b_ = fq[r, (qi + 1) % 3], fq[r, (qi + 2) % 3]
    new_faces += [np.stack([a_, b_, n0], axis=1), np.stack([n0, n0 + 1, a_], axis=1)]
    edges = [np.stack([n0, n0 + 1], axis=1)]

    ft, st = f[tri_], sf[tri_]
    nt = len(ft)
    ti = np.argmax(st == -1, axis=1) if nt else np.zeros(0, dtype=np.int64)
    r = np.arange(nt)
    q0, q1 = crossing(ft, ti), crossing(ft, (ti + 2) % 3)
    tri_pts = np.stack([q0, q1], axis=1).reshape(-1, 3)
    m0 = nv + 2 * nq + 2 * r
    new_faces.append(np.stack([ft[r, ti], m0, m0 + 1], axis=1))
    edges.append(np.stack([m0, m0 + 1], axis=1))
    return np.concatenate([v, quad_pts, tri_pts]), np.concatenate(new_faces).astype(np.int64), np.concatenate(edges).astype(np.int64)


def merge(pre_v, pre_f, pre_e):
    """The Trimesh constructor's merge_vertices (8 decimals, referenced vertices only) under the canonical numbering."""
    ref = np.zeros(len(pre_v), dtype=bool)
    ref[pre_f.reshape(-1)] = True
    keys = np.round(pre_v * 1e8).astype(np.int64)
    idx = np.nonzero(ref)[0]
    _, first, inv = np.unique(keys[idx], axis=0, return_index=True, return_inverse=True)
    rep = idx[first][inv.reshape(-1)]                 # smallest referenced index with the same key (np.unique keeps the first)
    reps = np.unique(rep)
    new_id = np.full(len(pre_v), -1, dtype=np.int64)
    new_id[idx] = np.searchsorted(reps, rep)
    return pre_v[reps], new_id[pre_f], new_id[pre_e]


def slice_plane(vertices, faces, origin, normal):
    """-> (vertices (n,3) f64, faces (m,3) i64, cut edges (k,2) i64)."""
    return merge(*slice_faces_plane(vertices, faces, origin, normal))


def loops_from_edges(edges):
    """Chain undirected cut edges into closed vertex loops (lists of vertex ids); open chains raise."""
    adj = {}
    for a, b in np.asarray(edges).tolist():
        if a == b:
            continue
        adj.setdefault(a, []).append(b)
        adj.setdefault(b, []).append(a)
    if any(len(v) != 2 for v in adj.values()):
        raise ValueError("section is not a set of simple closed loops")
    seen, loops = set(), []
    for s in sorted(adj):
        if s in seen:
            continue
        loop, prev, cur = [s], None, s
        seen.add(s)
        while True:
            nxt = [x for x in adj[cur] if x != prev]
            nx = nxt[0] if nxt else adj[cur][0]
            if nx == s:
                break
            loop.append(nx)
            seen.add(nx)
            prev, cur = cur, nx
        loops.append(loop)
    return loops
