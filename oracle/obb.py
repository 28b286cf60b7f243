"""K2/K3/K4: oriented bounding box frame + head-end detection (oracle; test infrastructure).

Restates reference `src/shoulder/humerus/mesh.py:63-125` (`FullObb._obb`).  The
arithmetic lives in third-party packages absent from this image -- parity UNPINNED:
  * `Trimesh.apply_obb()` (mesh.py:82) = trimesh 3.23.5 `bounds.oriented_bounds`
    (convex hull -> one candidate per hull face normal -> 2-D minimum-area
    rectangle of the projected hull by rotating calipers over 2-D hull edges ->
    minimum volume -> box centred at the origin -> axes ordered by ascending
    extent, x shortest, z longest);
  * `circle_fit.least_squares_circle` 0.1.3 (mesh.py:102): scipy `leastsq` on
    R_i - mean(R) from the barycentre, residu = sum (R_i - mean R)^2.
Canonical rules: every hull face normal is a candidate (trimesh
bins directions at 0.1 rad and keeps one representative per bin, which depends on
qhull's face order); axis signs are fixed physically: first merged vertex has
x >= 0 and z >= 0 (before the head-end flip), y = z x x.
"""
import numpy as np
import scipy.optimize
import scipy.spatial

from .section import ZSlicer
from .xform import transform_pts


def _basis(n):
    k = int(np.argmin(np.abs(n)))
    e = np.zeros(3)
    e[k] = 1.0
    u = np.cross(n, e)
    u /= np.linalg.norm(u)
    return u, np.cross(n, u)


def min_area_rect_2d(p2: np.ndarray):
    """Minimum-area enclosing rectangle over 2-D hull edge directions.
    -> (area, unit edge direction (2,), extent along it, extent across it)."""
    h = scipy.spatial.ConvexHull(p2)
    hp = p2[h.vertices]
    e = np.roll(hp, -1, axis=0) - hp
    ln = np.linalg.norm(e, axis=1)
    e = e[ln > 0] / ln[ln > 0][:, None]
    a = hp @ e.T
    b = hp @ np.c_[-e[:, 1], e[:, 0]].T
    ea = a.max(axis=0) - a.min(axis=0)
    eb = b.max(axis=0) - b.min(axis=0)
    area = ea * eb
    i = int(np.argmin(area))
    return float(area[i]), e[i], float(ea[i]), float(eb[i])


def hull(verts: np.ndarray):
    """-> (hull vertex ids, hull triangles (ids into verts), unit normals)."""
    h = scipy.spatial.ConvexHull(verts, qhull_options="QbB Pp Qt")
    tri = h.simplices
    a, b, c = verts[tri[:, 0]], verts[tri[:, 1]], verts[tri[:, 2]]
    n = np.cross(b - a, c - a)
    ln = np.linalg.norm(n, axis=1)
    ok = ln > 1e-12 * ln.max()
    return h.vertices, tri[ok], n[ok] / ln[ok][:, None]


def _frame_of(verts, a0, a1, a2, ext, vol):
    """the chosen box -> (T 4x4 CT->OBB, extents ascending, volume): axes by ascending extent, signs fixed by vertex 0, centred"""
    order = np.argsort(ext, kind="stable")
    A = np.stack([a0, a1, a2])[order]
    R = A.copy()
    p = verts @ R.T
    c = 0.5 * (p.min(axis=0) + p.max(axis=0))
    if p[0, 0] - c[0] < 0:
        R[0] = -R[0]
    if p[0, 2] - c[2] < 0:
        R[2] = -R[2]
    R[1] = np.cross(R[2], R[0])
    p = verts @ R.T
    c = 0.5 * (p.min(axis=0) + p.max(axis=0))
    T = np.identity(4)
    T[:3, :3] = R
    T[:3, 3] = -c
    return T, ext[order], vol


def oriented_bounds(verts: np.ndarray):
    """-> (T 4x4 CT->OBB, extents (3,) ascending, min volume)."""
    verts = np.asarray(verts, dtype=np.float64)
    hv_ids, _, normals = hull(verts)
    hv = verts[hv_ids]
    best = (np.inf, None)
    for n in normals:
        u, v = _basis(n)
        h = hv @ n
        height = h.max() - h.min()
        area, e2, ea, eb = min_area_rect_2d(np.c_[hv @ u, hv @ v])
        vol = area * height
        if vol < best[0]:
            e = e2[0] * u + e2[1] * v
            best = (vol, (n, e, np.cross(n, e), np.array([height, ea, eb])))
    vol, (a0, a1, a2, ext) = best
    return _frame_of(verts, a0, a1, a2, ext, vol)


def oriented_bounds_large(verts: np.ndarray, chunk: int = 128):
    """oriented_bounds for hulls with tens of thousands of faces, where one qhull call per face normal (the loop above) takes a
    quarter of an hour: the 2-D hull of the projection along a face normal is read off the 3-D hull instead -- its edges are the
    hull edges whose two faces see the direction from opposite sides -- and the rectangle of every such edge is taken over the
    silhouette's vertices, a chunk of directions per matrix product.  Same candidate set, same minimum; pinned against
    oriented_bounds on the fixtures (tests/test_oracle_obb_large.py).  Test infrastructure like everything in oracle/."""
    verts = np.asarray(verts, dtype=np.float64)
    hv_ids, tri, normals = hull(verts)
    hv = verts[hv_ids]
    # undirected hull edges with their two faces
    e = np.concatenate([tri[:, [0, 1]], tri[:, [1, 2]], tri[:, [2, 0]]])
    fid = np.tile(np.arange(len(tri)), 3)
    key = np.sort(e, axis=1)
    order = np.lexsort((key[:, 1], key[:, 0]))
    key, fid = key[order], fid[order]
    assert len(key) % 2 == 0 and (key[0::2] == key[1::2]).all(), "hull is not a closed 2-manifold"
    ev, ef, eg = key[0::2], fid[0::2], fid[1::2]
    # (qhull's simplices are not wound consistently: the side test needs outward normals, the candidates keep hull()'s)
    out = normals * np.where(np.einsum("ij,ij->i", normals, verts[tri[:, 0]] - hv.mean(axis=0)) < 0, -1.0, 1.0)[:, None]
    best = (np.inf, None)
    for c0 in range(0, len(normals), chunk):
        N = normals[c0:c0 + chunk]
        H = hv @ N.T
        height = H.max(axis=0) - H.min(axis=0)
        front = (out @ N.T) > 0.0
        sil = front[ef] != front[eg]
        for j in range(len(N)):
            n = N[j]
            u, v = _basis(n)
            se = ev[sil[:, j]]
            ids = np.unique(se)
            p2 = np.c_[verts[ids] @ u, verts[ids] @ v]
            d = np.c_[verts[se[:, 1]] @ u, verts[se[:, 1]] @ v] - np.c_[verts[se[:, 0]] @ u, verts[se[:, 0]] @ v]
            ln = np.linalg.norm(d, axis=1)
            d = d[ln > 0] / ln[ln > 0][:, None]
            a = p2 @ d.T
            b = p2 @ np.c_[-d[:, 1], d[:, 0]].T
            ea = a.max(axis=0) - a.min(axis=0)
            eb = b.max(axis=0) - b.min(axis=0)
            area = ea * eb
            i = int(np.argmin(area))
            vol = float(area[i]) * float(height[j])
            if vol < best[0]:
                e3 = d[i, 0] * u + d[i, 1] * v
                best = (vol, (n, e3, np.cross(n, e3), np.array([float(height[j]), float(ea[i]), float(eb[i])])))
    vol, (a0, a1, a2, ext) = best
    return _frame_of(verts, a0, a1, a2, ext, vol)


def least_squares_circle(xy: np.ndarray):
    """circle_fit 0.1.3 `least_squares_circle` -> (xc, yc, R, residu)."""
    x, y = xy[:, 0], xy[:, 1]

    def f(c):
        ri = np.sqrt((x - c[0]) ** 2 + (y - c[1]) ** 2)
        return ri - ri.mean()

    center, _ = scipy.optimize.leastsq(f, (x.mean(), y.mean()))
    ri = np.sqrt((x - center[0]) ** 2 + (y - center[1]) ** 2)
    r = ri.mean()
    return center[0], center[1], r, float(np.sum((ri - r) ** 2))


FLIP = np.array([[-1.0, 0, 0, 0], [0, 1, 0, 0], [0, 0, -1, 0], [0, 0, 0, 1]])


def head_end_flip(z_bounds, residus):
    """mesh.py:88-124: the end whose section fits a circle better (strictly smaller residual; the -z end is looked at
    first, so it keeps a tie) is the head; flip iff it lies at negative z.  Pinned by tests/golden/prox_golden.npz (the
    reference's own FullObb._obb run with given residuals, tests/golden/make_prox_golden.py)."""
    humeral_end, residu_init = 0.0, np.inf
    for z_limit, residu in zip(z_bounds, residus):
        if residu < residu_init:
            residu_init, humeral_end = residu, z_limit
    return bool(humeral_end < 0)                                 # mesh.py:112


def full_obb(verts: np.ndarray, faces: np.ndarray, bounds=None):
    """mesh.py:63-125 -> dict(transform, z_bounds, z_length, verts_obb, flipped, residus).  `bounds`: oriented_bounds (default) or
    oriented_bounds_large for a hull of tens of thousands of faces."""
    T_obb, ext, vol = (bounds or oriented_bounds)(verts)
    v = transform_pts(verts, T_obb)
    z_bounds = (float(v[:, 2].min()), float(v[:, 2].max()))      # mesh.py:85
    z_length = abs(z_bounds[0]) + abs(z_bounds[1])               # mesh.py:86
    sl = ZSlicer(v, faces)
    residus = [least_squares_circle(sl.points(0.95 * z_limit))[3] for z_limit in z_bounds]      # mesh.py:91-102
    flipped = head_end_flip(z_bounds, residus)
    flip = FLIP if flipped else np.identity(4)
    if flipped:
        v = transform_pts(v, flip)
    return dict(transform=np.matmul(flip, T_obb), z_bounds=z_bounds, z_length=z_length,
                verts_obb=v, flipped=bool(flipped), residus=residus, extents=ext, volume=vol)
