"""K24/K25 + a19: trans-epicondylar axis (oracle; test infrastructure).

Restates reference `src/shoulder/humerus/epicondyle.py:29-101` and the rectangle
helpers `utils.py:36-97`.  shapely 2.1.0 is absent from this image (parity
UNPINNED); its published behaviour is restated:
  * `minimum_rotated_rectangle`: minimum-AREA rectangle over convex-hull edges;
  * rotate(az) -> scale(1.5, 0.999) -> rotate(-az) about the rectangle centre with
    az = azimuth of the long side (utils.py:36-47) leaves a rectangle with the same
    centre and orientation, 1.5x wider and 0.999x as long, so
    `polygon.difference(rect)` = the parts of the polygon farther than 0.4995*L from
    the centre along the long axis, one or more pieces per end;
  * `centroid` = area centroid of each piece.
Canonical rules: B-2 (the polygon of a slice is its largest loop), B-7 (the two end
points are always the farthest-apart pair of piece centroids).
"""
import numpy as np

from .slices import cutoff_range
from .xform import construct_csys, inv_transform, transform_pts


def convex_hull_2d(p):
    """Andrew monotone chain -> hull vertices CCW, no duplicate end point."""
    p = np.asarray(p, dtype=np.float64)
    idx = np.lexsort((p[:, 1], p[:, 0]))
    q = p[idx]

    def half(seq):
        h = []
        for pt in seq:
            while len(h) >= 2 and ((h[-1][0] - h[-2][0]) * (pt[1] - h[-2][1])
                                   - (h[-1][1] - h[-2][1]) * (pt[0] - h[-2][0])) <= 0:
                h.pop()
            h.append(pt)
        return h

    lo, up = half(q), half(q[::-1])
    return np.array(lo[:-1] + up[:-1])


def min_area_rect(p):
    """-> dict(center (2,), major (2,) unit vector along the long side, L, W, area)."""
    h = convex_hull_2d(p)
    best = None
    for i in range(len(h)):
        e = h[(i + 1) % len(h)] - h[i]
        ln = np.hypot(e[0], e[1])
        if ln == 0:
            continue
        e = e / ln
        n = np.array([-e[1], e[0]])
        a, b = h @ e, h @ n
        ea, eb = a.max() - a.min(), b.max() - b.min()
        area = ea * eb
        if best is None or area < best["area"]:
            ctr = e * (0.5 * (a.max() + a.min())) + n * (0.5 * (b.max() + b.min()))
            if ea >= eb:
                best = dict(center=ctr, major=e, L=ea, W=eb, area=area)
            else:
                best = dict(center=ctr, major=n, L=eb, W=ea, area=area)
    return best


def _poly_centroid(poly):
    """Area centroid of an open vertex list, evaluated about its first vertex."""
    q = poly - poly[0]
    x, y = q[:, 0], q[:, 1]
    xn, yn = np.roll(x, -1), np.roll(y, -1)
    cr = x * yn - xn * y
    a = cr.sum() / 2
    cx = ((x + xn) * cr).sum() / (6 * a)
    cy = ((y + yn) * cr).sum() / (6 * a)
    return poly[0] + np.array([cx, cy]), a


def clip_halfplane_pieces(ring, c, m, w0):
    """Connected pieces of a simple closed CCW ring inside {(p-c).m > w0}.
    -> list of (centroid (2,), area)."""
    pts = ring[:-1]
    n = len(pts)
    f = (pts - c) @ m - w0
    inside = f > 0
    if inside.all():
        return [_poly_centroid(pts)]
    if not inside.any():
        return []
    perp = np.array([-m[1], m[0]])
    start = next(i for i in range(n) if not inside[i] and inside[(i + 1) % n])
    chains, cur, cross_s = [], None, []
    for k in range(n):
        i, j = (start + k) % n, (start + k + 1) % n
        if inside[i] != inside[j]:
            t = f[i] / (f[i] - f[j])
            x = pts[i] + t * (pts[j] - pts[i])
            s = (x - c) @ perp
            if inside[j]:          # entering
                cur = dict(pts=[x, pts[j]], s_in=s)
            else:                  # leaving
                cur["pts"].append(x)
                cur["s_out"] = s
                chains.append(cur)
                cur = None
        elif inside[j] and cur is not None:
            cur["pts"].append(pts[j])
    # pair crossings along the clip line: sorted positions (0,1),(2,3),...
    ev = []
    for k, ch in enumerate(chains):
        ev.append((ch["s_in"], k, "in"))
        ev.append((ch["s_out"], k, "out"))
    ev.sort(key=lambda e: e[0])
    partner = {}
    for a in range(0, len(ev), 2):
        e0, e1 = ev[a], ev[a + 1]
        partner[(e0[1], e0[2])] = (e1[1], e1[2])
        partner[(e1[1], e1[2])] = (e0[1], e0[2])
    used = [False] * len(chains)
    pieces = []
    for k0 in range(len(chains)):
        if used[k0]:
            continue
        poly, k = [], k0
        while not used[k]:
            used[k] = True
            poly.extend(chains[k]["pts"])
            k = partner[(k, "out")][0]
        pieces.append(_poly_centroid(np.array(poly)))
    return pieces


def first_max(dist):
    """epicondyle.py:39 `dist.index(max(dist))`: the first row with the longest rectangle."""
    dist = list(dist)
    return dist.index(max(dist))


def far_pair(cents):
    """epicondyle.py:57-81 (canonical B-7): the two end pieces.  With more than two pieces the reference takes the pair of
    centroids farthest apart (first maximum over itertools.combinations order); with exactly two, those two."""
    cents = np.asarray(cents, dtype=np.float64)
    d2 = ((cents[:, None, :] - cents[None, :, :]) ** 2).sum(axis=2)
    i, j = np.unravel_index(int(np.argmax(d2)), d2.shape)
    return min(i, j), max(i, j)


def medial_first(end_pts, T_obb, canal_axis, central_axis):
    """epicondyle.py:84-99: ends to CT, then row 0 = the end with the smaller x in construct_csys(canal, head-central).
    -> (end_pts (2,3) OBB, end_ct (2,3))."""
    end_ct = transform_pts(end_pts, inv_transform(T_obb))
    tfrm = construct_csys(canal_axis, central_axis)               # :90
    medial_idx = int(np.argmin(transform_pts(end_ct, tfrm)[:, 0]))
    if medial_idx == 1:
        end_pts, end_ct = end_pts[::-1], end_ct[::-1]
    return end_pts, end_ct


def te_axis(distal_largest_rings, distal_zs_all, T_obb, canal_axis_ct, central_axis_ct, cutoff=(0.8, 0.99)):
    """epicondyle.py:29-101 -> dict(axis_obb (2,3), axis_ct (2,3), idx_max, dists).  The glue (row choice, piece pair,
    medial-first order) is pinned by tests/golden/te_golden.npz: the reference's own epicondyle.py run with the shapely
    results injected (tests/golden/make_te_golden.py); the rectangles and the clipped pieces themselves are restated."""
    a, b = cutoff_range(len(distal_zs_all), cutoff)
    rings = distal_largest_rings[a:b]
    zs = distal_zs_all[a:b]
    rects = [min_area_rect(r[:-1]) for r in rings]
    dist = [r["L"] for r in rects]
    k = first_max(dist)
    ring, rect, z = rings[k], rects[k], zs[k]
    half = 0.5 * 0.999 * rect["L"]
    pieces = clip_halfplane_pieces(ring, rect["center"], rect["major"], half)
    pieces += clip_halfplane_pieces(ring, rect["center"], -rect["major"], half)
    if len(pieces) < 2:
        raise ValueError("trans-epicondylar axis: fewer than two end pieces")
    cents = np.array([p[0] for p in pieces])
    i, j = far_pair(cents)
    end_pts = np.c_[cents[[i, j]], np.repeat(z, 2)]
    end_pts, end_ct = medial_first(end_pts, T_obb, canal_axis_ct, central_axis_ct)
    return dict(axis_obb=end_pts, axis_ct=end_ct, idx_max=a + k, dists=np.array(dist))
