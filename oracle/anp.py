"""K18/K20 + a15-a18: anatomic neck (oracle; test infrastructure).

Restates reference `src/shoulder/humerus/anatomic_neck.py`:
  anp_image      :34-58   (even-theta re-interpolation, roll to groove angle, global min-max)
  mask_points    :79-118  (mask = logit > 0; |diff(mask, prepend=0)| along theta; (z,theta,r)->xyz)
  neck_plane     :123-153 (SVD plane, normal flipped to +z, centre := LSQ-ellipse centre)
  axis_normal    :174-200, axis_central :202-236 (canonical B-6: nearest hit per direction)
  plane_points   :155-172
The UNet itself (K19) is absent from the reference tree (.MISSING_LARGE_BLOBS); the
oracle takes the logits from `oracle.unet` (builder-defined architecture, see there).
"""
import numpy as np

from .fits import ellipse_center, plane_basis, plane_best_fit, ray_nearest
from .xform import inv_transform, transform_pts


def anp_image(itr, bg_theta):
    """itr = itr_start(cutoff) (R,2,M).  -> (image (R,M) in [0,1] float64, itr_shft (R,2,M), roll idx (R,))."""
    R, _, M = itr.shape
    image = np.zeros((R, M))
    itr_shft = np.zeros(itr.shape)
    roll = np.zeros(R, dtype=np.int64)
    for i, tr in enumerate(itr):
        t_sampling = np.linspace(tr[0][0], tr[0][-2], tr.shape[1])
        tr = np.c_[t_sampling, np.interp(t_sampling, tr[0, :-1], tr[1, :-1])].T
        k = int(np.argmin(np.abs(tr[0] - bg_theta)))
        tr = np.c_[tr[:, k:], tr[:, :k]]
        image[i] = tr[1]
        itr_shft[i] = tr
        roll[i] = k
    image = minmax_like_sklearn(image)           # MinMaxScaler on the flattened image (:56-58)
    return image, itr_shft, roll


def minmax_like_sklearn(image):
    """sklearn MinMaxScaler arithmetic: X * (1/(max-min)) + (0 - min*(1/(max-min)))."""
    lo, hi = image.min(), image.max()
    rng = hi - lo
    if rng == 0:
        rng = 1.0
    scale = 1.0 / rng
    return image * scale + (0.0 - lo * scale)


def mask_points(logits, itr_shft, zs):
    """-> dict(edge (R,M) bool, mask (R,M) bool, points_obb (K,3), articular_obb (A,3))."""
    mask = (np.squeeze(logits) > 0).astype(int)
    edge = np.abs(np.diff(mask, prepend=0)).astype(bool)
    mask = mask.astype(bool)
    t, r = itr_shft[:, 0, :], itr_shft[:, 1, :]
    zz = np.repeat(zs.reshape(-1, 1), t.shape[1], axis=1)

    def xyz(sel):
        return np.c_[r[sel] * np.cos(t[sel]), r[sel] * np.sin(t[sel]), zz[sel]]

    return dict(edge=edge, mask=mask, points_obb=xyz(edge), articular_obb=xyz(mask))


def neck_plane(points_obb):
    """-> (point (3,), normal (3,)) in the OBB frame."""
    c, n = plane_best_fit(points_obb)
    n = n.copy()
    if n[-1] < 0:
        n *= -1
    u, v, w = plane_basis(n)
    rel = points_obb - c
    ctr2 = ellipse_center(np.c_[rel @ u, rel @ v])
    return c + ctr2[0] * u + ctr2[1] * v, n


def _axis(verts_obb, faces, point, direction):
    up = ray_nearest(verts_obb, faces, point, direction)
    dn = ray_nearest(verts_obb, faces, point, -direction)
    if up is None or dn is None:
        raise ValueError("anatomic-neck axis: ray from the plane centre misses the mesh")
    return np.stack([up, dn])


def axis_normal(verts_obb, faces, point, normal):
    n = normal.copy()
    if n[2] < 0:
        n *= -1
    return _axis(verts_obb, faces, point, n)


def axis_central(verts_obb, faces, point, normal):
    n = normal.copy()
    if n[2] < 0:
        n *= -1
    n[2] = 0
    n = n / np.linalg.norm(n)
    return _axis(verts_obb, faces, point, n)


def plane_points(verts_ct, faces, origin, normal):
    """anatomic_neck.py:155-172: `mesh_ct.section(plane_origin, plane_normal).vertices` -- the crossing points of the
    CT mesh with the anatomic-neck plane, one per crossing triangle (every crossed mesh edge is the "downward" edge of
    exactly one triangle of a closed surface, so these are the unique vertices of the section path), unordered.
    trimesh `mesh_plane` semantics restated as in oracle/section.py, for a general plane: signed distance
    d = (v - origin) . n with n normalised, a vertex is "below" iff d < -1e-8, and the crossing point of an edge is
    computed from its lower to its higher vertex id: p = p_lo + d_lo / (d_lo - d_hi) * (p_hi - p_lo).  [3P unpinned]"""
    v = np.asarray(verts_ct, dtype=np.float64)
    f = np.asarray(faces, dtype=np.int64)
    o = np.asarray(origin, dtype=np.float64)
    n = np.asarray(normal, dtype=np.float64)
    n = n / np.sqrt(n[0] * n[0] + n[1] * n[1] + n[2] * n[2])
    rel = v - o
    d = (rel[:, 0] * n[0] + rel[:, 1] * n[1]) + rel[:, 2] * n[2]
    s = np.where(d < -1e-8, -1, 1)
    sf = s[f]
    cross = sf.min(axis=1) != sf.max(axis=1)
    f, sf = f[cross], sf[cross]
    dn = np.argmax((sf == 1) & (np.roll(sf, -1, axis=1) == -1), axis=1)       # the edge crossed downwards (+ -> -)
    r = np.arange(len(f))
    a, b = f[r, dn], f[r, (dn + 1) % 3]
    lo, hi = np.minimum(a, b), np.maximum(a, b)
    t = d[lo] / (d[lo] - d[hi])
    return v[lo] + t[:, None] * (v[hi] - v[lo])


def to_ct(pts_obb, T_obb):
    return transform_pts(pts_obb, inv_transform(T_obb))
