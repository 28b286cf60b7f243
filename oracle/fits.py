"""K11/K21/K22/K23: line / plane / ellipse fits and ray casts (oracle; test infrastructure).

Third-party arithmetic restated from the pinned versions (absent from this image;
parity UNPINNED):
  line_best_fit   scikit-spatial 6.8.1 `Line.best_fit`   (canal.py:66, bicipital_groove.py:252)
  plane_best_fit  scikit-spatial 6.8.1 `Plane.best_fit`  (anatomic_neck.py:128)
  ellipse_center  lsq-ellipse 2.2.1 `LsqEllipse.fit/as_parameters` (anatomic_neck.py:139-144)
  plane_basis     stands in for `trimesh.geometry.plane_transform` (anatomic_neck.py:138);
                  any in-plane rotation gives the same ellipse centre in 3-D
  ray_nearest     `mesh.ray.intersects_location` (anatomic_neck.py:184-191, :217-224);
                  canonical rule B-6: nearest hit only
"""
import numpy as np


def line_best_fit(P):
    P = np.asarray(P, dtype=np.float64)
    c = P.mean(axis=0)
    _, _, vh = np.linalg.svd(P - c)
    return c, vh[0]


def plane_best_fit(P):
    P = np.asarray(P, dtype=np.float64)
    c = P.mean(axis=0)
    u, _, _ = np.linalg.svd((P - c).T)
    return c, u[:, 2]


def ellipse_center(xy):
    x, y = np.asarray(xy, dtype=np.float64).T
    D1 = np.vstack([x ** 2, x * y, y ** 2]).T
    D2 = np.vstack([x, y, np.ones_like(x)]).T
    S1, S2, S3 = D1.T @ D1, D1.T @ D2, D2.T @ D2
    C1 = np.array([[0.0, 0.0, 2.0], [0.0, -1.0, 0.0], [2.0, 0.0, 0.0]])
    M = np.linalg.inv(C1) @ (S1 - S2 @ np.linalg.inv(S3) @ S2.T)
    _, vec = np.linalg.eig(M)
    vec = np.real(vec)
    cond = 4 * vec[0, :] * vec[2, :] - vec[1, :] ** 2
    a1 = vec[:, np.nonzero(cond > 0)[0]]
    if a1.shape[1] == 0:
        raise ValueError("ellipse fit: no admissible eigenvector")
    a1 = a1[:, :1]
    a2 = np.linalg.inv(-S3) @ S2.T @ a1
    a, b, c, d, f = a1[0, 0], a1[1, 0] / 2, a1[2, 0], a2[0, 0] / 2, a2[1, 0] / 2
    den = b ** 2 - a * c
    return np.array([(c * d - b * f) / den, (a * f - b * d) / den])


def plane_basis(normal):
    n = np.asarray(normal, dtype=np.float64)
    n = n / np.linalg.norm(n)
    k = int(np.argmin(np.abs(n)))
    e = np.zeros(3)
    e[k] = 1.0
    u = np.cross(n, e)
    u /= np.linalg.norm(u)
    return u, np.cross(n, u), n


def ray_nearest(verts, faces, origin, direction):
    """Nearest forward hit of one ray with a triangle mesh (Moller-Trumbore, fp64).
    -> (3,) hit point, or None."""
    v = np.asarray(verts, dtype=np.float64)
    o = np.asarray(origin, dtype=np.float64)
    d = np.asarray(direction, dtype=np.float64)
    a, b, c = v[faces[:, 0]], v[faces[:, 1]], v[faces[:, 2]]
    e1, e2 = b - a, c - a
    pv = np.cross(d, e2)
    det = np.einsum("ij,ij->i", e1, pv)
    ok = np.abs(det) > 1e-12
    inv = np.where(ok, 1.0 / np.where(ok, det, 1.0), 0.0)
    tv = o - a
    u = np.einsum("ij,ij->i", tv, pv) * inv
    qv = np.cross(tv, e1)
    w = np.einsum("j,ij->i", d, qv) * inv
    t = np.einsum("ij,ij->i", e2, qv) * inv
    hit = ok & (u >= 0) & (w >= 0) & (u + w <= 1) & (t > 1e-9)
    if not hit.any():
        return None
    t = np.where(hit, t, np.inf)
    return o + d * t.min()
