"""Host-side 4x4 frame construction for the facade (reference src/shoulder/utils.py:289-318
`construct_csys`, :227-256 `inv_transform`).  One 4x4 at a time -- glue, not a hot path; the
batched version runs on the device inside k_pack / k_te_orient."""
import numpy as np


def inv_transform(T):
    rot = np.identity(4)
    rot[:3, :3] = T[:3, :3]
    tr = np.identity(4)
    tr[:3, 3] = T[:3, 3]
    return np.linalg.inv(rot) @ np.linalg.inv(tr)


def construct_csys(vec_z, vec_y):
    vec_z = np.asarray(vec_z, dtype=np.float64)
    vec_y = np.asarray(vec_y, dtype=np.float64)
    origin = vec_z.mean(axis=0)
    z = vec_z[0] - vec_z[1]
    z /= np.linalg.norm(z)
    x = vec_y[0] - vec_y[1]
    x /= np.linalg.norm(x)
    y = np.cross(x, z)
    y /= np.linalg.norm(y)
    x = np.cross(y, z)
    x /= np.linalg.norm(x)
    T = np.identity(4)
    T[:3, 0], T[:3, 1], T[:3, 2], T[:3, 3] = x, y, z, origin
    if np.round(np.linalg.det(T)) == -1:
        T[:, 0] *= -1
    return inv_transform(T)


def unitxyz_to_spherical(xyz):
    """[r, theta (retroversion), phi (neck-shaft)] in mm and degrees (utils.py:321-330)."""
    xyz = np.asarray(xyz, dtype=np.float64)
    r = np.sqrt(np.sum(xyz ** 2))
    return np.array([r, np.rad2deg(np.arctan2(xyz[1], xyz[0])), np.rad2deg(np.arccos(xyz[2] / r))])


def spherical_to_unitxyz(sphr):
    """utils.py:333-339."""
    theta, phi = np.deg2rad(sphr[1]), np.deg2rad(sphr[2])
    return np.array([sphr[0] * np.sin(phi) * np.cos(theta), sphr[0] * np.sin(phi) * np.sin(theta), sphr[0] * np.cos(phi)])


def transform_plane_pn(point, normal, T):
    """(point, normal) of a plane under the 4x4 T (utils.py:191-206: the point moves, the normal only rotates)."""
    T = np.asarray(T, dtype=np.float64)
    p = T @ np.r_[np.asarray(point, dtype=np.float64).reshape(3), 1.0]
    return p[:3], T[:3, :3] @ np.asarray(normal, dtype=np.float64).reshape(3)
