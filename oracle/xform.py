"""K26/K27: 4x4 affine helpers (oracle; test infrastructure).

Restates reference `src/shoulder/utils.py`:
  transform_pts       utils.py:172-188
  transform_normal    utils.py:191-206 (normal part of transform_plane)
  inv_transform       utils.py:227-256
  translate_transform utils.py:259-264
  unit_vector         utils.py:267-271
  construct_csys      utils.py:289-318
Pinned by tests/golden/utils_golden.npz (captured from the reference's own code).
"""
import numpy as np


def transform_pts(pts, T):
    pts = np.asarray(pts, dtype=np.float64).reshape(-1, 3)
    h = np.c_[pts, np.ones(len(pts))].T
    return np.matmul(T, h).T[:, :3]


def transform_normal(normal, T):
    return np.matmul(T[:3, :3], np.asarray(normal, dtype=np.float64).reshape(3, 1)).ravel()


def inv_transform(T):
    translate = np.identity(4)
    translate[:3, 3] = T[:3, 3]
    rotate = np.c_[T[:, :3], np.array([[0.0], [0.0], [0.0], [1.0]])]
    return np.matmul(np.linalg.inv(rotate), np.linalg.inv(translate))


def translate_transform(t):
    T = np.identity(4)
    T[:3, 3] = np.asarray(t, dtype=np.float64).reshape(3)
    return T


def unit_vector(p1, p2):
    v = np.asarray(p1, dtype=np.float64) - np.asarray(p2, dtype=np.float64)
    return v / np.linalg.norm(v)


def construct_csys(vec_z, vec_y):
    vec_z = np.asarray(vec_z, dtype=np.float64)
    vec_y = np.asarray(vec_y, dtype=np.float64)
    pos = np.average(vec_z, axis=0).flatten()
    z_hat = unit_vector(vec_z[0], vec_z[1])
    x_hat = unit_vector(vec_y[0], vec_y[1])
    y_hat = np.cross(x_hat, z_hat)
    y_hat /= np.linalg.norm(y_hat)
    x_hat = np.cross(y_hat, z_hat)
    x_hat /= np.linalg.norm(x_hat)
    T = np.c_[x_hat, y_hat, z_hat, pos]
    T = np.r_[T, np.array([0.0, 0.0, 0.0, 1.0]).reshape(1, 4)]
    if np.round(np.linalg.det(T)) == -1:
        T[:, 0] *= -1
    return inv_transform(T)
