"""Host-side 4x4 frame construction for the facade (reference src/shoulder/utils.py:289-318
`construct_csys`, :227-256 `inv_transform`).  One 4x4 at a time -- glue, not a hot path; the
batched version runs on the device inside k_pack / k_te_final."""
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
