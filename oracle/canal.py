"""a11/a12: humeral canal (oracle; test infrastructure).

Restates reference `src/shoulder/humerus/canal.py`:
  canal_points :19-56  ([slice AABB centre, z] for the cut slices -> CT)
  canal_axis   :58-85  (line fit in the OBB frame, flipped proximally, endpoints
                        mean +- dir * z_length*mean(cutoff)/2 -> CT; row 0 proximal)
Pinned by tests/golden/canal_golden.npz: the reference's own canal.py run on a stand-in slices object
(tests/golden/make_canal_golden.py; `Line.best_fit` stubbed with the published SVD algorithm).
"""
import numpy as np

from .fits import line_best_fit
from .slices import cutoff_range
from .xform import inv_transform, transform_pts


def canal_points(centroids_all, zs_all, T_obb, cutoff=(0.35, 0.75)):
    a, b = cutoff_range(len(zs_all), cutoff)
    pts_obb = np.c_[centroids_all[a:b], zs_all[a:b]]
    return pts_obb, transform_pts(pts_obb, inv_transform(T_obb))


def canal_axis(points_obb, z_length, T_obb, cutoff=(0.35, 0.75)):
    mid, d = line_best_fit(points_obb)
    if d[-1] < 0:
        d = d * -1
    half = z_length * np.mean(cutoff) / 2
    pts = np.array([mid + d * half, mid - d * half])
    return pts, transform_pts(pts, inv_transform(T_obb))
