"""Golden vectors for Canal.points / Canal.axis / Canal.get_transform and DeepGroove.axis from the reference's OWN code
(src/shoulder/humerus/canal.py:19-124, bicipital_groove.py:244-265, with slice.Slices._cutoff and utils.transform_pts / inv_transform / unit_vector).

Run in the build container only (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_canal_golden.py
Third-party modules are stubbed as in make_golden.py.  `skspatial.objects.Line.best_fit` is the one third-party call on
this path; the stub implements its published algorithm (scikit-spatial 6.8.1: centre the points, `np.linalg.svd`, direction
= first right-singular vector, point = centroid), so these vectors pin everything AROUND the fit -- slice selection, the
[centroid, z] rows, the proximal flip, the end points at z_length * mean(cutoff) / 2, the frame of get_transform -- and the
fit itself only as far as that restatement goes.  The slices object is a stand-in that holds per-slice centroids and z's
and cuts them with the reference's own `Slices._cutoff`.  Output: tests/golden/canal_golden.npz (inputs and outputs only).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.dont_write_bytecode = True


class _Stub(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        child = _Stub(f"{self.__name__}.{name}")
        setattr(self, name, child)
        return child

    def __call__(self, *a, **k):
        raise RuntimeError(f"third-party stub {self.__name__} was called")

    def __or__(self, other):
        return self

    __ror__ = __or__


for _n in ["trimesh", "trimesh.geometry", "skspatial", "skspatial.objects", "circle_fit", "ruptures",
           "onnxruntime", "ellipse", "shapely", "shapely.affinity", "rtree"]:
    sys.modules[_n] = _Stub(_n)


def Points(x):
    return np.asarray(x, dtype=np.float64)


class Line:
    def __init__(self, point, direction):
        self.point, self.direction = np.array(point, dtype=np.float64), np.array(direction, dtype=np.float64)

    @classmethod
    def best_fit(cls, points):
        c = points.mean(axis=0)
        _, _, vh = np.linalg.svd(points - c)
        return cls(c, vh[0])


sys.modules["skspatial.objects"].Points = Points
sys.modules["skspatial.objects"].Line = Line
sys.modules["skspatial"].objects = sys.modules["skspatial.objects"]
sys.path.insert(0, "/root/reference/src")

from shoulder.base import Transform  # noqa: E402
from shoulder.humerus import bicipital_groove as r_bg  # noqa: E402
from shoulder.humerus import canal as r_canal  # noqa: E402
from shoulder.humerus import slice as r_slice  # noqa: E402


class _S:
    def __init__(self, c):
        self.centroid = c


class _Obb:
    pass


class StandInSlices:
    def __init__(self, centroids, zs, T_obb, z_length):
        self._c, self._z = centroids, zs
        self.return_odd = False               # the default of every Slices subclass (slice.py:11, :215)
        self.obb = _Obb()
        self.obb.transform, self.obb.z_length = T_obb, z_length

    def zs(self, cutoff_pcts=None):
        return r_slice.Slices._cutoff(self, self._z, cutoff_pcts)

    def slices(self, cutoff_pcts=None):
        return [_S(c) for c in r_slice.Slices._cutoff(self, self._c, cutoff_pcts)]


def rigid(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    T = np.identity(4)
    T[:3, :3], T[:3, 3] = q, rng.uniform(-300, 300, 3)
    return T


def main():
    rng = np.random.default_rng(31)
    out = {"n": np.int64(4)}
    for c in range(4):
        N = 200
        zs = np.linspace(0.99 * 160, 0.99 * -150, N)
        tilt = rng.normal(0, 0.02, 2) * (1 if c != 2 else -1)
        cent = np.c_[tilt[0] * zs + rng.normal(0, 0.3, N), tilt[1] * zs + rng.normal(0, 0.3, N)]
        T_obb, T_cur = rigid(rng), (np.identity(4) if c % 2 == 0 else rigid(rng))
        cut = (0.35, 0.75) if c < 3 else (0.3, 0.8)
        slc = StandInSlices(cent, zs, T_obb, 310.0 + c)
        tf = Transform()
        tf.matrix = T_cur
        cn = r_canal.Canal(slc, tf)
        pts = cn.points(cut).copy()
        axis = cn.axis(cut).copy()
        Tc = cn.get_transform()
        out.update({f"c{c}_centroids": cent, f"c{c}_zs": zs, f"c{c}_T_obb": T_obb, f"c{c}_T_current": T_cur, f"c{c}_z_length": np.float64(310.0 + c),
                    f"c{c}_cutoff": np.array(cut), f"c{c}_points": pts, f"c{c}_points_ct": cn._points_ct, f"c{c}_axis": axis, f"c{c}_axis_ct": cn._axis_ct,
                    f"c{c}_get_transform": Tc})
    # DeepGroove.axis (bicipital_groove.py:244-265) on injected groove points: ends = fit point +- direction * (z range / 2),
    # RAW sign of the fitted direction (no flip), OBB -> CT -> current csys
    for c in range(3):
        n = 330
        z = np.linspace(95.0, 20.0, n)
        pts_obb = np.c_[14 + 0.03 * z + rng.normal(0, 0.4, n), -6 + 0.01 * z + rng.normal(0, 0.4, n), z]
        if c == 1:
            pts_obb = pts_obb[::-1].copy()
        T_obb, T_cur = rigid(rng), (np.identity(4) if c == 0 else rigid(rng))
        bg = r_bg.DeepGroove.__new__(r_bg.DeepGroove)
        slc = StandInSlices(None, None, T_obb, 0.0)
        tf = Transform()
        tf.matrix = T_cur
        bg._slc, bg._tfrm, bg._axis_ct = slc, tf, None
        bg._points_obb, bg._points_ct = pts_obb, pts_obb      # (_points_ct only has to be "already computed")
        ax = bg.axis().copy()
        out.update({f"g{c}_points_obb": pts_obb, f"g{c}_T_obb": T_obb, f"g{c}_T_current": T_cur, f"g{c}_axis": ax, f"g{c}_axis_ct": bg._axis_ct})
    out["n_groove"] = np.int64(3)
    np.savez_compressed(os.path.join(HERE, "canal_golden.npz"), **out)
    print("canal_golden.npz", out["c0_points"].shape, out["c3_points"].shape, out["c0_axis"])


if __name__ == "__main__":
    main()
