"""Golden vectors for the glue of TransEpicondylar.axis from the reference's OWN code (src/shoulder/humerus/epicondyle.py:29-101
with utils.major_axis_dist / azimuth / _dist / construct_csys / transform_pts / inv_transform and slice.Slices._cutoff).

Run in the build container only (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_te_golden.py
shapely is absent; its RESULTS are injected: every distal slice carries a rectangle given by its 5 exterior corners
(`minimum_rotated_rectangle.exterior.xy`), `shapely.affinity.rotate / scale` hand their argument back, and
`polygon.difference(...)` returns given end pieces (objects with `.centroid.xy`).  What then runs unmodified is the
reference's choice of the row (first maximum of `utils.major_axis_dist`), its choice of the two pieces when there are more
than two (farthest centroids), the lift to 3-D, OBB -> CT, and the medial-first ordering in construct_csys(canal, central).
Output: tests/golden/te_golden.npz (inputs and outputs only).
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
sys.modules["shapely.affinity"].rotate = lambda g, angle: g
sys.modules["shapely.affinity"].scale = lambda g, xfact=1.0, yfact=1.0: g
sys.modules["shapely"].affinity = sys.modules["shapely.affinity"]
sys.path.insert(0, "/root/reference/src")

from shoulder.base import Transform  # noqa: E402
from shoulder.humerus import epicondyle as r_te  # noqa: E402
from shoulder.humerus import slice as r_slice  # noqa: E402


class _XY:
    def __init__(self, xy):
        self.xy = xy


class _Rect:
    def __init__(self, corners5):
        self.exterior = _XY((corners5[:, 0].copy(), corners5[:, 1].copy()))


class _Piece:
    def __init__(self, c):
        self.centroid = _XY((np.array([c[0]]), np.array([c[1]])))


class _Ends:
    def __init__(self, cents):
        self.geoms = [_Piece(c) for c in cents]


class _Polygon:
    def __init__(self, corners5, cents):
        self.minimum_rotated_rectangle = _Rect(corners5)
        self._cents = cents

    def difference(self, other):
        return _Ends(self._cents)


class _Slice:
    def __init__(self, corners5, cents):
        self.polygons_closed = [_Polygon(corners5, cents)]


class _Obb:
    pass


class StandInSlices:
    def __init__(self, slices, zs, T_obb):
        self._s, self._z, self.return_odd = slices, zs, False
        self.obb = _Obb()
        self.obb.transform = T_obb

    def zs(self, cutoff_pcts=None):
        return r_slice.Slices._cutoff(self, self._z, cutoff_pcts)

    def slices(self, cutoff_pcts=None):
        return r_slice.Slices._cutoff(self, self._s, cutoff_pcts)


class _Axis:
    def __init__(self, a):
        self._a = a

    def axis(self):
        return self._a

    def axis_central(self):
        return self._a


def rigid(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    T = np.identity(4)
    T[:3, :3], T[:3, 3] = q, rng.uniform(-300, 300, 3)
    return T


def rect(cx, cy, L, Wd, ang):
    u, v = np.array([np.cos(ang), np.sin(ang)]), np.array([-np.sin(ang), np.cos(ang)])
    c = np.array([cx, cy])
    p = np.array([c - u * L / 2 - v * Wd / 2, c + u * L / 2 - v * Wd / 2, c + u * L / 2 + v * Wd / 2, c - u * L / 2 + v * Wd / 2])
    return np.r_[p, p[:1]]


def main():
    rng = np.random.default_rng(5150)
    out = {"n": np.int64(4)}
    N = 200
    zs = np.linspace(0.99 * -150.0, 0.0, N)
    for c in range(4):
        T_obb = rigid(rng)
        L = 40 + 20 * np.exp(-((np.arange(N) - 12) / 6.0) ** 2) + rng.normal(0, 0.05, N)
        if c == 1:
            L[7] = L[16] = L.max() + 1.0                                    # two rows tie for the longest: the first one wins
        ang = rng.uniform(0, np.pi, N)
        rects = [rect(rng.normal(0, 2), rng.normal(0, 2), L[i], 0.45 * L[i], ang[i]) for i in range(N)]
        npieces = [2, 2, 3, 4][c]
        cents_all = []
        for i in range(N):
            u = np.array([np.cos(ang[i]), np.sin(ang[i])])
            base = [u * L[i] * 0.48 + rng.normal(0, 0.3, 2), -u * L[i] * 0.48 + rng.normal(0, 0.3, 2)]
            extra = [base[0] + rng.normal(0, 1.5, 2), base[1] + rng.normal(0, 1.5, 2)][: npieces - 2]
            cc = base + extra
            order = rng.permutation(len(cc))
            cents_all.append(np.array([cc[j] for j in order]))
        slices = [_Slice(rects[i], cents_all[i]) for i in range(N)]
        slc = StandInSlices(slices, zs, T_obb)
        canal = np.array([[0, 0, 150.0], [0, 0, -20.0]]) + rng.normal(0, 2, (2, 3))
        central = np.array([[20.0, 8.0, 160.0], [-18.0, -6.0, 160.0]]) + rng.normal(0, 2, (2, 3))
        W = rigid(rng)
        from shoulder import utils as rutils
        canal, central = rutils.transform_pts(canal, W), rutils.transform_pts(central, W)
        tf = Transform()
        te = r_te.TransEpicondylar(slc, _Axis(canal), _Axis(central), tf)
        ax = te.axis().copy()
        out.update({f"c{c}_T_obb": T_obb, f"c{c}_zs": zs, f"c{c}_rects": np.array(rects), f"c{c}_cents": np.array(cents_all), f"c{c}_canal": canal,
                    f"c{c}_central": central, f"c{c}_axis_ct": te._axis_ct.copy(), f"c{c}_axis": ax})
    np.savez_compressed(os.path.join(HERE, "te_golden.npz"), **out)
    print("te_golden.npz", out["c0_axis_ct"], out["c2_cents"].shape)


if __name__ == "__main__":
    main()
