"""Golden vectors for Side / RetroVersion / NeckShaft from the reference's OWN code
(src/shoulder/humerus/bone_props.py:12-112 with utils.construct_csys / transform_pts / unit_vector / unitxyz_to_spherical).

Run in the build container only (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_metrics_golden.py
Third-party modules are stubbed as in make_golden.py; the landmark objects handed to the reference classes are stand-ins
holding the arrays the classes read (`_axis_ct`, `_central_axis_ct`, `_normal_axis_ct`, `_points_ct`, `axis_normal()` in
the CURRENT csys -- the reference's quirk at bone_props.py:72-73).  Only inputs and outputs are written
(tests/golden/metrics_landmarks_golden.npz).
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
sys.path.insert(0, "/root/reference/src")

from shoulder import utils as rutils  # noqa: E402
from shoulder.humerus import bone_props  # noqa: E402


class Cn:
    def __init__(self, axis_ct):
        self._axis_ct = axis_ct

    def axis(self):
        return self._axis_ct


class Te(Cn):
    pass


class An:
    def __init__(self, central_ct, normal_ct, T_current):
        self._central_axis_ct, self._normal_axis_ct, self._T = central_ct, normal_ct, T_current

    def axis_central(self):
        return rutils.transform_pts(self._central_axis_ct, self._T)

    def axis_normal(self):
        return rutils.transform_pts(self._normal_axis_ct, self._T)


class Bg:
    def __init__(self, pts_ct):
        self._points_ct = pts_ct

    def points(self):
        return self._points_ct


def rigid(rng, scale=200):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    T = np.identity(4)
    T[:3, :3], T[:3, 3] = q, rng.uniform(-scale, scale, 3)
    return T


def main():
    rng = np.random.default_rng(4242)
    out = {}
    n = 8
    for c in range(n):
        # a humerus-like configuration in a random CT pose: canal along ~z, head axis tilted ~45 deg, TE axis ~ along x
        W = rigid(rng)
        canal = np.array([[0, 0, 150.0], [0, 0, -20.0]]) + rng.normal(0, 2, (2, 3))
        d = np.array([np.cos(rng.uniform(-1.2, 1.2)), np.sin(rng.uniform(-1.2, 1.2)), 1.0])
        d /= np.linalg.norm(d)
        c0 = np.array([5.0, 3.0, 160.0]) + rng.normal(0, 3, 3)
        normal = np.stack([c0 + 25 * d, c0 - 25 * d])
        dc = np.array([d[0], d[1], 0.0]) / np.linalg.norm(d[:2])
        central = np.stack([c0 + 22 * dc, c0 - 22 * dc])
        te = np.array([[35.0, 2.0, -150.0], [-30.0, -3.0, -148.0]]) + rng.normal(0, 2, (2, 3))
        ang = rng.uniform(0, 2 * np.pi)
        bg = np.c_[12 * np.cos(ang) + rng.normal(0, 0.5, 40), 12 * np.sin(ang) + rng.normal(0, 0.5, 40), np.linspace(100, 150, 40)]
        canal, normal, central, te, bg = (rutils.transform_pts(a, W) for a in (canal, normal, central, te, bg))
        T_cur = np.identity(4) if c % 2 == 0 else rigid(rng)
        cn, an, tee, bgo = Cn(canal), An(central, normal, T_cur), Te(te), Bg(bg)
        side = bone_props.Side(cn, an, bgo).calc()
        retro = bone_props.RetroVersion(cn, an, tee, lambda s=side: s).calc()
        ns = bone_props.NeckShaft(cn, an).calc()
        out.update({f"c{c}_canal": canal, f"c{c}_normal": normal, f"c{c}_central": central, f"c{c}_te": te, f"c{c}_groove": bg, f"c{c}_T_current": T_cur,
                    f"c{c}_side": np.array(side), f"c{c}_retroversion": np.float64(retro), f"c{c}_neckshaft": np.float64(ns)})
    out["n"] = np.int64(n)
    np.savez_compressed(os.path.join(HERE, "metrics_landmarks_golden.npz"), **out)
    print("metrics_landmarks_golden.npz", [(str(out[f"c{c}_side"]), round(float(out[f"c{c}_retroversion"]), 3), round(float(out[f"c{c}_neckshaft"]), 3)) for c in range(n)])


if __name__ == "__main__":
    main()
