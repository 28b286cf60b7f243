"""Golden vectors for HumeralHeadOsteotomy from the reference's OWN code (src/shoulder/arthroplasty.py:13-175 with
utils.py:191-206, :227-256, :321-339).

Run in the build container only (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_osteotomy_golden.py
Same mechanism as make_golden.py: the absent third-party modules are stubbed; `skspatial.objects.Plane` is given the one
behaviour the module relies on (it stores `point` / `normal` as fresh arrays).  The humerus handed to the reference class
is a stand-in with the attributes the class touches (`_tfrm`, the apply_csys_* methods, `anatomic_neck.plane()`, `side()`)
built from the reference's own `Transform` and `utils.transform_plane`.  Only inputs and outputs are written
(tests/golden/osteotomy_golden.npz): starting matrices, the anatomic-neck plane in CT, the side, a fixed script of
operations, and after every operation the resection plane in the caller's csys plus neckshaft_rel / retroversion_rel.
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


class Plane:      # skspatial.objects.Plane as far as arthroplasty.py / utils.transform_plane use it
    def __init__(self, point, normal):
        self.point = np.array(point, dtype=np.float64)
        self.normal = np.array(normal, dtype=np.float64)


sys.modules["skspatial.objects"].Plane = Plane
sys.modules["skspatial"].objects = sys.modules["skspatial.objects"]
sys.path.insert(0, "/root/reference/src")

from shoulder import utils as rutils  # noqa: E402
from shoulder.arthroplasty import HumeralHeadOsteotomy  # noqa: E402
from shoulder.base import Transform  # noqa: E402


class _Neck:
    def __init__(self, bone, point_ct, normal_ct):
        self._b, self._plane_ct = bone, Plane(point_ct, normal_ct)

    def plane(self):
        return rutils.transform_plane(self._plane_ct, self._b._tfrm.matrix)


class StandInHumerus:
    def __init__(self, T_start, T_anp, point_ct, normal_ct, side):
        self._tfrm = Transform()
        self._tfrm.matrix = T_start
        self._T_anp, self._side = T_anp, side
        self.anatomic_neck = _Neck(self, point_ct, normal_ct)

    def side(self):
        return self._side

    def apply_csys_canal_articular(self):
        self._tfrm.matrix = self._T_anp.copy()

    def apply_csys_ct(self):
        self._tfrm.reset()

    def apply_csys_custom(self, T, from_ct=True):
        self._tfrm.matrix = T


def rigid(rng):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    T = np.identity(4)
    T[:3, :3], T[:3, 3] = q, rng.uniform(-200, 200, 3)
    return T


SCRIPT = [("offset_retroversion", 10.0), ("offest_neckshaft", 5.0), ("offset_depth", 2.0, "canal"), ("offset_depth", 1.5, "anp"),
          ("offset_depth", -1.0, "resection"), ("offset_anterior_posterior", 1.0), ("offset_medial_lateral", 1.5),
          ("read_retroversion_rel",), ("read_retroversion_rel",), ("offset_retroversion", -25.0), ("offest_neckshaft", -12.5), ("move", 0)]


def main():
    rng = np.random.default_rng(777)
    out = {"script": np.array([repr(s) for s in SCRIPT])}
    for case, side in enumerate(["left", "right", "left"]):
        T_start = np.identity(4) if case == 0 else rigid(rng)
        T_anp, T_move = rigid(rng), rigid(rng)
        normal_ct = rng.standard_normal(3)
        normal_ct /= np.linalg.norm(normal_ct)
        point_ct = rng.uniform(-50, 50, 3)
        hum = StandInHumerus(T_start.copy(), T_anp, point_ct, normal_ct, side)
        ost = HumeralHeadOsteotomy(hum)
        after_init = hum._tfrm.matrix.copy()      # the constructor hands the caller's csys back (arthroplasty.py:27-31)
        rows = []

        def snap():
            p = ost.plane
            rows.append(np.r_[np.asarray(p.point, dtype=np.float64), np.asarray(p.normal, dtype=np.float64), float(ost.neckshaft_rel)])
        snap()
        retro = []
        for step in SCRIPT:
            if step[0] == "read_retroversion_rel":
                retro.append(float(ost.retroversion_rel))
            elif step[0] == "move":
                hum.apply_csys_custom(T_move.copy())
            else:
                getattr(ost, step[0])(*step[1:])
            snap()
        out.update({f"c{case}_T_start": T_start, f"c{case}_T_anp": T_anp, f"c{case}_T_move": T_move, f"c{case}_point_ct": point_ct,
                    f"c{case}_normal_ct": normal_ct, f"c{case}_side": np.array(side), f"c{case}_rows": np.array(rows), f"c{case}_retro": np.array(retro),
                    f"c{case}_transform_after_init": after_init})
    np.savez_compressed(os.path.join(HERE, "osteotomy_golden.npz"), **out)
    print("osteotomy_golden.npz", {k: np.asarray(v).shape for k, v in out.items() if k.startswith("c0")})


if __name__ == "__main__":
    main()
