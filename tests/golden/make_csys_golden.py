"""Golden vectors for `apply_csys_custom(from_ct=True / False)`, `apply_translation`, `apply_csys_ct` from the reference's OWN code
(src/shoulder/bone.py:66-105 with base.py:24-63 `Bone._update_landmark_data` / `Transform`, every landmark class's own
`transform_landmark`, utils.translate_transform / transform_pts).

Run in the build container only (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_csys_golden.py
Third-party modules are stubbed as in make_golden.py.  The bone is the reference's `Humerus` class created without its constructor
(the constructor needs trimesh); its landmark attributes are the reference's own landmark classes, also created bare, with their
CT caches (`_axis_ct`, `_points_ct`, ...) filled from arrays this script draws (seeded) -- from there on every number is computed by
reference code: the cumulative-matrix products of bone.py:92 and :100, the re-expression of every cached landmark, and the quirk
that `from_ct=False` / `apply_translation` apply the CUMULATIVE matrix to the ALREADY MOVED mesh (bone.py:94, :102).  The mesh is a
stand-in holding vertices with trimesh's two methods the code calls (`copy`, in-place `apply_transform` that returns the mesh),
its arithmetic = the reference's utils.transform_pts.  `AnatomicNeck._plane_ct` stays None (utils.transform_plane builds a
scikit-spatial object).  Only inputs and outputs are written (tests/golden/csys_golden.npz).
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

from shoulder import base as rbase  # noqa: E402
from shoulder import bone as rbone  # noqa: E402
from shoulder import utils as rutils  # noqa: E402
from shoulder.humerus import anatomic_neck, bicipital_groove, canal, epicondyle, surgical_neck  # noqa: E402


class TinyMesh:
    def __init__(self, vertices):
        self.vertices = np.array(vertices, dtype=np.float64)

    def copy(self):
        return TinyMesh(self.vertices.copy())

    def apply_transform(self, T):          # trimesh: in place, returns the mesh
        self.vertices = rutils.transform_pts(self.vertices, T)
        return self


class Obb:
    def __init__(self, verts):
        self._m = TinyMesh(verts)

    @property
    def mesh_ct(self):                     # mesh.py:28-33: a copy on every access
        return self._m.copy()


def bare(cls, tfrm, **attrs):
    o = cls.__new__(cls)
    o._tfrm = tfrm
    for k, v in attrs.items():
        o.__dict__[k] = v
    return o


def rigid(rng, scale=200.0):
    q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    T = np.identity(4)
    T[:3, :3], T[:3, 3] = q, rng.uniform(-scale, scale, 3)
    return T


def snapshot(b, tag, out):
    out[f"{tag}_transform"] = np.array(b.transform)
    out[f"{tag}_tfrm"] = np.array(b._tfrm.matrix)
    out[f"{tag}_mesh"] = np.array(b.mesh.vertices)
    out[f"{tag}_canal_axis"] = np.array(b.canal._axis)
    out[f"{tag}_canal_points"] = np.array(b.canal._points)
    out[f"{tag}_te_axis"] = np.array(b.trans_epiconylar._axis)
    out[f"{tag}_groove_axis"] = np.array(b.bicipital_groove._axis)
    out[f"{tag}_groove_points"] = np.array(b.bicipital_groove._points)
    out[f"{tag}_anp_points"] = np.array(b.anatomic_neck._points)
    out[f"{tag}_anp_plane_points"] = np.array(b.anatomic_neck._plane_points)
    out[f"{tag}_anp_axis_normal"] = np.array(b.anatomic_neck._normal_axis)
    out[f"{tag}_anp_axis_central"] = np.array(b.anatomic_neck._central_axis)
    out[f"{tag}_surgical_neck"] = np.array(b.surgical_neck.points)


def main():
    rng = np.random.default_rng(8405)
    W = rigid(rng, 600.0)                                   # a CT pose far from the origin (|coord| ~ 1e3 mm, like the fixtures)
    place = lambda a: rutils.transform_pts(np.asarray(a, dtype=np.float64), W)
    inp = {
        "verts": place(rng.uniform(-1, 1, (257, 3)) * [25.0, 30.0, 170.0]),
        "canal_axis": place([[0.5, -0.3, 90.0], [-0.4, 0.2, -60.0]]),
        "canal_points": place(np.c_[rng.normal(0, 0.6, (80, 2)), np.linspace(40, -50, 80)]),
        "te_axis": place([[32.0, 3.0, -150.0], [-29.0, -2.0, -148.0]]),
        "groove_axis": place([[11.0, 6.0, 150.0], [10.0, 5.0, 100.0]]),
        "groove_points": place(np.c_[10 + rng.normal(0, 0.4, 330), 5 + rng.normal(0, 0.4, 330), np.linspace(150, 100, 330)]),
        "anp_points": place(rng.uniform(-22, 22, (911, 3)) + [4.0, 2.0, 150.0]),
        "anp_plane_points": place(rng.uniform(-24, 24, (143, 3)) + [4.0, 2.0, 150.0]),
        "anp_axis_normal": place([[20.0, 12.0, 168.0], [-14.0, -9.0, 131.0]]),
        "anp_axis_central": place([[24.0, 15.0, 150.0], [-18.0, -11.0, 150.0]]),
        "surgical_neck": place(np.c_[18 * np.cos(np.linspace(0, 6.2, 97)), 16 * np.sin(np.linspace(0, 6.2, 97)), np.full(97, 118.0)]),
    }
    tf = rbase.Transform()
    b = rbone.Humerus.__new__(rbone.Humerus)
    b._tfrm = tf
    b.transform = tf.matrix
    b._obb = Obb(inp["verts"])
    b.mesh = b._obb.mesh_ct
    b.canal = bare(canal.Canal, tf, _axis_ct=inp["canal_axis"], _points_ct=inp["canal_points"])
    b.trans_epiconylar = bare(epicondyle.TransEpicondylar, tf, _axis_ct=inp["te_axis"])
    b.bicipital_groove = bare(bicipital_groove.DeepGroove, tf, _axis_ct=inp["groove_axis"], _points_ct=inp["groove_points"])
    b.anatomic_neck = bare(anatomic_neck.AnatomicNeck, tf, _points_ct=inp["anp_points"], _plane_ct=None, _plane_points_ct=inp["anp_plane_points"],
                           _normal_axis_ct=inp["anp_axis_normal"], _central_axis_ct=inp["anp_axis_central"])
    b.surgical_neck = bare(surgical_neck.SurgicalNeck, tf, points=inp["surgical_neck"].copy(), points_ct=inp["surgical_neck"].copy())
    assert len(b._list_landmarks()) == 5

    T1, T2, T3 = rigid(rng), rigid(rng, 40.0), rigid(rng, 15.0)
    t1, t2 = rng.uniform(-30, 30, 3), rng.uniform(-5, 5, 3)
    out = {"in_" + k: v for k, v in inp.items()}
    out.update({"T1": T1, "T2": T2, "T3": T3, "t1": t1, "t2": t2})
    # the sequence: ops[k] is applied, then everything is recorded under s{k}
    ops = [("custom_ct", "T1"), ("custom_rel", "T2"), ("translate", "t1"), ("custom_rel", "T3"), ("translate", "t2"), ("ct", ""),
           ("translate", "t1"), ("custom_ct", "T2"), ("custom_rel", "T1")]
    for k, (op, arg) in enumerate(ops):
        if op == "custom_ct":
            r = b.apply_csys_custom(out[arg].copy(), from_ct=True)          # bone.py:84-89
        elif op == "custom_rel":
            r = b.apply_csys_custom(out[arg].copy(), from_ct=False)         # bone.py:90-95
        elif op == "translate":
            r = b.apply_translation(out[arg].copy())                        # bone.py:97-105
        else:
            r = b.apply_csys_ct()                                           # bone.py:75-82
        out[f"s{k}_returned"] = np.array(r)
        snapshot(b, f"s{k}", out)
    out["ops"] = np.array([f"{op}:{arg}" for op, arg in ops])
    # Transform's validation (base.py:55-58)
    for bad in (np.identity(3), np.zeros((4, 4)).tolist()):
        try:
            b.apply_csys_custom(bad)
            raise SystemExit("the reference accepted a bad matrix")
        except ValueError as ex:
            assert str(ex) == "Invalid transformation matrix shape"
    np.savez_compressed(os.path.join(HERE, "csys_golden.npz"), **out)
    print("csys_golden.npz:", len(ops), "steps; |mesh| after the two cumulative quirks:", float(np.abs(out["s4_mesh"]).max()))


if __name__ == "__main__":
    main()
