"""Golden vectors for ProxObb._obb and FullObb._obb from the reference's OWN code (src/shoulder/humerus/mesh.py:133-192: z
grid, head-end decision, flip, savgol + gradient, longest consecutive run, cut-off fractions; :63-125: circle-fit residual
comparison of the two ends, flip, composed transform).

Run in the build container only (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_prox_golden.py
Third-party modules are stubbed as in make_golden.py.  The trimesh object the method works on is a stand-in that returns
a given box transform from `apply_obb()`, given `bounds`, and for the k-th `section(...)` call the k-th entry of a given
area profile (so the reference's own scan loop, argmax, flip, filter and run logic execute unmodified).  Profiles: the
area scan of tests/golden/bones/proximal_left_cut.stl (from the oracle), the same reversed (head at -z), and synthetic
profiles with a ragged end / two plateaus.  Output: tests/golden/prox_golden.npz (inputs and outputs only).
"""
import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
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

from shoulder.humerus import mesh as r_mesh  # noqa: E402


class _Planar:
    def __init__(self, area):
        self.area = area
        self.vertices = np.zeros((3, 2))      # FullObb hands these to circle_fit, which is stubbed below


class _Section:
    def __init__(self, area):
        self._a = area

    def to_planar(self):
        return _Planar(self._a), None


class StandInMesh:
    def __init__(self, T_obb, zmin, zmax, areas):
        self._T, self._areas, self._k = T_obb, list(areas), 0
        self.bounds = np.array([[-20.0, -25.0, zmin], [20.0, 25.0, zmax]])
        self.applied = []

    def apply_obb(self):
        return self._T

    def section(self, plane_origin, plane_normal):
        a = self._areas[self._k]
        self._k += 1
        return _Section(a)

    def apply_transform(self, T):
        self.applied.append(np.array(T, dtype=np.float64))
        return self


def main():
    from oracle import prox as o_prox
    from shoulder_amd.stl import load_stl
    rng = np.random.default_rng(99)
    v, f = load_stl(os.path.join(HERE, "bones", "proximal_left_cut.stl"))
    real = o_prox.prox_obb(v.astype(np.float64), f)
    scan = real["z_area"][::-1] if real["flipped"] else real["z_area"]       # in the order the scan produced it
    zb = real["z_bounds"]
    x = np.arange(100)
    synth1 = 300 + 5 * np.sin(x / 7.0) + 900 * np.exp(-((x - 85) / 8.0) ** 2)
    synth1[:6] = np.linspace(40, 290, 6)                                       # ragged cut at the low end
    synth2 = 320 + 3 * np.cos(x / 5.0) + 1000 * np.exp(-((x - 12) / 7.0) ** 2)      # head at the low end
    synth2[60:64] += 80                                                        # a bump that splits the plateau
    profiles = [(scan, zb), (scan[::-1].copy(), (-zb[1], -zb[0])), (synth1, (-90.0, 95.0)), (synth2, (-100.0, 92.0))]
    out = {"n": np.int64(len(profiles))}
    for c, (areas, (zmin, zmax)) in enumerate(profiles):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        T = np.identity(4)
        T[:3, :3], T[:3, 3] = q, rng.uniform(-100, 100, 3)
        m = StandInMesh(T, zmin, zmax, areas)
        po = r_mesh.ProxObb.__new__(r_mesh.ProxObb)
        po.__dict__["mesh"] = m                                               # cached_property slot (mesh.py:36-41)
        transform, cutoff_pcts = po._obb()
        out.update({f"c{c}_areas": np.asarray(areas, dtype=np.float64), f"c{c}_T_obb": T, f"c{c}_zmin": np.float64(zmin), f"c{c}_zmax": np.float64(zmax),
                    f"c{c}_transform": transform, f"c{c}_cutoff_pcts": np.array(cutoff_pcts, dtype=np.float64), f"c{c}_cutoff_bot": np.int64(po.cutoff_bot),
                    f"c{c}_z_length": np.float64(po.z_length), f"c{c}_flipped": np.bool_(len(m.applied) > 0)})
    # FullObb._obb (mesh.py:63-125) with the circle fit stubbed to return given residuals for the -z end and the +z end:
    # which end is the head, the flip, the composed transform, z_length
    full = [((-150.0, 160.0), (1.0, 2.0)), ((-150.0, 160.0), (2.0, 1.0)), ((-150.0, 160.0), (1.5, 1.5)), ((-140.0, 170.0), (0.3, 0.30000001))]
    for c, ((zmin, zmax), res) in enumerate(full):
        q, _ = np.linalg.qr(rng.standard_normal((3, 3)))
        T = np.identity(4)
        T[:3, :3], T[:3, 3] = q, rng.uniform(-100, 100, 3)
        m = StandInMesh(T, zmin, zmax, [0.0, 0.0])
        seq = list(res)
        r_mesh.circle_fit.least_squares_circle = lambda pts, seq=seq: (0.0, 0.0, 1.0, seq.pop(0))
        fo = r_mesh.FullObb.__new__(r_mesh.FullObb)
        fo.__dict__["mesh"] = m
        transform = fo._obb()
        out.update({f"f{c}_T_obb": T, f"f{c}_zmin": np.float64(zmin), f"f{c}_zmax": np.float64(zmax), f"f{c}_residus": np.array(res), f"f{c}_transform": transform,
                    f"f{c}_z_length": np.float64(fo.z_length), f"f{c}_flipped": np.bool_(len(m.applied) > 0)})
    out["n_full"] = np.int64(len(full))
    np.savez_compressed(os.path.join(HERE, "prox_golden.npz"), **out)
    print("prox_golden.npz", [(bool(out[f"c{c}_flipped"]), out[f"c{c}_cutoff_pcts"].tolist(), int(out[f"c{c}_cutoff_bot"])) for c in range(len(profiles))])


if __name__ == "__main__":
    main()
