"""Generate golden vectors from the reference's OWN NumPy/SciPy/sklearn code.

Run in the build container only (needs /root/reference; never runs on the GPU box):
    python tests/golden/make_golden.py
The reference (`/root/reference/src/shoulder`) cannot be imported as-is: trimesh,
shapely, rtree, scikit-spatial, circle-fit, ruptures, onnxruntime, lsq-ellipse are
not installed (ordinary ModuleNotFoundError).  Following SURVEY App. G, stub modules
are registered for exactly those names; the reference's own arithmetic then runs
unmodified.  Only inputs and outputs are written (tests/golden/*.npz) -- no reference
source or bytecode is copied.

What is captured (reference file:line):
  utils_golden.npz   utils.transform_pts :172-188, inv_transform :227-256,
                     construct_csys :289-318, translate_transform :259-264,
                     unit_vector :267-271, _azimuth :50-55, major_axis_dist :89-97
  slice_golden.npz   Slices._cutoff slice.py:157-164, _resample_polygon :166-189,
                     _cart2pol_no_sort :200-206, _ixy_centered :85-87, _itr_start
                     :102-108, _itr_centered_start :136-144
  groove_golden.npz  DeepGroove.points() bicipital_groove.py:26-242 on injected contours
                     (onnxruntime.InferenceSession replaced by a walker over the decoded
                     rfc_bg3.onnx tables)
  anp_golden.npz     AnatomicNeck.points() anatomic_neck.py:31-121: the float32 image fed
                     to the network and the edge points, with the network replaced by
                     supplied logits
Contour inputs are the oracle's proximal contours of humerus_left.stl rounded to
float32 (so the committed inputs are exactly what the reference code was fed).
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

import shoulder  # noqa: E402
from shoulder import utils as rutils  # noqa: E402
from shoulder.base import Transform  # noqa: E402
from shoulder.humerus import anatomic_neck as r_anp  # noqa: E402
from shoulder.humerus import bicipital_groove as r_bg  # noqa: E402
from shoulder.humerus import slice as r_slice  # noqa: E402

from oracle import rfc as o_rfc  # noqa: E402
from oracle import unet as o_unet  # noqa: E402


def golden_utils():
    rng = np.random.default_rng(20240)
    out = {}
    R = np.linalg.qr(rng.standard_normal((3, 3)))[0]
    T = np.identity(4)
    T[:3, :3] = R
    T[:3, 3] = rng.uniform(-500, 500, 3)
    pts = rng.uniform(-300, 300, (17, 3))
    out["T"], out["pts"] = T, pts
    out["transform_pts"] = rutils.transform_pts(pts, T)
    out["inv_transform"] = rutils.inv_transform(T)
    vz = rng.uniform(-200, 200, (2, 3))
    vy = rng.uniform(-200, 200, (2, 3))
    out["vz"], out["vy"] = vz, vy
    out["construct_csys"] = rutils.construct_csys(vz, vy)
    out["construct_csys_swapped"] = rutils.construct_csys(vz, vy[::-1])
    t = rng.uniform(-10, 10, 3)
    out["t"], out["translate_transform"] = t, rutils.translate_transform(t)
    out["unit_vector"] = rutils.unit_vector(vz[0], vz[1])
    p = rng.uniform(-50, 50, (6, 2, 2))
    out["az_pts"] = p
    out["azimuth"] = np.array([rutils._azimuth(a, b) for a, b in p])

    class _MRR:
        def __init__(self, xy):
            self.exterior = types.SimpleNamespace(xy=(xy[:, 0], xy[:, 1]))

    c, s = np.cos(0.37), np.sin(0.37)
    rect = np.array([[0, 0], [30, 0], [30, 11], [0, 11], [0, 0]], dtype=float) @ np.array([[c, s], [-s, c]]) + 3.0
    out["rect"] = rect
    out["major_axis_dist"] = rutils.major_axis_dist(_MRR(rect))
    out["rect_azimuth"] = rutils.azimuth(_MRR(rect))
    np.savez_compressed(os.path.join(HERE, "utils_golden.npz"), **out)


class _Obb:
    def __init__(self, T):
        self.transform = T


class _FakeSlices(r_slice.Slices):
    def __init__(self, zs, ixy, centroids, T):
        self._interp_num = ixy.shape[2]
        self.return_odd = False
        self.obb = _Obb(T)
        self.__dict__["_zs"] = zs
        self.__dict__["_ixy"] = ixy
        self.__dict__["_centroids"] = centroids

    @property
    def _zs(self):  # abstract in the reference; the instance dict entry shadows nothing for a
        return self.__dict__["_zs"]  # plain property, so read it back explicitly


def golden_slices(ixy, cents, zs, T):
    out = {}
    s = _FakeSlices(zs, ixy, cents, T)
    cases = [(200, (0.35, 0.75)), (200, (0.70, 0.99)), (600, (0.2, 0.75)), (600, (0.0, 0.852)),
             (200, (0.8, 0.99)), (100, (0.5, 0.8)), (600, (0.1, 0.9))]
    out["cutoff_cases"] = np.array([[n, c0, c1] for n, (c0, c1) in cases])
    out["cutoff_ranges"] = np.array([[int(s._cutoff(np.arange(n), c)[0]), int(s._cutoff(np.arange(n), c)[-1]) + 1]
                                     for n, c in cases])
    rng = np.random.default_rng(7)
    ang = np.sort(rng.uniform(0, 2 * np.pi, 57))
    rad = 20 + 3 * np.sin(3 * ang) + rng.uniform(-0.5, 0.5, 57)
    poly = np.c_[rad * np.cos(ang), rad * np.sin(ang)]
    poly = np.r_[poly, poly[:1]]
    out["poly"] = poly
    out["resample_100"] = s._resample_polygon(poly, 100)
    out["resample_512"] = s._resample_polygon(poly, 512)
    out["cart2pol_no_sort"] = s._cart2pol_no_sort(poly[:, 0], poly[:, 1])
    rows = [0, 150, 333, 599]
    out["rows"] = np.array(rows)
    out["ixy_centered_rows"] = s._ixy_centered[rows]
    out["itr_start_rows"] = s._itr_start[rows]
    out["itr_centered_start_rows"] = s._itr_centered_start[rows]
    np.savez_compressed(os.path.join(HERE, "slice_golden.npz"), **out)
    return s


class _FakeCanal:
    def __init__(self, axis):
        self._a = axis

    def axis(self):
        return self._a


class _FakeSession:
    """Stands in for onnxruntime.InferenceSession: RFC by table walk, UNet by supplied weights."""
    tables = None
    unet_w = None
    captured = {}

    def __init__(self, model_bytes, providers=None):
        pass

    def get_inputs(self):
        return [types.SimpleNamespace(name="input")]

    def run(self, _, feeds):
        if "X" in feeds:
            p1 = o_rfc.predict_proba1(self.tables, feeds["X"])
            proba = np.c_[np.float32(1) - p1, p1].astype(np.float32)
            return [(p1 > 0.5).astype(np.int64), proba]
        img = feeds["input"]
        _FakeSession.captured["image"] = img.copy()
        lg = o_unet.forward_f64(self.unet_w, img[0, 0]).astype(np.float32)
        _FakeSession.captured["logits"] = lg
        return [lg.reshape(1, 1, *lg.shape)]


def golden_groove_anp(s, canal_axis_ct, tag, with_anp):
    sys.modules["onnxruntime"].InferenceSession = _FakeSession
    r_bg.rt.InferenceSession = _FakeSession
    r_anp.rt.InferenceSession = _FakeSession
    tf = Transform()
    g = r_bg.DeepGroove(s, _FakeCanal(canal_axis_ct), tf)
    g.points()
    out = dict(canal_axis_ct=canal_axis_ct, T_obb=s.obb.transform, zs=s._zs,
               X=g._X, peak_theta=g._peak_theta, bg_theta=np.float64(g.bg_theta),
               points_obb=g._points_obb, points_ct=g._points_ct)
    np.savez_compressed(os.path.join(HERE, f"groove_golden_{tag}.npz"), **out)
    if not with_anp:
        return
    # AnatomicNeck.points(): needs bcptl.axis() (skspatial Line.best_fit) only to force the
    # groove; bg_theta is already set, so axis() is replaced by a no-op on this instance.
    g.axis = lambda: None
    import importlib.resources as ir
    real_files = ir.files

    def files(pkg):
        class _P:
            def __truediv__(self, other):
                return os.path.join("/root/reference/src/shoulder/humerus/models", "rfc_bg3.onnx")
        return _P()

    r_anp.importlib.resources.files = files
    try:
        a = r_anp.AnatomicNeck(s, g, tf)
        a.points()
    finally:
        r_anp.importlib.resources.files = real_files
    np.savez_compressed(os.path.join(HERE, f"anp_golden_{tag}.npz"),
                        image_f32=_FakeSession.captured["image"][0, 0], logits_f32=_FakeSession.captured["logits"],
                        bg_theta=np.float64(g.bg_theta), points_obb=a._points_obb,
                        n_articular=np.int64(len(a._points_all_articular_obb)), points_ct=a._points_ct)


def synthetic_contours(n=600, m=512, seed=99):
    """Analytic closed contours with two notches (one deeper), arclength-resampled by the
    reference's own _resample_polygon; regenerated identically by the tests."""
    rng = np.random.default_rng(seed)
    zs = np.linspace(170.0, 128.0, n)
    s = r_slice.FullSlices.__new__(r_slice.FullSlices)
    ixy = np.zeros((n, 2, m))
    cents = np.zeros((n, 2))
    ph = rng.uniform(-0.1, 0.1, n)
    for i in range(n):
        t = np.linspace(-np.pi, np.pi, 400)
        f = i / (n - 1)
        r = 22 - 6 * f + 1.5 * np.cos(2 * t + ph[i]) - 2.2 * np.exp(-((t - 0.9) / 0.12) ** 2) \
            - 1.1 * np.exp(-((t + 1.7) / 0.15) ** 2)
        xy = np.c_[3 + r * np.cos(t), -2 + r * np.sin(t)]
        xy[-1] = xy[0]
        ixy[i] = s._resample_polygon(xy, m).T
        cents[i] = 0.5 * (xy.min(axis=0) + xy.max(axis=0))
    return zs, ixy, cents


def golden_metrics():
    """bone_props.RadiusCurvature._spherefit :126-148 and utils.unitxyz_to_spherical :321-332."""
    from shoulder.humerus import bone_props
    rng = np.random.default_rng(31)
    d = rng.standard_normal((5000, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    d = d[d[:, 0] + 0.3 * d[:, 2] > 0.2]                      # a spherical cap
    pts = np.array([4.0, -13.0, 147.0]) + 23.5 * d + rng.normal(0, 0.15, d.shape)
    rc = bone_props.RadiusCurvature.__new__(bone_props.RadiusCurvature)
    radius, center = rc._spherefit(pts)
    v = rng.standard_normal((6, 3))
    v /= np.linalg.norm(v, axis=1, keepdims=True)
    sph = np.array([rutils.unitxyz_to_spherical(x) for x in v])
    np.savez_compressed(os.path.join(HERE, "metrics_golden.npz"), pts=pts, radius=np.float64(radius), center=center, unit_vecs=v, spherical=sph)


if __name__ == "__main__":
    golden_utils()
    golden_metrics()
    # realistic contours: oracle's proximal slices of humerus_left, rounded to float32
    from oracle.humerus import OracleHumerus
    spec_path = os.path.join(ROOT, "shoulder_amd", "unet_spec.py")
    import importlib.util
    sp = importlib.util.spec_from_file_location("unet_spec", spec_path)
    us = importlib.util.module_from_spec(sp)
    sp.loader.exec_module(us)
    _FakeSession.tables = o_rfc.load_tables(os.path.join(ROOT, "shoulder_amd", "models", "rfc_bg3.npz"))
    _FakeSession.unet_w = us.make_teacher_weights()
    h = OracleHumerus.from_stl(os.path.join(HERE, "bones", "humerus_left.stl"), _FakeSession.tables)
    px = h.proximal
    ixy = px.ixy_all.astype(np.float32).astype(np.float64)
    cents = px.centroids_all.astype(np.float32).astype(np.float64)
    zs = px.zs_all.astype(np.float32).astype(np.float64)
    T = h.T_obb
    np.savez_compressed(os.path.join(HERE, "contours_left.npz"), ixy=ixy.astype(np.float32),
                        centroids=cents.astype(np.float32), zs=zs.astype(np.float32), T_obb=T,
                        canal_axis_ct=h.canal["axis_ct"])
    s = golden_slices(ixy, cents, zs, T)
    golden_groove_anp(s, h.canal["axis_ct"], "left", with_anp=True)
    zs2, ixy2, cents2 = synthetic_contours()
    T2 = np.identity(4)
    s2 = _FakeSlices(zs2, ixy2, cents2, T2)
    golden_groove_anp(s2, np.array([[1.0, -2.0, 40.0], [-1.0, 2.0, -40.0]]), "synth", with_anp=False)
    print("golden vectors written to", HERE)
