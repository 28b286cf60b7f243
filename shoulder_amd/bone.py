"""`shoulder.Humerus` drop-in facade (reference src/shoulder/bone.py:109-157 and the accessor API
of humerus/{canal,bicipital_groove,anatomic_neck,epicondyle,surgical_neck}.py).

Same names (incl. `trans_epiconylar`, `apply_csys_canal_transepiconylar`), argument meaning,
return shapes, float64 NumPy arrays, lazy + memoised evaluation, landmarks cached in CT and
re-expressed through one shared `Transform`.  Every number comes from libshoulder_hip.so through
`Engine` (one sh_run over a batch of one); nothing here falls back to the CPU.
"""
import os
import pathlib
import threading
import warnings

import numpy as np

from . import _lib, unet_spec
from .base import Bone, Landmark, Mesh, Plane, Transform
from .engine import Engine
from .stl import load_stl

_DEFAULT_ENGINE = None
_DEFAULT_LOCK = threading.Lock()

EARLY = _lib.STAGE_OBB | _lib.STAGE_FULL | _lib.STAGE_NECK | _lib.STAGE_CANAL
LATE = _lib.STAGE_PROXIMAL | _lib.STAGE_GROOVE | _lib.STAGE_ANP | _lib.STAGE_DISTAL | _lib.STAGE_TE | _lib.STAGE_CSYS | _lib.STAGE_APPLY


def default_engine(device=0, unet_weights=None, unet_onnx=None):
    """Process-wide engine on `device` (created once, under a lock; an Engine itself is single-threaded like the
    reference's objects -- give threads their own engines).

    The anatomic-neck network: `unet_onnx` (or the environment variable SHOULDER_UNET_ONNX) names an ONNX file of the
    supported UNet family -- the reference opens its packaged `humerus/models/unetcrf_anp.onnx` (anatomic_neck.py:62-66),
    which is not part of this tree; `unet_weights` takes a parameter dict.  With neither, the seeded TEACHER weights of
    `unet_spec` are loaded and a RuntimeWarning says so: they make every stage computable and testable, but anatomic-neck
    dependent values (anatomic_neck.*, neckshaft, retroversion, radius_curvature, side, apply_csys_canal_articular) are then
    NOT those of a trained model."""
    global _DEFAULT_ENGINE
    with _DEFAULT_LOCK:
        if _DEFAULT_ENGINE is None:
            e = Engine(device)
            e.load_rfc()
            onnx_path = unet_onnx or os.environ.get("SHOULDER_UNET_ONNX")
            if onnx_path:
                e.load_unet_onnx(onnx_path)
            elif unet_weights is not None:
                e.load_unet(unet_weights, unet_spec.BASE, unet_spec.DEPTH)
            else:
                warnings.warn("shoulder_amd: no anatomic-neck model given (SHOULDER_UNET_ONNX / default_engine(unet_onnx=...)): loading the "
                              "seeded TEACHER stand-in -- anatomic-neck dependent landmarks and metrics are not those of a trained model",
                              RuntimeWarning, stacklevel=3)
                e.load_unet(unet_spec.make_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
            _DEFAULT_ENGINE = e
    return _DEFAULT_ENGINE


def _inv(T):
    # utils.inv_transform (utils.py:227-256) on the host for a single 4x4 (glue; the device
    # computes the same inside every stage that maps OBB -> CT)
    R = np.identity(4)
    R[:3, :3] = T[:3, :3]
    t = np.identity(4)
    t[:3, 3] = T[:3, 3]
    return np.linalg.inv(R) @ np.linalg.inv(t)


def _scatter(pts, name, **kw):
    """One plotly trace per landmark, as the reference's `_graph_obj` methods build them (e.g. canal.py:132-142)."""
    import plotly.graph_objects as go
    return go.Scatter3d(x=pts[:, 0], y=pts[:, 1], z=pts[:, 2], name=name, **kw)


class _Lm(Landmark):
    def __init__(self, bone):
        self._b = bone
        self._tfrm = bone._tfrm

    def _t(self, pts_ct):
        return self._b._engine.transform_points(pts_ct, self._tfrm.matrix)


class Canal(_Lm):
    """canal.py"""

    def __init__(self, bone):
        super().__init__(bone)
        self._points_ct = None
        self._axis_ct = None

    def points(self, cutoff_pcts=(0.35, 0.75)) -> np.ndarray:
        if self._points_ct is None:
            # DeepGroove.__init__ calls canal.axis() in the reference (bicipital_groove.py:21), so the
            # default cutoff is already fixed when user code first gets here (canal.py:60-62)
            e = self._b._engine
            self._b._ensure_loaded()
            c0, c1 = self._b._canal_cutoff
            n = int((1 - c0) * 200) - int((1 - c1) * 200)                 # Slices._cutoff (slice.py:157-164) on the 200 full slices
            obb = e.fetch("canal.points_obb", np.float64, (1, 200, 3))[0][:n]
            self._points_obb = obb
            self._points_ct = e.transform_points(obb, _inv(self._b._obb_transform))
        self._points = self._t(self._points_ct)
        return self._points

    def axis(self, cutoff_pcts=(0.35, 0.75)) -> np.ndarray:
        if self._axis_ct is None:
            self._axis_ct = np.array(self._b._early["canal_axis"], dtype=np.float64)
        self._axis = self._t(self._axis_ct)
        return self._axis

    def get_transform(self) -> np.ndarray:
        """canal.py:88-124 (CT -> canal csys, x from the OBB frame)."""
        ax = self.axis()
        z_hat = (ax[0] - ax[1]) / np.linalg.norm(ax[0] - ax[1])
        x_hat = self._b._obb_transform[:3, :1].flatten().copy()
        x_hat -= z_hat * np.dot(x_hat, z_hat) / np.dot(z_hat, z_hat)
        x_hat /= np.linalg.norm(x_hat)
        y_hat = np.cross(z_hat, x_hat)
        y_hat /= np.linalg.norm(y_hat)
        T = np.r_[np.c_[x_hat, y_hat, z_hat, np.average(ax, axis=0)], np.array([[0, 0, 0, 1.0]])]
        return _inv(T)

    def transform_landmark(self) -> None:
        if self._axis_ct is not None:
            self.axis()
        if self._points_ct is not None:
            self.points()

    def _graph_obj(self):                                  # canal.py:132-142
        return None if self._points_ct is None else _scatter(self._points, "Canal Axis")


class SurgicalNeck(_Lm):
    """surgical_neck.py"""

    def __init__(self, bone):
        super().__init__(bone)
        e = bone._engine
        self.neck_z = float(bone._early["neck_z"])
        ring = e.ring("neckc", 0, 0)
        pts_obb = np.c_[ring, np.full(len(ring), self.neck_z)]
        self.points_ct = e.transform_points(pts_obb, _inv(bone._obb_transform))
        self.points = self.points_ct.copy()

    def cutoff_zs(self, bottom_pct=0.35, top_pct=0.85):
        z_max = self._b._z_bounds[1]
        span = z_max - self.neck_z
        return [self.neck_z + span * bottom_pct, self.neck_z + span * top_pct]

    def z_percent(self):
        z_min, z_max = self._b._z_bounds
        return (self.neck_z - z_min) / (abs(z_max) + abs(z_min))

    def transform_landmark(self) -> None:
        if self.points is not None:
            self.points = self._t(self.points_ct)

    def _graph_obj(self):                                  # surgical_neck.py:82-93
        return None if self.points is None else _scatter(self.points, "Surgical Neck")


class DeepGroove(_Lm):
    """bicipital_groove.py"""

    def __init__(self, bone):
        super().__init__(bone)
        self._points_ct = None
        self._axis_ct = None

    def points(self, cutoff_pcts=(0.2, 0.75), deg_window=7) -> np.ndarray:
        if self._points_ct is None:
            lm = self._b._all(groove_cutoff=cutoff_pcts, deg_window=deg_window)
            self.bg_theta = float(lm["bg_theta"])
            self._points_ct = np.array(lm["groove_points"], dtype=np.float64)
        self._points = self._t(self._points_ct)
        return self._points

    def axis(self) -> np.ndarray:
        if self._axis_ct is None:
            if self._points_ct is None:
                self.points()
            self._axis_ct = np.array(self._b._all()["groove_axis"], dtype=np.float64)
        self._axis = self._t(self._axis_ct)
        return self._axis

    def transform_landmark(self) -> None:
        if self._axis_ct is not None:
            self.axis()
        if self._points_ct is not None:
            self.points()

    def _graph_obj(self):                                  # bicipital_groove.py:273-284
        return None if self._points_ct is None else _scatter(self._points, "Bicipital Groove")


class AnatomicNeck(_Lm):
    """anatomic_neck.py"""

    def __init__(self, bone):
        super().__init__(bone)
        self._points_ct = self._plane_ct = self._plane_points_ct = None
        self._central_axis_ct = self._normal_axis_ct = None

    def points(self) -> np.ndarray:
        if self._points_ct is None:
            self._b.bicipital_groove.axis()               # anatomic_neck.py:47 forces the groove
            lm = self._b._all()
            k = int(lm["n_anp"])
            if k > _lib.ANP_MAX_PTS:                       # the record is padded to 4096 rows; fetch the rest
                self._b._ensure_loaded()
                obb =self._b._engine.fetch("anp.points_obb", np.float64, (65536, 3))[:k]
                self._points_ct = self._b._engine.transform_points(obb, _inv(self._b._obb_transform))
            else:
                self._points_ct = np.array(lm["anp_points"][:k], dtype=np.float64)
        self._points = self._t(self._points_ct)
        return self._points

    def plane(self) -> Plane:
        if self._plane_ct is None:
            self.points()
            lm = self._b._all()
            self._plane_ct = Plane(lm["anp_plane_point"], lm["anp_plane_normal"])
        T = self._tfrm.matrix
        self._plane = Plane(self._t(self._plane_ct.point.reshape(1, 3))[0], T[:3, :3] @ self._plane_ct.normal)   # utils.py:191-206
        return self._plane

    def plane_points(self) -> np.ndarray:
        if self._plane_points_ct is None:
            self.plane()
            self._b._ensure_loaded()
            self._plane_points_ct = self._b._engine.section_plane(0, self._plane_ct.point, self._plane_ct.normal)
        self._plane_points = self._t(self._plane_points_ct)
        return self._plane_points

    def axis_normal(self) -> np.ndarray:
        if self._normal_axis_ct is None:
            self.plane()
            self._normal_axis_ct = np.array(self._b._all()["anp_axis_normal"], dtype=np.float64)
        self._normal_axis = self._t(self._normal_axis_ct)
        return self._normal_axis

    def axis_central(self) -> np.ndarray:
        if self._central_axis_ct is None:
            self.plane()
            self._central_axis_ct = np.array(self._b._all()["anp_axis_central"], dtype=np.float64)
        self._central_axis = self._t(self._central_axis_ct)
        return self._central_axis

    def transform_landmark(self) -> None:
        if self._points_ct is not None:
            self.points()
        if self._plane_ct is not None:
            self.plane()
        if self._plane_points_ct is not None:
            self.plane_points()
        if self._normal_axis_ct is not None:
            self.axis_normal()
        if self._central_axis_ct is not None:
            self.axis_central()

    def _graph_obj(self):                                  # anatomic_neck.py:250-273
        if self._points_ct is None:
            return None
        pp = self.plane_points()
        return [_scatter(self._points, "Anatomic Neck", mode="markers", showlegend=True),
                _scatter(pp, "Anatomic Neck Plane", mode="markers", showlegend=True)]


class TransEpicondylar(_Lm):
    """epicondyle.py"""

    def __init__(self, bone):
        super().__init__(bone)
        self._axis_ct = None

    def axis(self, num_slices: int = 50) -> np.ndarray:       # num_slices is ignored by the reference too
        if self._axis_ct is None:
            self._b.anatomic_neck.axis_central()              # epicondyle.py:90 forces groove + neck
            self._axis_ct = np.array(self._b._all()["te_axis"], dtype=np.float64)
        self._axis = self._t(self._axis_ct)
        return self._axis

    def transform_landmark(self) -> None:
        if self._axis_ct is not None:
            self.axis()

    def _graph_obj(self):                                  # epicondyle.py:107-117
        return None if self._axis_ct is None else _scatter(self._axis, "Transverse Epicondylar Axis")


class ProximalHumerus(Bone):
    """bone.py:24-105 -- a humerus whose scan ends in the shaft (and, as in the reference, the base class of `Humerus`:
    `isinstance(Humerus(...), ProximalHumerus)` holds).  The device runs the ProxObb head-end rule and canal range
    (mesh.py:128-192), the (0.2, 0.99) neck cut-off (surgical_neck.py:25-26) and the canal cut-offs taken from the box
    (canal.py:33-38); there is no trans-epicondylar axis and no retroversion, and the coordinate system is
    `apply_csys_canal_articular` (bone.py:53-62)."""
    _BONE_KIND = _lib.BONE_PROXIMAL
    _EARLY = EARLY
    _LATE = _lib.STAGE_PROXIMAL | _lib.STAGE_GROOVE | _lib.STAGE_ANP | _lib.STAGE_CSYS | _lib.STAGE_APPLY

    def __init__(self, stl_file, engine=None):
        self._tfrm = Transform()
        self.transform = self._tfrm.matrix
        self.stl_file = stl_file if isinstance(stl_file, pathlib.Path) else pathlib.Path(stl_file)
        self._engine = engine if engine is not None else default_engine()
        verts, faces = load_stl(self.stl_file)
        self._verts, self._faces = verts, faces
        self._engine.upload([(verts, faces)])
        self._engine.set_params(bone_kind=self._BONE_KIND)      # (read-modify-write: the engine's UNet dtype and cut-offs stay as configured)
        self._lm_all = None
        self._groove_params = None      # (cutoff_pcts, deg_window) this bone's late stages ran with
        # eager part of the reference constructor: OBB, full slices, surgical neck, canal axis
        self._early = self._engine.run(self._EARLY)[0].copy()
        self._canal_cutoff = tuple(float(x) for x in self._early["canal_cutoff"])
        self._obb_transform = np.array(self._early["obb_transform"], dtype=np.float64)
        self._z_bounds = tuple(self._engine.fetch("z_bounds", np.float64, (1, 2))[0])
        self._mesh_ct = Mesh(verts, faces, self._engine)
        self.mesh = self._mesh_ct.copy()
        self.surgical_neck = SurgicalNeck(self)
        self.canal = Canal(self)
        self.bicipital_groove = DeepGroove(self)
        self.anatomic_neck = AnatomicNeck(self)
        self._init_kind_specific()

    def _init_kind_specific(self):
        # metrics (bone.py:46-51 -> bone_props.py); values come from the device record (k_metrics)
        self.side = self._side
        self.neckshaft = self._neckshaft
        self.radius_curvature = self._radius_curvature
        e = self._engine
        self.cutoff_pcts = [float(x) for x in self._canal_cutoff]           # ProxObb.cutoff_pcts (mesh.py:190)
        self.cutoff_bot = int(e.fetch("pobb.cutoff_idx", np.int32, (1, 2))[0][0])      # mesh.py:187

    def _ensure_loaded(self):
        """The shared engine may have been used for another bone since: bring this one back."""
        e = self._engine
        if e.B != 1 or not np.array_equal(e.fetch("obb_transform", np.float64, (1, 4, 4))[0], self._obb_transform):
            e.upload([(self._verts, self._faces)])
            e.set_params(bone_kind=self._BONE_KIND, canal_cutoff=self._canal_cutoff if self._BONE_KIND == _lib.BONE_HUMERUS else None)
            e.run(self._EARLY, fetch=False)
            if self._lm_all is not None:      # device buffers fetched later (e.g. anp.points_obb beyond 4096 rows) must match the cached record
                e.set_params(groove_cutoff=self._groove_params[0], groove_deg_window=self._groove_params[1])
                e.run(self._LATE, fetch=False)

    # -- the late stages, once ------------------------------------------------------------------------
    def _all(self, groove_cutoff=(0.2, 0.75), deg_window=7):
        if self._lm_all is None:
            e = self._engine
            self._ensure_loaded()
            try:
                e.set_params(groove_cutoff=tuple(groove_cutoff), groove_deg_window=float(deg_window), bone_kind=self._BONE_KIND)
            except Exception as ex:
                raise ValueError(f"bicipital_groove cutoff_pcts {groove_cutoff} is not supported: {ex}") from ex
            self._groove_params = (tuple(float(x) for x in groove_cutoff), float(deg_window))
            self._lm_all = e.run(self._LATE)[0].copy()
            if int(self._lm_all["status"]) != 0:
                raise ValueError(f"landmark stage failed with status {int(self._lm_all['status'])}")
        return self._lm_all

    # -- metrics (bone_props.py) ---------------------------------------------------------------------------
    def _side(self) -> str:
        """bone_props.py:23-47"""
        self.canal.axis(); self.anatomic_neck.axis_central(); self.bicipital_groove.points()
        return "right" if int(self._all()["side"]) == 1 else "left"

    def _neckshaft(self) -> float:
        """bone_props.py:93-112"""
        self.canal.axis(); self.anatomic_neck.axis_normal()
        return float(self._all()["neckshaft"])

    def _radius_curvature(self) -> float:
        """bone_props.py:119-125"""
        self.anatomic_neck.points()
        return float(self._all()["radius_curvature"])

    def _retroversion(self) -> float:
        """bone_props.py:64-85.  The reference transforms `axis_normal()` as returned in the CURRENT coordinate
        system (:72-73); with the identity Transform that is the device value, otherwise the same two-point
        arithmetic is redone here on the re-expressed axis (2 points, host glue)."""
        self.canal.axis(); self.trans_epiconylar.axis()
        an = self.anatomic_neck.axis_normal()
        lm = self._all()
        if np.array_equal(self._tfrm.matrix, np.identity(4)):
            return float(lm["retroversion"])
        q = self._engine.transform_points(an, np.array(lm["csys"], dtype=np.float64))
        v = (q[0] - q[1]) / np.linalg.norm(q[0] - q[1])
        theta = float(np.rad2deg(np.arctan2(v[1], -1 * v[0])))
        return -theta if int(lm["side"]) == 1 else theta

    # -- coordinate systems (bone.py:53-105, :146-157) --------------------------------------------------
    def _mesh_in(self, T):
        """`mesh_ct.copy().apply_transform(T)` (bone.py:155) on the device."""
        self._ensure_loaded()
        return Mesh(self._engine.mesh_transformed(0, T), self._faces, self._engine)

    def _apply(self, matrix, mesh):
        self._tfrm.matrix = matrix
        self._update_landmark_data()
        self.mesh = mesh
        self.transform = self._tfrm.matrix
        return self.transform

    def _mesh_in_record_csys(self, T):
        """The mesh in the record's own csys: the batch path's device buffer (SH_STAGE_APPLY, k_apply_csys) rather than a
        second transform pass, when this bone's late stages are what the engine holds."""
        e = self._engine
        self._ensure_loaded()
        V = len(self._verts)
        return Mesh(e.fetch("verts_csys", np.float64, (V, 3)), self._faces, e)

    def apply_csys_canal_articular(self) -> np.ndarray:
        self.canal.axis()
        self.anatomic_neck.axis_central()
        self.anatomic_neck.axis_normal()
        T = np.array(self._all()["csys_articular"], dtype=np.float64)     # construct_csys(canal, neck-normal axis) on the device (k_pack)
        mesh = self._mesh_in_record_csys(T) if self._BONE_KIND == _lib.BONE_PROXIMAL else self._mesh_in(T)
        return self._apply(T, mesh)

    def apply_csys_obb(self) -> np.ndarray:
        T = self._obb_transform.copy()
        return self._apply(T, self._mesh_in(T))

    def apply_csys_ct(self) -> np.ndarray:
        self._tfrm.reset()
        self._update_landmark_data()
        self.mesh = self._mesh_ct.copy()
        self.transform = self._tfrm.matrix
        return self.transform

    def apply_csys_custom(self, transform, from_ct=True) -> np.ndarray:
        if from_ct:
            self._tfrm.matrix = transform
            self._update_landmark_data()
            self.mesh = self._mesh_ct.copy().apply_transform(self._tfrm.matrix)
        else:   # reference quirk kept: the cumulative matrix is applied to the already moved mesh (bone.py:92-94)
            self._tfrm.matrix = np.dot(transform, self._tfrm.matrix)
            self._update_landmark_data()
            self.mesh = self.mesh.apply_transform(self._tfrm.matrix)
        self.transform = self._tfrm.matrix
        return self.transform

    def apply_translation(self, translation) -> np.ndarray:
        T = np.identity(4)
        T[:3, 3] = np.asarray(translation, dtype=np.float64).reshape(3)        # utils.py:259-264
        self._tfrm.matrix = np.dot(T, self._tfrm.matrix)
        self._update_landmark_data()
        self.mesh = self.mesh.apply_transform(self._tfrm.matrix)               # same quirk (bone.py:101-103)
        self.transform = self._tfrm.matrix
        return self.transform


# we are inheriting the functions but the init will be unique (bone.py:108-109)
class Humerus(ProximalHumerus):
    """bone.py:109-157"""
    _BONE_KIND = _lib.BONE_HUMERUS
    _LATE = LATE

    def _init_kind_specific(self):
        self.trans_epiconylar = TransEpicondylar(self)
        # metrics (bone.py:134-144 -> bone_props.py); values come from the device record (k_metrics)
        self.side = self._side
        self.retroversion = self._retroversion
        self.neckshaft = self._neckshaft
        self.radius_curvature = self._radius_curvature

    def apply_csys_canal_transepiconylar(self) -> np.ndarray:
        self.canal.axis()
        self.trans_epiconylar.axis()
        T = np.array(self._all()["csys"], dtype=np.float64)           # construct_csys on the device (k_pack)
        return self._apply(T, self._mesh_in_record_csys(T))
