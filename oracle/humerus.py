"""The whole hot path, CPU (oracle; test infrastructure).

Restates `shoulder.Humerus.__init__` + `apply_csys_canal_transepiconylar`
(reference `src/shoulder/bone.py:110-157`) wiring the per-stage restatements of this
package in the order the reference evaluates them.
"""
import numpy as np

from . import anp, canal, cpd, groove, obb, slices, te, unet
from .section import ZSlicer, ring_area
from .slices import cutoff_range
from .stl import load_stl
from .xform import construct_csys, inv_transform, transform_pts


class OracleHumerus:
    def __init__(self, verts, faces, rfc_tables, unet_weights=None, unet_eval="f64"):
        self.verts = np.asarray(verts, dtype=np.float32)
        self.faces = np.asarray(faces, dtype=np.int32)
        self.rfc = rfc_tables
        self.unet_w = unet_weights
        self.unet_eval = unet_eval
        self._c = {}

    @classmethod
    def from_stl(cls, path, *a, **k):
        v, f = load_stl(path)
        return cls(v, f, *a, **k)

    def _memo(self, key, fn):
        if key not in self._c:
            self._c[key] = fn()
        return self._c[key]

    # mesh.py:63-125 ------------------------------------------------------------------
    @property
    def obb(self):
        return self._memo("obb", lambda: obb.full_obb(self.verts.astype(np.float64), self.faces))

    @property
    def T_obb(self):
        return self.obb["transform"]

    @property
    def verts_obb(self):
        return self.obb["verts_obb"]

    def _bounds_z(self):
        z = self.verts_obb[:, 2]
        return float(z.max()), float(z.min())

    # slice.py:209-276 ----------------------------------------------------------------
    @property
    def full(self):
        zmax, zmin = self._bounds_z()
        return self._memo("full", lambda: slices.Slices(self.verts_obb, self.faces, slices.full_zs(zmax, zmin), 100))

    @property
    def distal(self):
        _, zmin = self._bounds_z()
        return self._memo("distal", lambda: slices.Slices(self.verts_obb, self.faces, slices.distal_zs(zmin), 500))

    @property
    def proximal(self):
        zmax, _ = self._bounds_z()
        return self._memo("prox", lambda: slices.Slices(self.verts_obb, self.faces,
                                                        slices.proximal_zs(zmax, self.neck["neck_z"]), 512))

    # surgical_neck.py:22-56 ------------------------------------------------------------
    @property
    def neck(self):
        def f():
            cutoff = (0.70, 0.99)
            areas = self.full.cut(self.full.areas1_all, cutoff)
            bkp = cpd.kernel_cpd_one_bkp(areas)
            neck_z = float(self.full.zs(cutoff)[bkp])
            rings = ZSlicer(self.verts_obb, self.faces).loops(neck_z)
            if len(rings) > 1:   # :40-48 loop whose vertex mean is nearest the origin (L1)
                ring = rings[int(np.argmin([np.sum(np.abs(np.mean(r[:, :2], axis=0))) for r in rings]))]
            else:
                ring = rings[0]
            pts_obb = np.c_[ring, np.full(len(ring), neck_z)]
            return dict(neck_z=neck_z, bkp=int(bkp), areas=areas, points_obb=pts_obb,
                        points_ct=transform_pts(pts_obb, inv_transform(self.T_obb)))
        return self._memo("neck", f)

    # canal.py ------------------------------------------------------------------------
    @property
    def canal(self):
        def f():
            p_obb, p_ct = canal.canal_points(self.full.centroids_all, self.full.zs_all, self.T_obb)
            a_obb, a_ct = canal.canal_axis(p_obb, self.obb["z_length"], self.T_obb)
            return dict(points_obb=p_obb, points_ct=p_ct, axis_obb=a_obb, axis_ct=a_ct)
        return self._memo("canal", f)

    # bicipital_groove.py ---------------------------------------------------------------
    @property
    def groove(self):
        def f():
            cutoff = (0.2, 0.75)
            px = self.proximal
            g = groove.groove_points(px.cut(px.itr_centered_start_all, cutoff), px.zs(cutoff),
                                     px.cut(px.centroids_all, cutoff), self.canal["axis_ct"], self.T_obb, self.rfc)
            g["axis_ct"] = groove.groove_axis(g["points_obb"], self.T_obb)
            return g
        return self._memo("groove", f)

    # anatomic_neck.py ------------------------------------------------------------------
    @property
    def anp_input(self):
        def f():
            cutoff = (0.0, 0.852)
            px = self.proximal
            img, shft, roll = anp.anp_image(px.cut(px.itr_start_all, cutoff), self.groove["bg_theta"])
            return dict(image=img, itr_shft=shft, roll=roll, zs=px.zs(cutoff))
        return self._memo("anp_in", f)

    def logits(self):
        def f():
            if self.unet_w is None:
                raise ValueError("no UNet weights supplied")
            img32 = self.anp_input["image"].astype(np.float32)
            fwd = unet.forward_chain if self.unet_eval == "chain" else unet.forward_f64
            return fwd(self.unet_w, img32)
        return self._memo("logits", f)

    @property
    def anp(self):
        def f():
            a = self.anp_input
            m = anp.mask_points(self.logits(), a["itr_shft"], a["zs"])
            point, normal = anp.neck_plane(m["points_obb"])
            T_inv = inv_transform(self.T_obb)
            an = anp.axis_normal(self.verts_obb, self.faces, point, normal)
            ac = anp.axis_central(self.verts_obb, self.faces, point, normal)
            m.update(points_ct=transform_pts(m["points_obb"], T_inv), plane_point_obb=point, plane_normal_obb=normal,
                     plane_point_ct=transform_pts(point.reshape(1, 3), T_inv)[0],
                     plane_normal_ct=np.matmul(T_inv[:3, :3], normal),
                     axis_normal_ct=transform_pts(an, T_inv), axis_central_ct=transform_pts(ac, T_inv))
            return m
        return self._memo("anp", f)

    def anp_plane_points(self):
        """AnatomicNeck.plane_points() in CT (anatomic_neck.py:155-172)."""
        return self._memo("anp_plane_points", lambda: anp.plane_points(self.verts.astype(np.float64), self.faces,
                                                                      self.anp["plane_point_ct"], self.anp["plane_normal_ct"]))

    # epicondyle.py ---------------------------------------------------------------------
    @property
    def te(self):
        return self._memo("te", lambda: te.te_axis(self.distal.largest, self.distal.zs_all, self.T_obb,
                                                  self.canal["axis_ct"], self.anp["axis_central_ct"]))

    # bone.py:146-157 -------------------------------------------------------------------
    def csys_canal_transepicondylar(self):
        return construct_csys(self.canal["axis_ct"], self.te["axis_ct"])

    def csys_canal_articular(self):
        """bone.py:53-62"""
        return construct_csys(self.canal["axis_ct"], self.anp["axis_normal_ct"])

    # bone_props.py (bone.py:134-144) -----------------------------------------------------
    def metrics(self, axis_normal_current=None):
        """side / retroversion / neckshaft / radius_curvature.  `axis_normal_current` = AnatomicNeck.axis_normal()
        in whatever coordinate system is applied when retroversion() is called (reference quirk,
        bone_props.py:72-73); default: CT (identity Transform)."""
        from . import metrics as m
        an = self.anp["axis_normal_ct"] if axis_normal_current is None else axis_normal_current
        side = m.side(self.canal["axis_ct"], self.anp["axis_central_ct"], self.groove["points_ct"])
        return dict(side=side, retroversion=m.retroversion(self.canal["axis_ct"], self.te["axis_ct"], an, side),
                    neckshaft=m.neckshaft(self.canal["axis_ct"], self.anp["axis_normal_ct"]),
                    radius_curvature=m.spherefit(self.anp["articular_obb"])[0])

    def landmarks(self, with_unet=True):
        """Everything in CT coordinates + the canal/TE coordinate system."""
        out = dict(T_obb=self.T_obb, z_length=self.obb["z_length"], neck_z=self.neck["neck_z"],
                   canal_axis=self.canal["axis_ct"], canal_points=self.canal["points_ct"],
                   groove_points=self.groove["points_ct"], groove_axis=self.groove["axis_ct"],
                   bg_theta=self.groove["bg_theta"], groove_local_idx=self.groove["local_idx"])
        if with_unet:
            out.update(anp_points=self.anp["points_ct"], anp_plane_point=self.anp["plane_point_ct"],
                       anp_plane_normal=self.anp["plane_normal_ct"], anp_axis_normal=self.anp["axis_normal_ct"],
                       anp_axis_central=self.anp["axis_central_ct"], te_axis=self.te["axis_ct"],
                       csys=self.csys_canal_transepicondylar(), csys_articular=self.csys_canal_articular())
        return out
