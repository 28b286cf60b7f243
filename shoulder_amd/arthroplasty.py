"""`HumeralHeadOsteotomy` -- resection of the humeral head at / offset from the anatomic-neck plane.

Mirror of reference `src/shoulder/arthroplasty.py:13-175` (same method names, including the reference's spelling
`offest_neckshaft`, same argument meaning, same in-place quirks).  The plane bookkeeping is a handful of 3-vectors on
the host; the two mesh operations -- `points()` (section) and `resect_mesh()` (two `slice_plane` cuts) -- run on the
device through `sh_slice_mesh_planes` (k_clip.h), both halves of a resection in one pass.
"""
from typing import Tuple

import numpy as np

from .base import Mesh, Plane, Section
from .csys import inv_transform, spherical_to_unitxyz, transform_plane_pn, unitxyz_to_spherical


class HumeralHeadOsteotomy:
    """resects the humeral head at the anp or offset from the anp (arthroplasty.py:13-31)"""

    def __init__(self, humerus) -> None:
        self._humerus = humerus
        self._tfrm_og = self._humerus._tfrm.matrix.copy()
        # in the canal / articular csys the version and neck-shaft angles are spherical angles of the plane normal
        self._humerus.apply_csys_canal_articular()
        self._tfrm_anp = self._humerus._tfrm.matrix.copy()
        p = self._humerus.anatomic_neck.plane()
        self._anp_plane_csys_anp = Plane(p.point.copy(), p.normal.copy())
        self._res_plane_csys_anp = Plane(p.point.copy(), p.normal.copy())
        # back to the csys the caller had: through CT, every matrix is CT-based (arthroplasty.py:27-31)
        self._humerus.apply_csys_ct()
        self._humerus.apply_csys_custom(self._tfrm_og)

    @property
    def plane(self) -> Plane:
        """the resection plane in the current csys (arthroplasty.py:33-40)"""
        pt, n = transform_plane_pn(self._res_plane_csys_anp.point, self._res_plane_csys_anp.normal, inv_transform(self._tfrm_anp))
        pt, n = transform_plane_pn(pt, n, self._humerus._tfrm.matrix)
        return Plane(pt, n)

    @property
    def neckshaft_rel(self):
        """neck-shaft angle of the cut relative to native (arthroplasty.py:42-54)"""
        ns = 180 - unitxyz_to_spherical(self._res_plane_csys_anp.normal)[2]
        ns_og = 180 - unitxyz_to_spherical(self._anp_plane_csys_anp.normal)[2]
        return ns - ns_og

    @property
    def retroversion_rel(self):
        """retroversion of the cut relative to native (arthroplasty.py:56-67).  Reference quirk kept: the x component of the
        stored resection normal is negated IN PLACE on every read."""
        an = self._res_plane_csys_anp.normal
        an[0] = -1 * an[0]
        ret = unitxyz_to_spherical(an)[1]
        if self._humerus.side() == "right":
            ret *= -1
        return ret

    def points(self):
        """points of the largest closed polygon of the resection plane / mesh intersection (arthroplasty.py:69-78)"""
        pl = self.plane
        sec = self._humerus.mesh.section(pl.normal, pl.point)
        if not sec.discrete:
            raise ValueError("the resection plane does not cut the mesh")
        if len(sec.entities) > 1:
            return sec.discrete[int(np.argmax([p.area for p in sec.polygons_closed]))]
        return sec.discrete[0]

    def resect_mesh(self) -> Tuple[Mesh, Mesh]:
        """(head, resected humerus) in the current csys (arthroplasty.py:80-87); both cuts in one device pass"""
        pl = self.plane
        m = self._humerus.mesh
        if m._engine is None:
            raise RuntimeError("resect_mesh needs the HIP engine (no CPU fallback)")
        (hv, hf), (rv, rf) = m._engine.slice_mesh_planes(m.vertices, m.faces, [pl.point, pl.point], [pl.normal, -1 * pl.normal])
        return Mesh(hv, hf, m._engine), Mesh(rv, rf, m._engine)

    # ---- modify the plane (arthroplasty.py:89-175) ------------------------------------------------------
    def offset_retroversion(self, deg: float) -> None:
        sphr = unitxyz_to_spherical(self._res_plane_csys_anp.normal)
        sphr[1] += -1 * deg if self._humerus.side() == "left" else deg
        self._res_plane_csys_anp = Plane(self._res_plane_csys_anp.point, spherical_to_unitxyz(sphr))

    def offest_neckshaft(self, deg: float) -> None:
        sphr = unitxyz_to_spherical(self._res_plane_csys_anp.normal)
        sphr[2] += -1 * deg
        self._res_plane_csys_anp = Plane(self._res_plane_csys_anp.point, spherical_to_unitxyz(sphr))

    def offset_depth(self, mm, direction="canal") -> None:
        new_point = self._res_plane_csys_anp.point
        if direction == "canal":
            new_point[2] += mm
        elif direction == "anp":
            new_point += mm * np.array(self._anp_plane_csys_anp.normal)
        elif direction == "resection":
            new_point += mm * np.array(self._res_plane_csys_anp.normal)
        else:
            raise ValueError("Invalid direction. Choose from: 'canal', 'anp', or 'resection'")
        self._res_plane_csys_anp = Plane(new_point, self._res_plane_csys_anp.normal)

    def offset_anterior_posterior(self, mm):
        new_point = self._res_plane_csys_anp.point
        if self._humerus.side() == "left":
            new_point[0] -= mm
        else:
            new_point[0] += mm
        self._res_plane_csys_anp = Plane(new_point, self._res_plane_csys_anp.normal)

    def offset_medial_lateral(self, mm):
        new_point = self._res_plane_csys_anp.point
        new_point[1] -= mm
        self._res_plane_csys_anp = Plane(new_point, self._res_plane_csys_anp.normal)
