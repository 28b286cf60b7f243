"""`HumeralHeadOsteotomy` -- resection of the humeral head at, or offset from, the anatomic-neck plane.

Same public surface as reference `src/shoulder/arthroplasty.py:13-175` (`plane`, `neckshaft_rel`, `retroversion_rel`,
`points()`, `resect_mesh()`, `offset_retroversion`, `offest_neckshaft` [sic], `offset_depth`, `offset_anterior_posterior`,
`offset_medial_lateral`), with its own bookkeeping: the resection plane is a (point, normal) pair of arrays kept in the
canal / articular coordinate system, where retroversion and neck-shaft angle are the two spherical angles of the normal;
every offset is one of two primitives (turn the normal by spherical-angle increments, shift the point), and `plane` maps
the pair into whatever coordinate system the humerus currently has.  The two mesh operations run on the device through
`sh_slice_mesh_planes` (k_clip.h): a resection sends the plane and its mirror image in one pass.

Results are pinned against the reference's own class (tests/golden/make_osteotomy_golden.py, tests/test_osteotomy_golden.py),
including its quirk that reading `retroversion_rel` flips the sign of the stored normal's x component.
"""
from typing import Tuple

import numpy as np

from .base import Mesh, Plane
from .csys import inv_transform, spherical_to_unitxyz, transform_plane_pn, unitxyz_to_spherical

_DEPTH_DIRECTIONS = ("canal", "anp", "resection")


class HumeralHeadOsteotomy:
    def __init__(self, humerus) -> None:
        self._humerus = humerus
        caller_csys = np.array(humerus._tfrm.matrix, dtype=np.float64)
        # visit the canal / articular csys once to read the anatomic-neck plane there, then hand the caller's csys back
        # (through CT: every matrix of the facade is CT-based) -- arthroplasty.py:18-31
        humerus.apply_csys_canal_articular()
        self._to_anp = np.array(humerus._tfrm.matrix, dtype=np.float64)
        native = humerus.anatomic_neck.plane()
        self._native_point = np.array(native.point, dtype=np.float64)
        self._native_normal = np.array(native.normal, dtype=np.float64)
        self._point = self._native_point.copy()
        self._normal = self._native_normal.copy()
        humerus.apply_csys_ct()
        humerus.apply_csys_custom(caller_csys)

    # ---- state ---------------------------------------------------------------------------------------
    def _side_sign(self) -> float:
        """+1 for a right humerus, -1 for a left one: the sense in which retroversion / anterior count (arthroplasty.py:97-100, :155-158)."""
        return 1.0 if self._humerus.side() == "right" else -1.0

    def _turn(self, d_theta=0.0, d_phi=0.0) -> None:
        r, theta, phi = unitxyz_to_spherical(self._normal)
        self._normal = spherical_to_unitxyz(np.array([r, theta + d_theta, phi + d_phi]))

    def _shift(self, vec) -> None:
        self._point = self._point + np.asarray(vec, dtype=np.float64)

    @property
    def plane(self) -> Plane:
        """the resection plane in the humerus's current csys (arthroplasty.py:33-40): anp csys -> CT -> current"""
        p, n = transform_plane_pn(self._point, self._normal, inv_transform(self._to_anp))
        return Plane(*transform_plane_pn(p, n, self._humerus._tfrm.matrix))

    @property
    def neckshaft_rel(self):
        """neck-shaft angle of the cut minus the native one, degrees (arthroplasty.py:42-54)"""
        phi_cut = unitxyz_to_spherical(self._normal)[2]
        phi_native = unitxyz_to_spherical(self._native_normal)[2]
        return (180 - phi_cut) - (180 - phi_native)

    @property
    def retroversion_rel(self):
        """retroversion of the cut, degrees, measured from -x (arthroplasty.py:56-67).  As in the reference the mirrored x
        stays in the stored normal, so two consecutive reads differ in sign and the second restores the state."""
        self._normal[0] = -self._normal[0]
        theta = unitxyz_to_spherical(self._normal)[1]
        return theta * (-1.0 if self._humerus.side() == "right" else 1.0)

    # ---- mesh operations (device) -----------------------------------------------------------------------
    def points(self):
        """closed polyline where the resection plane meets the bone: the loop of largest area (arthroplasty.py:69-78)"""
        cut = self.plane
        section = self._humerus.mesh.section(cut.normal, cut.point)
        if not section.discrete:
            raise ValueError("the resection plane does not cut the mesh")
        areas = [poly.area for poly in section.polygons_closed]
        return section.discrete[int(np.argmax(areas)) if len(section.entities) > 1 else 0]

    def resect_mesh(self) -> Tuple[Mesh, Mesh]:
        """(head, resected humerus) in the current csys (arthroplasty.py:80-87)"""
        cut = self.plane
        bone = self._humerus.mesh
        if bone._engine is None:
            raise RuntimeError("resect_mesh needs the HIP engine (no CPU fallback)")
        halves = bone._engine.slice_mesh_planes(bone.vertices, bone.faces, [cut.point, cut.point], [cut.normal, -cut.normal])
        return tuple(Mesh(v, f, bone._engine) for v, f in halves)

    # ---- offsets (arthroplasty.py:89-175) ------------------------------------------------------------------
    def offset_retroversion(self, deg: float) -> None:
        """more retroversion for positive `deg` (the azimuth decreases on a left humerus, increases on a right one)"""
        self._turn(d_theta=self._side_sign() * deg)

    def offest_neckshaft(self, deg: float) -> None:
        """larger neck-shaft angle for positive `deg` (the polar angle of the normal decreases)"""
        self._turn(d_phi=-deg)

    def offset_depth(self, mm, direction="canal") -> None:
        """move the plane by `mm` along the canal (z of the anp csys), the native neck normal, or the current cut normal"""
        if direction not in _DEPTH_DIRECTIONS:
            raise ValueError("Invalid direction. Choose from: 'canal', 'anp', or 'resection'")
        along = {"canal": np.array([0.0, 0.0, 1.0]), "anp": self._native_normal, "resection": self._normal}[direction]
        self._shift(mm * along)

    def offset_anterior_posterior(self, mm):
        """anterior (+) / posterior (-): x of the anp csys, mirrored for a left humerus"""
        self._shift([self._side_sign() * mm, 0.0, 0.0])

    def offset_medial_lateral(self, mm):
        """medial (+) / lateral (-): -y of the anp csys"""
        self._shift([0.0, -mm, 0.0])
