"""HumeralHeadOsteotomy plane bookkeeping (oracle; test infrastructure only).

Restates reference `src/shoulder/arthroplasty.py:13-175` on plain arrays: the resection plane lives in the canal /
articular csys (`_tfrm_anp`), is moved there by the offset_* methods, and is mapped into the caller's csys by
`plane()`.  The mesh operations go through oracle/clip.py.  The algebra is PINNED: tests/golden/osteotomy_golden.npz holds
the planes / angles the reference's own arthroplasty.py produced for a recorded script of operations on a stand-in humerus
(tests/golden/make_osteotomy_golden.py, third-party modules stubbed), and tests/test_osteotomy_golden.py replays it here.
"""
import numpy as np

from . import clip, xform
from .metrics import unitxyz_to_spherical


def spherical_to_unitxyz(s):                          # utils.py:333-339
    th, ph = np.deg2rad(s[1]), np.deg2rad(s[2])
    return np.array([s[0] * np.sin(ph) * np.cos(th), s[0] * np.sin(ph) * np.sin(th), s[0] * np.cos(ph)])


def transform_plane(point, normal, T):                # utils.py:191-206
    return xform.transform_pts(np.asarray(point, dtype=np.float64).reshape(1, 3), T)[0], T[:3, :3] @ np.asarray(normal, dtype=np.float64)


class OracleOsteotomy:
    def __init__(self, tfrm_anp, plane_point_ct, plane_normal_ct, side):
        """tfrm_anp: CT -> canal/articular matrix (bone.py:53-62); the anatomic-neck plane is given in CT."""
        self.tfrm_anp = np.array(tfrm_anp, dtype=np.float64)
        p, n = transform_plane(plane_point_ct, plane_normal_ct, self.tfrm_anp)
        self.anp_point, self.anp_normal = p.copy(), n.copy()
        self.res_point, self.res_normal = p.copy(), n.copy()
        self.side = side

    def plane(self, tfrm_current):                    # arthroplasty.py:33-40
        p, n = transform_plane(self.res_point, self.res_normal, xform.inv_transform(self.tfrm_anp))
        return transform_plane(p, n, np.asarray(tfrm_current, dtype=np.float64))

    def neckshaft_rel(self):                          # :42-54
        return (180 - unitxyz_to_spherical(self.res_normal)[2]) - (180 - unitxyz_to_spherical(self.anp_normal)[2])

    def retroversion_rel(self):                       # :56-67 (negates the stored x in place, as the reference does)
        self.res_normal[0] = -1 * self.res_normal[0]
        ret = unitxyz_to_spherical(self.res_normal)[1]
        return -ret if self.side == "right" else ret

    def offset_retroversion(self, deg):               # :90-104
        s = unitxyz_to_spherical(self.res_normal)
        s[1] += -deg if self.side == "left" else deg
        self.res_normal = spherical_to_unitxyz(s)

    def offest_neckshaft(self, deg):                  # :106-118
        s = unitxyz_to_spherical(self.res_normal)
        s[2] += -deg
        self.res_normal = spherical_to_unitxyz(s)

    def offset_depth(self, mm, direction="canal"):    # :120-145
        if direction == "canal":
            self.res_point[2] += mm
        elif direction == "anp":
            self.res_point = self.res_point + mm * self.anp_normal
        elif direction == "resection":
            self.res_point = self.res_point + mm * self.res_normal
        else:
            raise ValueError("Invalid direction. Choose from: 'canal', 'anp', or 'resection'")

    def offset_anterior_posterior(self, mm):          # :147-162
        self.res_point[0] += -mm if self.side == "left" else mm

    def offset_medial_lateral(self, mm):              # :164-175
        self.res_point[1] -= mm

    def points(self, verts_current, faces, tfrm_current):       # :69-78, loop start / direction canonicalised by the caller
        p, n = self.plane(tfrm_current)
        v, f, e = clip.slice_plane(verts_current, faces, p, n)
        loops = clip.loops_from_edges(e)
        un = n / np.linalg.norm(n)
        u = np.cross(un, [1.0, 0.0, 0.0] if abs(un[0]) < 0.9 else [0.0, 1.0, 0.0])
        u /= np.linalg.norm(u)
        w = np.cross(un, u)
        best, best_a = None, -1.0
        for lp in loops:
            d = v[lp + lp[:1]]
            x, y = d @ u, d @ w
            a = 0.5 * abs(np.sum(x[:-1] * y[1:] - x[1:] * y[:-1]))
            if a > best_a:
                best, best_a = d, a
        return best

    def resect(self, verts_current, faces, tfrm_current):       # :80-87
        p, n = self.plane(tfrm_current)
        return clip.slice_plane(verts_current, faces, p, n)[:2], clip.slice_plane(verts_current, faces, p, -n)[:2]
