"""oracle.anp.plane_points (AnatomicNeck.plane_points, anatomic_neck.py:155-172) on CPU: against the z-plane slicer the
rest of the oracle uses (same canonical mesh_plane rules), and through invariants for oblique planes."""
import os

import numpy as np

from conftest import BONES
from oracle import anp
from oracle.section import ZSlicer
from oracle.stl import load_stl


def _rows(a):
    return a[np.lexsort(tuple(a[:, k] for k in reversed(range(a.shape[1]))))]


def test_z_plane_equals_zslicer():
    v, f = load_stl(os.path.join(BONES, "humerus_left.stl"))
    v = v.astype(np.float64)
    zs = ZSlicer(v, f)
    for frac in (0.2, 0.5, 0.83):
        z = v[:, 2].min() + frac * np.ptp(v[:, 2])
        pts = anp.plane_points(v, f, np.array([3.0, -2.0, z]), np.array([0.0, 0.0, 2.5]))      # any in-plane origin, unnormalised normal
        ref = zs.points(z)
        assert len(pts) == len(ref) > 20
        np.testing.assert_allclose(_rows(pts[:, :2]), _rows(ref), rtol=0, atol=1e-12)
        np.testing.assert_allclose(pts[:, 2], z, rtol=0, atol=1e-12)


def test_oblique_plane_invariants():
    v, f = load_stl(os.path.join(BONES, "humerus_right.stl"))
    v = v.astype(np.float64)
    rng = np.random.default_rng(5)
    c = v.mean(axis=0)
    for _ in range(4):
        n = rng.normal(size=3)
        pts = anp.plane_points(v, f, c, n)
        nn = n / np.linalg.norm(n)
        assert np.abs((pts - c) @ nn).max() < 1e-9                 # on the plane
        # a closed surface: every crossed edge is shared by two crossing triangles, the downward one yields its point once
        d = (v - c) @ nn
        s = np.where(d < -1e-8, -1, 1)
        e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
        e = np.unique(e, axis=0)
        assert len(pts) == int((s[e[:, 0]] != s[e[:, 1]]).sum())
        assert len(np.unique(np.round(pts, 9), axis=0)) == len(pts)
