"""oracle/clip.py (restatement of trimesh slice_plane for HumeralHeadOsteotomy.resect_mesh, arthroplasty.py:80-87):
known answers on small solids and size-independent properties on the humerus fixture.  trimesh itself is not in the image."""
import os

import numpy as np
import pytest

from conftest import BONES
from oracle import clip
from shoulder_amd.stl import load_stl


def cube():
    v = np.array([[x, y, z] for x in (0, 1) for y in (0, 1) for z in (0, 1)], dtype=np.float64)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    f = []
    for a, b, c, d in quads:
        f += [(a, b, c), (a, c, d)]
    f = np.array(f)
    # outward orientation check: signed volume positive
    assert volume_about(v, f, np.zeros(3)) > 0
    return v, f


def area(v, f):
    t = v[f]
    return 0.5 * np.linalg.norm(np.cross(t[:, 1] - t[:, 0], t[:, 2] - t[:, 0]), axis=1).sum()


def volume_about(v, f, p):
    t = v[f] - p
    return np.einsum("ij,ij->i", t[:, 0], np.cross(t[:, 1], t[:, 2])).sum() / 6.0


def test_cube_cut_in_half():
    v, f = cube()
    ov, of, oe = clip.slice_plane(v, f, [0, 0, 0.5], [0, 0, 1])
    assert np.isclose(area(ov, of), 1 + 4 * 0.5)                  # top face + four half sides, no cap
    assert np.isclose(volume_about(ov, of, np.array([0.5, 0.5, 0.5])), 0.5)
    assert np.allclose(ov[np.unique(oe)][:, 2], 0.5)
    loops = clip.loops_from_edges(oe)
    assert len(loops) == 1 and len(loops[0]) == 8                 # 4 corners + 4 face-diagonal crossings
    assert ov[:, 2].min() == 0.5 and len(ov) == 4 + 8


def test_plane_through_vertices_and_faces_in_the_plane():
    v, f = cube()
    # plane through the top face: faces lying in the plane are kept only if they face against the normal
    ov, of, oe = clip.slice_plane(v, f, [0, 0, 1], [0, 0, 1])
    assert len(of) == 0 and len(oe) == 0
    ov, of, oe = clip.slice_plane(v, f, [0, 0, 1], [0, 0, -1])
    assert len(of) == 12 and len(ov) == 8
    # diagonal plane through four vertices (x = y): signs 0 on the plane, no new vertices off the existing ones
    ov, of, oe = clip.slice_plane(v, f, [0, 0, 0], [1, -1, 0])
    assert np.isclose(volume_about(ov, of, np.zeros(3)), 0.5)
    assert len(ov) == 6
    # a plane that misses the solid
    ov, of, oe = clip.slice_plane(v, f, [0, 0, 2], [0, 0, 1])
    assert len(ov) == 0 and len(of) == 0
    ov, of, oe = clip.slice_plane(v, f, [0, 0, 2], [0, 0, -1])
    assert len(ov) == 8 and np.array_equal(of, f)


@pytest.mark.parametrize("normal,frac", [((0, 0, 1), 0.8), ((0.3, -0.5, 0.81), 0.85), ((1, 0.2, 0.1), 0.5)])
def test_humerus_halves_add_up(normal, frac):
    v, f = load_stl(os.path.join(BONES, "humerus_left.stl"))
    v = v.astype(np.float64)
    n = np.asarray(normal, dtype=np.float64)
    s = v @ n
    o = v.mean(axis=0) + n / (n @ n) * ((s.min() + frac * (s.max() - s.min())) - v.mean(axis=0) @ n)
    hv, hf, he = clip.slice_plane(v, f, o, n)
    rv, rf, re = clip.slice_plane(v, f, o, -n)
    assert len(hf) and len(rf)
    A = area(v, f)
    assert abs(area(hv, hf) + area(rv, rf) - A) < 1e-9 * A
    V = volume_about(v, f, o)
    assert abs(volume_about(hv, hf, o) + volume_about(rv, rf, o) - V) < 1e-9 * abs(V)      # the missing caps are flat about o
    un = n / np.linalg.norm(n)
    for pv, pe in ((hv, he), (rv, re)):
        assert np.abs((pv[np.unique(pe)] - o) @ un).max() < 1e-9
        loops = clip.loops_from_edges(pe)
        assert len(loops) >= 1
    assert ((hv - o) @ un).min() > -1e-7 and ((rv - o) @ un).max() < 1e-7
    # no two vertices closer than the merge tolerance survive
    keys = np.round(hv * 1e8).astype(np.int64)
    assert len(np.unique(keys, axis=0)) == len(hv)
    assert hf.max() == len(hv) - 1 and len(np.unique(hf)) == len(hv)


def test_osteotomy_algebra():
    """oracle/osteotomy.py: the offsets do what arthroplasty.py:89-175 documents."""
    from oracle.osteotomy import OracleOsteotomy, spherical_to_unitxyz
    from oracle.metrics import unitxyz_to_spherical
    rng = np.random.default_rng(2)
    q, _ = np.linalg.qr(rng.normal(size=(3, 3)))
    if np.linalg.det(q) < 0:
        q[:, 0] *= -1
    T = np.identity(4)
    T[:3, :3], T[:3, 3] = q, rng.normal(size=3) * 10
    n_ct = np.array([0.5, 0.3, 0.81])
    n_ct /= np.linalg.norm(n_ct)
    for side in ("left", "right"):
        O = OracleOsteotomy(T, [1.0, 2.0, 3.0], n_ct, side)
        p, n = O.plane(np.identity(4))                      # back in CT: the plane we put in
        assert np.allclose(p, [1, 2, 3]) and np.allclose(n, n_ct)
        assert abs(O.neckshaft_rel()) < 1e-12
        s0 = unitxyz_to_spherical(O.res_normal)
        O.offest_neckshaft(5.0)
        assert O.neckshaft_rel() == pytest.approx(5.0)      # "increasing neckshaft angle is negative" in phi, positive in the measure
        O.offset_retroversion(10.0)
        s1 = unitxyz_to_spherical(O.res_normal)
        assert s1[1] - s0[1] == pytest.approx(-10.0 if side == "left" else 10.0)
        assert np.linalg.norm(O.res_normal) == pytest.approx(1.0)
        z0 = O.res_point.copy()
        O.offset_depth(2.0)
        O.offset_anterior_posterior(1.0)
        O.offset_medial_lateral(1.5)
        assert np.allclose(O.res_point - z0, [-1.0 if side == "left" else 1.0, -1.5, 2.0])
        a = O.retroversion_rel()
        b = O.retroversion_rel()
        assert a != b or abs(O.res_normal[0]) < 1e-15       # the in-place negation of the reference
        assert np.allclose(spherical_to_unitxyz(unitxyz_to_spherical(n_ct)), n_ct)
