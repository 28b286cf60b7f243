"""shoulder_amd.HumeralHeadOsteotomy (reference src/shoulder/arthroplasty.py:13-175) on the GPU against
oracle/osteotomy.py + oracle/clip.py: plane bookkeeping within 1e-4 mm / 1e-8, section and resection through the device."""
import os

import numpy as np
import pytest

from conftest import BONES
from oracle import metrics, xform
from oracle.osteotomy import OracleOsteotomy
from test_oracle_clip import area, volume_about

pytestmark = pytest.mark.gpu
MM = 1e-4


def oracle_for(h):
    L = h.landmarks()
    T_anp = xform.construct_csys(L["canal_axis"], L["anp_axis_normal"])
    L["side"] = h.metrics()["side"]
    return OracleOsteotomy(T_anp, L["anp_plane_point"], L["anp_plane_normal"], L["side"]), L


def same_plane(got, want):
    np.testing.assert_allclose(got.point, want[0], rtol=0, atol=MM)
    np.testing.assert_allclose(got.normal, want[1], rtol=0, atol=1e-7)


def hausdorff(a, b):
    d = np.linalg.norm(a[:, None, :] - b[None, :, :], axis=2)
    return max(d.min(axis=1).max(), d.min(axis=0).max())


@pytest.mark.parametrize("start_csys", ["ct", "canal_te"])
def test_osteotomy_against_oracle(engine, oracle_bones, start_csys):
    import shoulder_amd as shoulder
    hum = shoulder.Humerus(os.path.join(BONES, "humerus_left.stl"), engine=engine)
    h = oracle_bones("humerus_left")
    O, L = oracle_for(h)
    if start_csys == "canal_te":
        hum.apply_csys_canal_transepiconylar()
    T0 = hum.transform.copy()
    ost = shoulder.HumeralHeadOsteotomy(hum)
    np.testing.assert_allclose(hum.transform, T0, rtol=0, atol=1e-12)       # the caller's csys is restored (arthroplasty.py:27-31)
    assert hum.side() == L["side"]
    same_plane(ost.plane, O.plane(T0))
    assert ost.neckshaft_rel == pytest.approx(0.0, abs=1e-9)
    # native plane: the section's largest loop is the anatomic-neck ring; both halves re-assemble the surface
    verts_cur = xform.transform_pts(h.verts.astype(np.float64), T0)
    pts, want = ost.points(), O.points(verts_cur, h.faces, T0)
    assert np.array_equal(pts[0], pts[-1]) and len(pts) > 50
    assert hausdorff(pts, want) < 1e-2
    per = lambda d: np.linalg.norm(np.diff(d, axis=0), axis=1).sum()
    assert per(pts) == pytest.approx(per(want), rel=1e-5)
    head, rest = ost.resect_mesh()
    (ohv, ohf), (orv, orf) = O.resect(verts_cur, h.faces, T0)
    A = area(verts_cur, h.faces)
    assert abs(area(head.vertices, head.faces) + area(rest.vertices, rest.faces) - A) < 1e-9 * A
    p0 = ost.plane.point
    assert volume_about(head.vertices, head.faces, p0) == pytest.approx(volume_about(ohv, ohf, p0), rel=1e-5)
    assert area(head.vertices, head.faces) == pytest.approx(area(ohv, ohf), rel=1e-5)
    assert 0.02 * A < area(head.vertices, head.faces) < 0.3 * A          # a head, not half a bone
    # the offsets, in the order a planning session would apply them
    for step in [("offset_retroversion", (10.0,)), ("offest_neckshaft", (5.0,)), ("offset_depth", (2.0,)), ("offset_depth", (1.5, "anp")),
                 ("offset_depth", (-1.0, "resection")), ("offset_anterior_posterior", (1.0,)), ("offset_medial_lateral", (1.5,))]:
        getattr(ost, step[0])(*step[1])
        getattr(O, step[0])(*step[1])
        same_plane(ost.plane, O.plane(T0))
        assert ost.neckshaft_rel == pytest.approx(O.neckshaft_rel(), abs=1e-6)
    # retroversion_rel flips the stored normal on every read in the reference; two reads restore it
    r1, r2 = ost.retroversion_rel, ost.retroversion_rel
    assert (r1, r2) == (pytest.approx(O.retroversion_rel(), abs=1e-6), pytest.approx(O.retroversion_rel(), abs=1e-6))
    same_plane(ost.plane, O.plane(T0))
    with pytest.raises(ValueError):
        ost.offset_depth(1.0, direction="sideways")
    # after moving the bone the plane follows (plane() maps through the current matrix)
    hum.apply_csys_obb()
    same_plane(ost.plane, O.plane(hum.transform))
    pts2 = ost.points()
    assert np.abs((pts2 - ost.plane.point) @ (ost.plane.normal / np.linalg.norm(ost.plane.normal))).max() < 1e-6


def test_osteotomy_on_a_proximal_humerus(engine):
    """arthroplasty.py:16 takes `bone.ProximalHumerus | bone.Humerus`: the cut-humerus facade supplies everything the class
    touches (apply_csys_canal_articular, anatomic_neck.plane, side, mesh)."""
    import shoulder_amd as shoulder
    hum = shoulder.ProximalHumerus(os.path.join(BONES, "proximal_left_cut.stl"), engine=engine)
    ost = shoulder.HumeralHeadOsteotomy(hum)
    np.testing.assert_array_equal(hum.transform, np.identity(4))
    p = hum.anatomic_neck.plane()
    np.testing.assert_allclose(ost.plane.point, p.point, rtol=0, atol=1e-9)          # native resection plane = anatomic-neck plane
    np.testing.assert_allclose(ost.plane.normal, p.normal, rtol=0, atol=1e-12)
    ost.offset_depth(3.0)
    ost.offest_neckshaft(4.0)
    assert ost.neckshaft_rel == pytest.approx(4.0, abs=1e-9)
    pts = ost.points()
    n = ost.plane.normal / np.linalg.norm(ost.plane.normal)
    assert len(pts) > 50 and np.abs((pts - ost.plane.point) @ n).max() < 1e-6
    head, rest = ost.resect_mesh()
    A = area(hum.mesh.vertices, hum.mesh.faces)
    assert abs(area(head.vertices, head.faces) + area(rest.vertices, rest.faces) - A) < 1e-9 * A
    assert len(head.faces) > 100 and len(rest.faces) > 100
