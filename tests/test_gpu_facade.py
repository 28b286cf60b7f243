"""The drop-in facade (shoulder_amd.Humerus) on the GPU: README flow of the reference
(README.md:22-41), accessor shapes / caching / csys semantics of SURVEY 8(b), checked against the
oracle."""
import os

import numpy as np
import pytest

from conftest import BONES
from oracle import xform

pytestmark = pytest.mark.gpu
MM = 1e-4


@pytest.fixture(scope="module")
def hum(engine):
    engine.reset_params()      # (whatever an earlier test file left: bone kind, UNet element type, cut-offs)
    import shoulder_amd as shoulder
    return shoulder.Humerus(os.path.join(BONES, "humerus_left.stl"), engine=engine)


def test_constructor_state(hum, oracle_bones):
    h = oracle_bones("humerus_left")
    assert hum.stl_file.name == "humerus_left.stl"
    np.testing.assert_array_equal(hum.transform, np.identity(4))
    assert hum.mesh.vertices.shape == (16222, 3) and hum.mesh.faces.shape == (32440, 3)
    assert hum.surgical_neck.neck_z == pytest.approx(h.neck["neck_z"], abs=1e-7)
    np.testing.assert_allclose(hum.surgical_neck.points, h.neck["points_ct"], rtol=0, atol=MM)


def test_accessors_in_ct(hum, oracle_bones):
    h = oracle_bones("humerus_left")
    L = h.landmarks()
    np.testing.assert_allclose(hum.canal.axis(), L["canal_axis"], rtol=0, atol=MM)
    np.testing.assert_allclose(hum.canal.points(), L["canal_points"], rtol=0, atol=MM)
    assert hum.canal.points().shape == (80, 3)
    np.testing.assert_allclose(hum.bicipital_groove.points(), L["groove_points"], rtol=0, atol=MM)
    assert hum.bicipital_groove.bg_theta == L["bg_theta"]
    np.testing.assert_allclose(hum.bicipital_groove.axis(), L["groove_axis"], rtol=0, atol=MM)
    np.testing.assert_allclose(hum.anatomic_neck.points(), L["anp_points"], rtol=0, atol=MM)
    p = hum.anatomic_neck.plane()
    np.testing.assert_allclose(p.point, L["anp_plane_point"], rtol=0, atol=MM)
    np.testing.assert_allclose(p.normal, L["anp_plane_normal"], rtol=0, atol=1e-8)
    np.testing.assert_allclose(hum.anatomic_neck.axis_normal(), L["anp_axis_normal"], rtol=0, atol=MM)
    np.testing.assert_allclose(hum.anatomic_neck.axis_central(), L["anp_axis_central"], rtol=0, atol=MM)
    np.testing.assert_allclose(hum.trans_epiconylar.axis(), L["te_axis"], rtol=0, atol=MM)
    # plane_points (anatomic_neck.py:155-172): the section of the CT mesh by the neck plane, vs the oracle -- once on the
    # device's own plane (same input: same crossing triangles, coordinates to rounding), once end to end
    from oracle import anp as oanp
    pp = hum.anatomic_neck.plane_points()
    assert len(pp) > 50
    assert np.abs((pp - p.point) @ p.normal).max() < 1e-9
    key = lambda a: a[np.lexsort((a[:, 2], a[:, 1], a[:, 0]))]
    same_plane = oanp.plane_points(h.verts.astype(np.float64), h.faces, p.point, p.normal)
    assert len(same_plane) == len(pp)
    np.testing.assert_allclose(key(pp), key(same_plane), rtol=0, atol=1e-9)
    ref_pp = h.anp_plane_points()
    assert len(ref_pp) == len(pp)
    np.testing.assert_allclose(key(pp), key(ref_pp), rtol=0, atol=MM)


def test_apply_csys_roundtrip(hum, oracle_bones):
    h = oracle_bones("humerus_left")
    L = h.landmarks()
    T = hum.apply_csys_canal_transepiconylar()
    np.testing.assert_allclose(T, L["csys"], rtol=0, atol=1e-6)
    assert hum.transform is hum._tfrm.matrix
    # in the canal/TE frame: canal axis along +z through the origin, TE axis in the xz... (y = 0 plane normal)
    ax = hum.canal.axis()
    np.testing.assert_allclose(ax, xform.transform_pts(L["canal_axis"], L["csys"]), rtol=0, atol=MM)
    assert abs(ax[:, 0]).max() < 1e-6 and abs(ax[:, 1]).max() < 1e-6 and ax[0, 2] > 0 > ax[1, 2]
    np.testing.assert_allclose(hum.anatomic_neck.points(), xform.transform_pts(L["anp_points"], L["csys"]), rtol=0, atol=MM)
    np.testing.assert_allclose(hum.mesh.vertices, xform.transform_pts(h.verts.astype(np.float64), L["csys"]), rtol=0, atol=MM)
    T2 = hum.apply_csys_obb()
    np.testing.assert_allclose(T2, h.T_obb, rtol=0, atol=1e-6)
    np.testing.assert_allclose(hum.canal.axis(), h.canal["axis_obb"], rtol=0, atol=MM)
    hum.apply_csys_ct()
    np.testing.assert_array_equal(hum.transform, np.identity(4))
    np.testing.assert_allclose(hum.canal.axis(), L["canal_axis"], rtol=0, atol=MM)
    t = np.array([1.0, -2.0, 3.0])
    hum.apply_translation(t)
    np.testing.assert_allclose(hum.canal.axis(), L["canal_axis"] + t, rtol=0, atol=MM)
    hum.apply_csys_ct()


def test_metrics(hum, oracle_bones):
    """README.md:33-36: radius_curvature / neckshaft / retroversion (+ side), incl. the reference's quirk that
    retroversion() uses axis_normal() in the coordinate system applied at call time."""
    h = oracle_bones("humerus_left")
    hum.apply_csys_ct()
    m = h.metrics()
    assert hum.side() == m["side"]      # (with the synthetic teacher UNet the head axis is not anatomical)
    assert hum.neckshaft() == pytest.approx(m["neckshaft"], abs=1e-6)
    assert hum.radius_curvature() == pytest.approx(m["radius_curvature"], abs=1e-6)
    assert hum.retroversion() == pytest.approx(m["retroversion"], abs=1e-6)
    T = hum.apply_csys_canal_transepiconylar()
    m2 = h.metrics(axis_normal_current=xform.transform_pts(h.anp["axis_normal_ct"], T))
    assert hum.retroversion() == pytest.approx(m2["retroversion"], abs=1e-6)
    assert hum.neckshaft() == pytest.approx(m["neckshaft"], abs=1e-6)       # frame independent
    hum.apply_csys_ct()


def test_metrics_right_side(engine, oracle_bones):
    import shoulder_amd as shoulder
    r = shoulder.Humerus(os.path.join(BONES, "humerus_right.stl"), engine=engine)
    m = oracle_bones("humerus_right").metrics()
    assert r.side() == m["side"]
    assert r.retroversion() == pytest.approx(m["retroversion"], abs=1e-6)
    assert r.radius_curvature() == pytest.approx(m["radius_curvature"], abs=1e-6)


def test_class_hierarchy_and_articular_csys(hum, oracle_bones):
    """bone.py:24,109: `Humerus(ProximalHumerus)`; the inherited apply_csys_canal_articular (bone.py:53-62) takes its matrix from
    the device record (k_pack) and is checked against the oracle's construct_csys(canal axis, neck-normal axis)."""
    import shoulder_amd as shoulder
    assert isinstance(hum, shoulder.ProximalHumerus) and issubclass(shoulder.Humerus, shoulder.ProximalHumerus)
    import shoulder as ref_name                      # the reference's package name resolves to the same classes
    assert ref_name.Humerus is shoulder.Humerus and ref_name.bone.ProximalHumerus is shoulder.ProximalHumerus
    h = oracle_bones("humerus_left")
    L = h.landmarks()
    T = hum.apply_csys_canal_articular()
    np.testing.assert_allclose(T, L["csys_articular"], rtol=0, atol=1e-6)
    np.testing.assert_allclose(hum.mesh.vertices, xform.transform_pts(h.verts.astype(np.float64), L["csys_articular"]), rtol=0, atol=MM)
    hum.apply_csys_ct()


def test_default_engine_warns_about_teacher_weights(monkeypatch):
    """Without a model the facade still runs (teacher stand-in) but says so loudly (RuntimeWarning)."""
    from shoulder_amd import bone
    monkeypatch.setattr(bone, "_DEFAULT_ENGINE", None)
    monkeypatch.delenv("SHOULDER_UNET_ONNX", raising=False)
    with pytest.warns(RuntimeWarning, match="TEACHER"):
        e = bone.default_engine()
    try:
        assert bone.default_engine() is e            # created once
    finally:
        e.close()
        monkeypatch.setattr(bone, "_DEFAULT_ENGINE", None)


def test_caller_engine_configuration_survives_the_facade(oracle_bones):
    """Engine.set_params is read-modify-write: a caller's bf16 engine stays bf16 when a facade object sets its bone kind, and a
    second bone on the same engine brings its own groove parameters back (_ensure_loaded)."""
    import shoulder_amd as shoulder
    from conftest import _teacher_weights
    from shoulder_amd import _lib, unet_spec
    from shoulder_amd.engine import Engine
    e = Engine(0)
    try:
        e.load_rfc(); e.load_unet(_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
        e.set_params(unet_dtype=_lib.UNET_BF16)
        a = shoulder.Humerus(os.path.join(BONES, "humerus_left.stl"), engine=e)
        assert e.get_params()["unet_dtype"] == _lib.UNET_BF16 and e.get_params()["bone_kind"] == _lib.BONE_HUMERUS
        pa = a.bicipital_groove.points(deg_window=9).copy()
        b = shoulder.Humerus(os.path.join(BONES, "humerus_right.stl"), engine=e)
        b.bicipital_groove.points()                  # default window 7 on the shared engine
        a._ensure_loaded()
        assert e.get_params()["groove_deg_window"] == 9.0 and e.get_params()["unet_dtype"] == _lib.UNET_BF16
        np.testing.assert_array_equal(a.bicipital_groove.points(), pa)
    finally:
        e.close()


def test_errors(hum):
    with pytest.raises(ValueError, match="Invalid transformation matrix shape"):
        hum.apply_csys_custom(np.identity(3))
    import shoulder_amd as shoulder
    with pytest.raises((FileNotFoundError, ValueError)):
        shoulder.ProximalHumerus("x.stl", engine=hum._engine)
    with pytest.raises((FileNotFoundError, ValueError)):
        shoulder.Humerus("does_not_exist.stl", engine=hum._engine)


def test_csys_bookkeeping_against_the_references_own_vectors(engine):
    """VERDICT r3 item 7 / SURVEY a24: `apply_csys_custom(from_ct=True / False)`, `apply_translation`, `apply_csys_ct` of the facade
    (shoulder_amd/bone.py) against tests/golden/csys_golden.npz -- vectors the reference's own bone.py:66-105 / base.py:24-63 /
    landmark `transform_landmark` methods produced for a seeded set of CT landmarks and vertices (make_csys_golden.py).  The
    facade object gets the same CT caches, then the same nine calls; matrices are compared exactly, every re-expressed landmark
    and the mesh -- incl. the reference's quirk that `from_ct=False` and `apply_translation` apply the cumulative matrix to the
    already moved mesh -- to 1e-9 mm (the device's k_affine_f64 vs NumPy's dot)."""
    import shoulder_amd as shoulder
    from conftest import GOLDEN
    from shoulder_amd.base import Mesh
    g = np.load(os.path.join(GOLDEN, "csys_golden.npz"))
    engine.reset_params()
    h = shoulder.Humerus(os.path.join(BONES, "humerus_left.stl"), engine=engine)
    h._mesh_ct = Mesh(g["in_verts"].copy(), np.array([[0, 1, 2]], dtype=np.int32), engine)
    h.mesh = h._mesh_ct.copy()
    h.canal._axis_ct, h.canal._points_ct = g["in_canal_axis"].copy(), g["in_canal_points"].copy()
    h.trans_epiconylar._axis_ct = g["in_te_axis"].copy()
    h.bicipital_groove._axis_ct, h.bicipital_groove._points_ct = g["in_groove_axis"].copy(), g["in_groove_points"].copy()
    an = h.anatomic_neck
    an._points_ct, an._plane_points_ct = g["in_anp_points"].copy(), g["in_anp_plane_points"].copy()
    an._normal_axis_ct, an._central_axis_ct = g["in_anp_axis_normal"].copy(), g["in_anp_axis_central"].copy()
    h.surgical_neck.points_ct = g["in_surgical_neck"].copy()
    h.surgical_neck.points = g["in_surgical_neck"].copy()
    for k, s in enumerate(g["ops"]):
        op, arg = str(s).split(":")
        if op == "custom_ct":
            r = h.apply_csys_custom(g[arg].copy(), from_ct=True)
        elif op == "custom_rel":
            r = h.apply_csys_custom(g[arg].copy(), from_ct=False)
        elif op == "translate":
            r = h.apply_translation(g[arg].copy())
        else:
            r = h.apply_csys_ct()
        np.testing.assert_array_equal(r, g[f"s{k}_returned"], err_msg=f"step {k} returned matrix")
        np.testing.assert_array_equal(h.transform, g[f"s{k}_transform"])
        np.testing.assert_array_equal(h._tfrm.matrix, g[f"s{k}_tfrm"])
        tol = dict(rtol=0, atol=1e-9)
        np.testing.assert_allclose(h.mesh.vertices, g[f"s{k}_mesh"], err_msg=f"step {k} mesh", **tol)
        got = {"canal_axis": h.canal._axis, "canal_points": h.canal._points, "te_axis": h.trans_epiconylar._axis,
               "groove_axis": h.bicipital_groove._axis, "groove_points": h.bicipital_groove._points, "anp_points": an._points,
               "anp_plane_points": an._plane_points, "anp_axis_normal": an._normal_axis, "anp_axis_central": an._central_axis,
               "surgical_neck": h.surgical_neck.points}
        for n, v in got.items():
            np.testing.assert_allclose(v, g[f"s{k}_{n}"], err_msg=f"step {k} {n}", **tol)
    for bad in (np.identity(3), np.zeros((4, 4)).tolist()):
        with pytest.raises(ValueError, match="Invalid transformation matrix shape"):
            h.apply_csys_custom(bad)
