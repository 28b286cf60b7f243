"""`shoulder.ProximalHumerus` (bone.py:24-64) on the device against the oracle restatement
(oracle/prox.py) on a humerus cut in the shaft (tests/golden/make_proximal_fixture.py)."""
import os

import numpy as np
import pytest

from shoulder_amd import _lib

pytestmark = pytest.mark.gpu
MM = 1e-4
PROX_MASK = _lib.STAGE_ALL & ~(_lib.STAGE_DISTAL | _lib.STAGE_TE)


@pytest.fixture(scope="module")
def oracle_prox(rfc_tables, unet_weights):
    from oracle.prox import OracleProximalHumerus
    from tests.conftest import BONES, _ensure_chain_lib
    _ensure_chain_lib()
    return OracleProximalHumerus.from_stl(os.path.join(BONES, "proximal_left_cut.stl"), rfc_tables, unet_weights, unet_eval="chain")


@pytest.fixture(scope="module")
def ran(engine, oracle_prox):
    h = oracle_prox
    engine.set_params(bone_kind=_lib.BONE_PROXIMAL)
    try:
        engine.upload([(h.verts, h.faces)])
        lm = engine.run(PROX_MASK).copy()
        scan = engine.fetch("pobb.area_total", np.float64, (1, 100))[0].copy()
        cut_idx = engine.fetch("pobb.cutoff_idx", np.int32, (1, 2))[0].copy()
    finally:
        engine.reset_params()
    return lm, scan, cut_idx


def test_prox_obb(oracle_prox, ran):
    h = oracle_prox
    lm, scan, cut_idx = ran
    assert lm["status"][0] == 0
    o = h.obb
    assert bool(lm["flipped"][0]) == o["flipped"]
    ref_scan = o["z_area"][::-1] if o["flipped"] else o["z_area"]        # the device keeps scan order (ascending raw z)
    np.testing.assert_allclose(scan, ref_scan, rtol=1e-10, atol=1e-7)
    assert tuple(cut_idx) == o["canal_zs"]                                 # canal range: exact
    np.testing.assert_array_equal(lm["canal_cutoff"][0], np.asarray(o["cutoff_pcts"]))
    np.testing.assert_allclose(lm["obb_transform"][0][:3, :3], h.T_obb[:3, :3], rtol=0, atol=1e-9)
    np.testing.assert_allclose(lm["obb_transform"][0][:3, 3], h.T_obb[:3, 3], rtol=0, atol=1e-6)


def test_prox_landmarks(oracle_prox, ran):
    h = oracle_prox
    lm, _, _ = ran
    L = h.landmarks()
    assert lm["neck_index"][0] == h.neck["bkp"]                           # change point on areas1((0.2, 0.99)): exact
    assert abs(lm["neck_z"][0] - h.neck["neck_z"]) < 1e-7
    assert lm["bg_theta"][0] == L["bg_theta"]
    assert lm["n_anp"][0] == len(L["anp_points"])
    for key, ref in (("canal_axis", L["canal_axis"]), ("groove_axis", L["groove_axis"]), ("anp_axis_normal", L["anp_axis_normal"]),
                     ("anp_axis_central", L["anp_axis_central"]), ("anp_plane_point", L["anp_plane_point"])):
        np.testing.assert_allclose(lm[key][0], ref, rtol=0, atol=MM, err_msg=key)
    np.testing.assert_allclose(lm["groove_points"][0], L["groove_points"], rtol=0, atol=MM)
    np.testing.assert_allclose(lm["anp_points"][0][: lm["n_anp"][0]], L["anp_points"], rtol=0, atol=MM)
    np.testing.assert_allclose(lm["csys"][0], L["csys"], rtol=0, atol=1e-6)      # apply_csys_canal_articular
    m = h.metrics()
    assert ("left", "right")[lm["side"][0]] == m["side"]
    assert abs(lm["neckshaft"][0] - m["neckshaft"]) < 1e-6
    assert abs(lm["radius_curvature"][0] - m["radius_curvature"]) < 1e-6
    assert np.isnan(lm["retroversion"][0])


def test_prox_rejects_distal_stages(engine, oracle_prox):
    engine.set_params(bone_kind=_lib.BONE_PROXIMAL)
    try:
        engine.upload([(oracle_prox.verts, oracle_prox.faces)])
        with pytest.raises(Exception):
            engine.run(_lib.STAGE_ALL)
    finally:
        engine.reset_params()


def test_prox_batch_equivariance(engine, oracle_prox):
    """A batch of similarity copies of the cut mesh: every copy's landmarks are the transform of the first one's."""
    from shoulder_amd import synth
    h = oracle_prox
    B = 4
    T = synth.similarity_transforms(B, h.verts, seed=21)
    c = np.asarray(h.verts, dtype=np.float64).mean(axis=0)
    for b in range(B):      # rigid copies only: the ProxObb threshold `grad < 10` (mm^2 per section) is not scale invariant
        sc = np.cbrt(np.linalg.det(T[b][:3, :3]))
        R = T[b][:3, :3] / sc
        t = T[b][:3, 3] - c + sc * (R @ c)
        T[b][:3, :3] = R
        T[b][:3, 3] = c + t - R @ c
    T[0] = np.eye(4)
    engine.set_params(bone_kind=_lib.BONE_PROXIMAL)
    try:
        engine.upload([(h.verts, h.faces)])
        engine.synth_batch(T)
        lm = engine.run(PROX_MASK).copy()
    finally:
        engine.reset_params()
    assert (lm["status"] == 0).all()
    assert (lm["neck_index"] == lm["neck_index"][0]).all()
    np.testing.assert_array_equal(lm["canal_cutoff"], np.repeat(lm["canal_cutoff"][:1], B, axis=0))
    for b in range(1, B):
        ref = synth.apply_similarity(T[b], lm["canal_axis"][0])
        np.testing.assert_allclose(lm["canal_axis"][b], ref, rtol=0, atol=2e-3)


def test_proximal_facade(engine, oracle_prox):
    """`shoulder.ProximalHumerus(stl)` accessors (bone.py:24-105) against the oracle."""
    import shoulder_amd as shoulder
    from tests.conftest import BONES
    h = oracle_prox
    L = h.landmarks()
    p = shoulder.ProximalHumerus(os.path.join(BONES, "proximal_left_cut.stl"), engine=engine)
    try:
        assert p.cutoff_pcts == [float(x) for x in h.obb["cutoff_pcts"]] and p.cutoff_bot == h.obb["cutoff_bot"]
        assert p.surgical_neck.neck_z == pytest.approx(h.neck["neck_z"], abs=1e-7)
        np.testing.assert_allclose(p.canal.axis(), L["canal_axis"], rtol=0, atol=MM)
        np.testing.assert_allclose(p.canal.points(), L["canal_points"], rtol=0, atol=MM)
        assert p.canal.points().shape == h.canal["points_obb"].shape
        np.testing.assert_allclose(p.bicipital_groove.points(), L["groove_points"], rtol=0, atol=MM)
        np.testing.assert_allclose(p.anatomic_neck.axis_normal(), L["anp_axis_normal"], rtol=0, atol=MM)
        m = h.metrics()
        assert p.side() == m["side"]
        assert p.neckshaft() == pytest.approx(m["neckshaft"], abs=1e-6)
        assert p.radius_curvature() == pytest.approx(m["radius_curvature"], abs=1e-6)
        assert not hasattr(p, "trans_epiconylar") and not hasattr(p, "retroversion")
        T = p.apply_csys_canal_articular()
        np.testing.assert_allclose(T, L["csys"], rtol=0, atol=1e-6)
        ax = p.canal.axis()                               # canal axis is +z through the origin in its own csys
        assert abs(ax[0][0]) < 1e-6 and abs(ax[0][1]) < 1e-6 and ax[0][2] > 0
        with pytest.raises(AttributeError):
            p.apply_csys_canal_transepiconylar()
        p.apply_csys_ct()
        np.testing.assert_allclose(p.canal.axis(), L["canal_axis"], rtol=0, atol=MM)
    finally:
        engine.reset_params()


def test_prox_similarity_copies(engine, rfc_tables, unet_weights):
    """Regression from a randomized sweep: in a similarity copy of the cut humerus the one feature that is constant in
    exact arithmetic has a standard deviation of ~1e-13; sklearn's StandardScaler calls it constant (`_is_constant_feature`)
    and so must the device, or the scaled column is noise, the forest probabilities change and bg_theta jumps."""
    from oracle.prox import OracleProximalHumerus
    from shoulder_amd import synth
    from shoulder_amd.stl import load_stl
    from tests.conftest import BONES, _ensure_chain_lib
    _ensure_chain_lib()
    v, f = load_stl(os.path.join(BONES, "proximal_left_cut.stl"))
    T = synth.similarity_transforms(24, v, seed=900)
    meshes = [(synth.apply_similarity(T[i], v), f) for i in (4, 10)]
    engine.set_params(bone_kind=_lib.BONE_PROXIMAL)
    try:
        engine.upload(meshes)
        lm = engine.run(PROX_MASK).copy()
        xs = engine.fetch("groove.xs", np.float64, (2, 330 * 7, 9)).copy()
        npk = engine.fetch("groove.npk", np.int32, (2, 330)).copy()
    finally:
        engine.reset_params()
    for b, (mv, mf) in enumerate(meshes):
        h = OracleProximalHumerus(mv, mf, rfc_tables, unet_weights, unet_eval="chain")
        L = h.landmarks()
        valid = (np.arange(7)[None, :] < npk[b][:, None]).ravel()
        np.testing.assert_allclose(xs[b][valid], h.groove["X"], rtol=0, atol=1e-9)      # the scaled features incl. the constant column
        assert lm["status"][b] == 0 and lm["bg_theta"][b] == L["bg_theta"] and lm["n_anp"][b] == len(L["anp_points"])
        np.testing.assert_allclose(lm["anp_points"][b].reshape(-1, 3)[: lm["n_anp"][b]], L["anp_points"], rtol=0, atol=1e-6)
        np.testing.assert_allclose(lm["canal_axis"][b], L["canal_axis"], rtol=0, atol=1e-6)
