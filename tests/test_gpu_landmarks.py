"""GPU parity, groove -> anatomic neck (UNet) -> trans-epicondylar -> csys, through the C-ABI,
against the oracle on the reference's STL fixtures (oracle OBB transform injected).
Integer selections exact; coordinates within 1e-6 mm here (north-star budget 1e-4 mm)."""
import numpy as np
import pytest

from oracle import unet as o_unet
from shoulder_amd import _lib

pytestmark = pytest.mark.gpu
NAMES = ["humerus_left", "humerus_right", "humerus_left_trab"]
TOL = 1e-6
STAGES = _lib.STAGE_ALL & ~_lib.STAGE_OBB


@pytest.fixture(scope="module")
def ran(engine, oracle_bones):
    engine.reset_params()      # (whatever an earlier test file left: bone kind, UNet element type, cut-offs)
    hs = [oracle_bones(n) for n in NAMES]
    engine.upload([(h.verts, h.faces) for h in hs])
    engine.store("obb_transform", np.stack([h.T_obb for h in hs]))
    lm = engine.run(STAGES)
    return hs, lm


def test_groove_features_and_rfc(engine, ran):
    hs, _ = ran
    B = len(hs)
    npk = engine.fetch("groove.npk", np.int32, (B, 330))
    xraw = engine.fetch("groove.xraw", np.float64, (B, 330 * 7, 9))
    xs = engine.fetch("groove.xs", np.float64, (B, 330 * 7, 9))
    pth = engine.fetch("groove.ptheta", np.float64, (B, 330 * 7))
    proba = engine.fetch("groove.proba", np.float32, (B, 330 * 7))
    for b, h in enumerate(hs):
        g = h.groove
        valid = (np.arange(7)[None, :] < npk[b][:, None]).ravel()
        np.testing.assert_array_equal(npk[b], np.bincount(g["rows"], minlength=330))      # peaks per row: exact
        np.testing.assert_allclose(pth[b][valid], g["peak_theta"], rtol=0, atol=1e-12)
        np.testing.assert_allclose(xraw[b][valid], g["X_raw"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(xs[b][valid], g["X"], rtol=0, atol=1e-9)
        np.testing.assert_array_equal(proba[b][valid], g["proba"])                        # forest output: exact f32


def test_groove_theta_points_axis(engine, ran):
    hs, lm = ran
    B = len(hs)
    li = engine.fetch("groove.local_idx", np.int32, (B, 330))
    for b, h in enumerate(hs):
        g = h.groove
        assert lm["bg_theta"][b] == g["bg_theta"]                                          # KDE grid argmax: exact
        np.testing.assert_array_equal(li[b], g["local_idx"])                               # local minima: exact
        np.testing.assert_allclose(lm["groove_points"][b], g["points_ct"], rtol=0, atol=TOL)
        np.testing.assert_allclose(lm["groove_axis"][b], g["axis_ct"], rtol=0, atol=TOL)


def test_anp_image(engine, ran):
    hs, _ = ran
    B = len(hs)
    img = engine.fetch("anp.image", np.float32, (B, 512, 512))
    roll = engine.fetch("anp.roll", np.int32, (B, 512))
    for b, h in enumerate(hs):
        np.testing.assert_array_equal(roll[b], h.anp_input["roll"])                        # roll index: exact
        exp = h.anp_input["image"].astype(np.float32)
        assert np.abs(img[b] - exp).max() <= 2 * np.finfo(np.float32).eps                  # 1-ulp atan2/rounding headroom


def test_unet_bit_exact(engine, ran, unet_weights):
    """f32 MFMA path == the float32 fma-chain restatement, bit for bit, on the images it was fed."""
    hs, _ = ran
    B = len(hs)
    img = engine.fetch("anp.image", np.float32, (B, 512, 512))
    lg = engine.fetch("anp.logits", np.float32, (B, 512, 512))
    for b in range(B):
        exp = o_unet.forward_chain(unet_weights, img[b])
        np.testing.assert_array_equal(lg[b], exp)


def test_anp_points_plane_axes(engine, ran):
    hs, lm = ran
    for b, h in enumerate(hs):
        a = h.anp
        assert lm["n_anp"][b] == len(a["points_obb"])                                      # edge pixels: exact
        assert lm["n_articular"][b] == len(a["articular_obb"])
        K = min(len(a["points_ct"]), 4096)
        np.testing.assert_allclose(lm["anp_points"][b][:K], a["points_ct"][:K], rtol=0, atol=TOL)
        np.testing.assert_allclose(lm["anp_plane_point"][b], a["plane_point_ct"], rtol=0, atol=TOL)
        np.testing.assert_allclose(lm["anp_plane_normal"][b], a["plane_normal_ct"], rtol=0, atol=1e-9)
        np.testing.assert_allclose(lm["anp_axis_normal"][b], a["axis_normal_ct"], rtol=0, atol=TOL)
        np.testing.assert_allclose(lm["anp_axis_central"][b], a["axis_central_ct"], rtol=0, atol=TOL)


def test_te_csys_and_record(engine, ran):
    hs, lm = ran
    B = len(hs)
    row = engine.fetch("te.row", np.int32, (B,))
    for b, h in enumerate(hs):
        assert row[b] == h.te["idx_max"]                                                    # widest slice: exact
        np.testing.assert_allclose(lm["te_axis"][b], h.te["axis_ct"], rtol=0, atol=TOL)
        np.testing.assert_allclose(lm["canal_axis"][b], h.canal["axis_ct"], rtol=0, atol=TOL)
        np.testing.assert_allclose(lm["csys"][b], h.csys_canal_transepicondylar(), rtol=0, atol=1e-7)
        assert lm["neck_index"][b] == h.neck["bkp"] and lm["status"][b] == 0
        assert abs(lm["neck_z"][b] - h.neck["neck_z"]) < 1e-12
        assert abs(lm["z_length"][b] - h.obb["z_length"]) < 1e-9
