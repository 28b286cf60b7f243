"""The oracle against golden vectors captured from the reference's own code
(tests/golden/make_golden.py).  CPU only."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import anp, groove, slices, xform


def _g(name):
    return np.load(os.path.join(GOLDEN, name))


def test_utils_golden():
    g = _g("utils_golden.npz")
    np.testing.assert_array_equal(xform.transform_pts(g["pts"], g["T"]), g["transform_pts"])
    np.testing.assert_array_equal(xform.inv_transform(g["T"]), g["inv_transform"])
    np.testing.assert_array_equal(xform.construct_csys(g["vz"], g["vy"]), g["construct_csys"])
    np.testing.assert_array_equal(xform.construct_csys(g["vz"], g["vy"][::-1]), g["construct_csys_swapped"])
    np.testing.assert_array_equal(xform.translate_transform(g["t"]), g["translate_transform"])
    np.testing.assert_array_equal(xform.unit_vector(g["vz"][0], g["vz"][1]), g["unit_vector"])


def test_rect_helpers_golden():
    from oracle import te
    g = _g("utils_golden.npz")
    r = te.min_area_rect(g["rect"][:-1])
    assert abs(r["L"] - float(g["major_axis_dist"])) < 1e-12
    # azimuth of the long side, degrees clockwise from +y in (0,180] (utils.py:36-55)
    az = np.degrees(np.arctan2(r["major"][0], r["major"][1]))
    az = az if az > 0 else az + 180
    assert abs((az - float(g["rect_azimuth"]) + 90) % 180 - 90) < 1e-9


def test_cutoff_golden():
    g = _g("slice_golden.npz")
    for (n, c0, c1), (a, b) in zip(g["cutoff_cases"], g["cutoff_ranges"]):
        assert slices.cutoff_range(int(n), (c0, c1)) == (a, b)
    # the exact index ranges the path relies on
    assert slices.cutoff_range(200, (0.35, 0.75)) == (50, 130)
    assert slices.cutoff_range(200, (0.70, 0.99)) == (2, 60)
    assert slices.cutoff_range(600, (0.2, 0.75)) == (150, 480)
    assert slices.cutoff_range(600, (0.0, 0.852)) == (88, 600)
    assert slices.cutoff_range(200, (0.8, 0.99)) == (2, 39)


def test_resample_polar_golden():
    g = _g("slice_golden.npz")
    np.testing.assert_array_equal(slices.resample_polygon(g["poly"], 100), g["resample_100"])
    np.testing.assert_array_equal(slices.resample_polygon(g["poly"], 512), g["resample_512"])
    np.testing.assert_array_equal(slices.cart2pol_no_sort(g["poly"][:, 0], g["poly"][:, 1]), g["cart2pol_no_sort"])
    c = _g("contours_left.npz")
    ixy = c["ixy"].astype(np.float64)
    cen = c["centroids"].astype(np.float64)
    rows = g["rows"]
    cent = ixy[rows] - cen[rows][:, :, None]
    np.testing.assert_array_equal(cent, g["ixy_centered_rows"])
    for k, r in enumerate(rows):
        np.testing.assert_array_equal(slices.roll_to_argmin_theta(slices.cart2pol_no_sort(ixy[r][0], ixy[r][1])),
                                      g["itr_start_rows"][k])
        np.testing.assert_array_equal(slices.roll_to_argmin_theta(slices.cart2pol_no_sort(cent[k][0], cent[k][1])),
                                      g["itr_centered_start_rows"][k])


def _polar_centered_start(ixy, cen):
    return np.stack([slices.roll_to_argmin_theta(slices.cart2pol_no_sort(p[0], p[1])) for p in ixy - cen[:, :, None]])


def _synthetic_contours(n=600, m=512, seed=99):
    rng = np.random.default_rng(seed)
    zs = np.linspace(170.0, 128.0, n)
    ixy = np.zeros((n, 2, m))
    cents = np.zeros((n, 2))
    ph = rng.uniform(-0.1, 0.1, n)
    for i in range(n):
        t = np.linspace(-np.pi, np.pi, 400)
        f = i / (n - 1)
        r = 22 - 6 * f + 1.5 * np.cos(2 * t + ph[i]) - 2.2 * np.exp(-((t - 0.9) / 0.12) ** 2) \
            - 1.1 * np.exp(-((t + 1.7) / 0.15) ** 2)
        xy = np.c_[3 + r * np.cos(t), -2 + r * np.sin(t)]
        xy[-1] = xy[0]
        ixy[i] = slices.resample_polygon(xy, m).T
        cents[i] = 0.5 * (xy.min(axis=0) + xy.max(axis=0))
    return zs, ixy, cents


@pytest.mark.parametrize("tag", ["left", "synth"])
def test_groove_golden(tag, rfc_tables):
    g = _g(f"groove_golden_{tag}.npz")
    if tag == "left":
        c = _g("contours_left.npz")
        ixy, cen, zs = c["ixy"].astype(np.float64), c["centroids"].astype(np.float64), c["zs"].astype(np.float64)
    else:
        zs, ixy, cen = _synthetic_contours()
    a, b = slices.cutoff_range(600, (0.2, 0.75))
    polar = _polar_centered_start(ixy, cen)[a:b]
    out = groove.groove_points(polar, zs[a:b], cen[a:b], g["canal_axis_ct"], g["T_obb"], rfc_tables)
    assert out["bg_theta"] == float(g["bg_theta"])
    # rows of X may be ordered differently only when a slice has > 7 peaks (argpartition order, B-5)
    assert out["X"].shape == g["X"].shape
    np.testing.assert_allclose(np.sort(out["X"], axis=0), np.sort(g["X"], axis=0), rtol=0, atol=1e-12)
    np.testing.assert_allclose(np.sort(out["peak_theta"]), np.sort(g["peak_theta"]), rtol=0, atol=0)
    np.testing.assert_array_equal(out["points_obb"], g["points_obb"])
    np.testing.assert_array_equal(out["points_ct"], g["points_ct"])


def test_anp_golden():
    g = _g("anp_golden_left.npz")
    c = _g("contours_left.npz")
    ixy, zs = c["ixy"].astype(np.float64), c["zs"].astype(np.float64)
    a, b = slices.cutoff_range(600, (0.0, 0.852))
    itr = np.stack([slices.roll_to_argmin_theta(slices.cart2pol_no_sort(p[0], p[1])) for p in ixy])[a:b]
    img, shft, _ = anp.anp_image(itr, float(g["bg_theta"]))
    np.testing.assert_array_equal(img.astype(np.float32), g["image_f32"])
    m = anp.mask_points(g["logits_f32"], shft, zs[a:b])
    np.testing.assert_array_equal(m["points_obb"], g["points_obb"])
    assert len(m["articular_obb"]) == int(g["n_articular"])
    np.testing.assert_array_equal(anp.to_ct(m["points_obb"], c["T_obb"]), g["points_ct"])


def test_metrics_against_reference_bone_props():
    """oracle/metrics.py side / retroversion / neckshaft vs the reference's own bone_props.py:12-112
    (tests/golden/make_metrics_golden.py; odd cases carry a non-identity current csys: the retroversion quirk)."""
    from oracle import metrics as m
    from oracle import xform
    G = np.load(os.path.join(GOLDEN, "metrics_landmarks_golden.npz"))
    sides = set()
    for c in range(int(G["n"])):
        g = lambda k: G[f"c{c}_{k}"]
        s = m.side(g("canal"), g("central"), g("groove"))
        assert s == str(g("side"))
        sides.add(s)
        an_cur = xform.transform_pts(g("normal"), g("T_current"))
        assert m.retroversion(g("canal"), g("te"), an_cur, s) == pytest.approx(float(g("retroversion")), abs=1e-9)
        assert m.neckshaft(g("canal"), g("normal")) == pytest.approx(float(g("neckshaft")), abs=1e-9)
    assert sides == {"left", "right"}


def test_canal_against_reference_canal_py():
    """oracle/canal.py (points, axis) and the facade's get_transform frame vs the reference's own canal.py:19-124
    (tests/golden/make_canal_golden.py; Line.best_fit there is the published SVD algorithm)."""
    from oracle import canal as ocanal
    G = np.load(os.path.join(GOLDEN, "canal_golden.npz"))
    for c in range(int(G["n"])):
        g = lambda k: G[f"c{c}_{k}"]
        cut = tuple(g("cutoff").tolist())
        pts_obb, pts_ct = ocanal.canal_points(g("centroids"), g("zs"), g("T_obb"), cut)
        np.testing.assert_allclose(pts_ct, g("points_ct"), rtol=0, atol=1e-10)
        np.testing.assert_allclose(xform.transform_pts(pts_ct, g("T_current")), g("points"), rtol=0, atol=1e-9)
        _, ax_ct = ocanal.canal_axis(pts_obb, float(g("z_length")), g("T_obb"), cut)
        np.testing.assert_allclose(ax_ct, g("axis_ct"), rtol=0, atol=1e-9)
        # get_transform (canal.py:88-124) on the axis in the current csys, x from the OBB frame
        ax = xform.transform_pts(ax_ct, g("T_current"))
        z = (ax[0] - ax[1]) / np.linalg.norm(ax[0] - ax[1])
        x = g("T_obb")[:3, :1].flatten().copy()
        x -= z * np.dot(x, z) / np.dot(z, z)
        x /= np.linalg.norm(x)
        y = np.cross(z, x)
        y /= np.linalg.norm(y)
        T = np.r_[np.c_[x, y, z, np.average(ax, axis=0)], np.array([[0, 0, 0, 1.0]])]
        np.testing.assert_allclose(xform.inv_transform(T), g("get_transform"), rtol=0, atol=1e-9)


def test_groove_axis_against_reference():
    """oracle/groove.py groove_axis vs the reference's own DeepGroove.axis (bicipital_groove.py:244-265).  The reference keeps
    the RAW sign of the SVD direction (LAPACK's choice); canonical rule B-4 of the restatement fixes it to +z, so the two end
    points may come in either order -- everything else (fit point, half length = z range / 2, OBB -> CT) must agree."""
    G = np.load(os.path.join(GOLDEN, "canal_golden.npz"))
    for c in range(int(G["n_groove"])):
        g = lambda k: G[f"g{c}_{k}"]
        ax_ct = groove.groove_axis(g("points_obb"), g("T_obb"))
        ref = g("axis_ct")
        if np.abs(ax_ct - ref).max() > 1e-9:
            ref = ref[::-1]
        np.testing.assert_allclose(ax_ct, ref, rtol=0, atol=1e-9)
        # row 0 of the restatement is the +z end in the OBB frame
        ends_obb = xform.transform_pts(ax_ct, g("T_obb"))
        assert ends_obb[0, 2] > ends_obb[1, 2]


def test_te_glue_against_reference():
    """oracle/te.py first_max / far_pair / medial_first vs the reference's own TransEpicondylar.axis (epicondyle.py:29-101) run
    with the shapely results injected (tests/golden/make_te_golden.py): row choice incl. a tie, 2 / 3 / 4 end pieces in shuffled
    order, lift to 3-D, OBB -> CT, medial-first ordering."""
    from oracle import te
    G = np.load(os.path.join(GOLDEN, "te_golden.npz"))
    for c in range(int(G["n"])):
        g = lambda k: G[f"c{c}_{k}"]
        a, b = slices.cutoff_range(len(g("zs")), (0.8, 0.99))
        rects = g("rects")[a:b]
        # utils.major_axis_dist (utils.py:89-97): the longer of the two sides at corner 0
        d = [max(np.hypot(*(r[3] - r[0])), np.hypot(*(r[1] - r[0]))) for r in rects]
        k = te.first_max(d)
        cents = g("cents")[a:b][k]
        i, j = te.far_pair(cents)
        end_pts = np.c_[cents[[i, j]], np.repeat(g("zs")[a:b][k], 2)]
        _, end_ct = te.medial_first(end_pts, g("T_obb"), g("canal"), g("central"))
        np.testing.assert_allclose(end_ct, g("axis_ct"), rtol=0, atol=1e-9)
        np.testing.assert_allclose(g("axis"), g("axis_ct"), rtol=0, atol=0)      # identity Transform in the golden run
