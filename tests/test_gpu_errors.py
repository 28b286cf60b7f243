"""Failure behaviour through the C-ABI: bad input is rejected with an error code and message, degenerate
geometry is reported per mesh -- nothing aborts, nothing silently falls back."""
import os

import numpy as np
import pytest

from shoulder_amd import _lib
from shoulder_amd.engine import ShoulderHipError

pytestmark = pytest.mark.gpu


def test_upload_validation(engine, oracle_bones):
    h = oracle_bones("humerus_left")
    bad = h.faces.copy()
    bad[5, 1] = len(h.verts) + 7
    with pytest.raises(ShoulderHipError, match="face index out of range") as e:
        engine.upload([(h.verts, bad)])
    assert e.value.code == -1
    with pytest.raises(ShoulderHipError, match="fewer than 4"):
        engine.upload([(h.verts[:3], h.faces[:1] * 0 + np.array([[0, 1, 2]], dtype=np.int32))])


def test_open_mesh_is_a_geometry_error(engine, oracle_bones):
    """Dropping a band of triangles opens the surface: slices through the hole are not closed loops."""
    h = oracle_bones("humerus_left")
    zc = h.verts[:, 2][h.faces].mean(axis=1)
    keep = ~((zc > np.percentile(zc, 45)) & (zc < np.percentile(zc, 47)) & (h.verts[:, 0][h.faces].mean(axis=1) > np.median(h.verts[:, 0])))
    engine.upload([(h.verts, h.faces[keep]), (h.verts, h.faces)])
    with pytest.raises(ShoulderHipError) as e:
        engine.run(_lib.STAGE_ALL)
    assert e.value.code == -5 and "mesh 0" in str(e.value)


def test_missing_parameters_and_stage_order(oracle_bones):
    from shoulder_amd.engine import Engine
    h = oracle_bones("humerus_left")
    e = Engine(0)
    try:
        with pytest.raises(ShoulderHipError, match="no meshes uploaded"):
            e.run(_lib.STAGE_ALL)
        e.upload([(h.verts, h.faces)])
        with pytest.raises(ShoulderHipError, match="no OBB transform"):
            e.run(_lib.STAGE_FULL)
        e.run(_lib.STAGE_OBB | _lib.STAGE_FULL | _lib.STAGE_NECK | _lib.STAGE_CANAL | _lib.STAGE_PROXIMAL)
        with pytest.raises(ShoulderHipError, match="sh_load_rfc"):
            e.run(_lib.STAGE_GROOVE)
        with pytest.raises(ShoulderHipError, match="330 proximal rows"):
            e.set_params(groove_cutoff=(0.1, 0.9))
    finally:
        e.close()


def test_failed_run_in_a_lane_does_not_wedge_the_other(oracle_bones):
    """Two contexts with UNet turns: a run that ends with a geometry error on one lane is reported by its collect, the
    other lane's runs before and after it are unaffected, and the failing lane works again with a good batch."""
    from conftest import _teacher_weights
    from shoulder_amd import unet_spec
    from shoulder_amd.engine import Engine
    h = oracle_bones("humerus_left")
    zc = h.verts[:, 2][h.faces].mean(axis=1)
    keep = ~((zc > np.percentile(zc, 45)) & (zc < np.percentile(zc, 47)) & (h.verts[:, 0][h.faces].mean(axis=1) > np.median(h.verts[:, 0])))
    lanes = []
    try:
        for _ in range(2):
            e = Engine(0)
            e.load_rfc()
            e.load_unet(_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
            e.set_unet_turns(True)
            lanes.append(e)
        good, bad = lanes
        good.upload([(h.verts, h.faces)])
        bad.upload([(h.verts, h.faces[keep])])
        ref = good.run(_lib.STAGE_ALL).copy()
        good.submit(_lib.STAGE_ALL)
        bad.submit(_lib.STAGE_ALL)
        good_1 = good.collect().copy()
        good.submit(_lib.STAGE_ALL)
        with pytest.raises(ShoulderHipError) as err:
            bad.collect()
        assert err.value.code == -5
        assert good.collect().tobytes() == ref.tobytes() == good_1.tobytes()
        bad.upload([(h.verts, h.faces)])
        bad.submit(_lib.STAGE_ALL)
        good.submit(_lib.STAGE_ALL)
        assert bad.collect().tobytes() == ref.tobytes()
        assert good.collect().tobytes() == ref.tobytes()
    finally:
        for e in lanes:
            e.close()


def test_parameter_ranges(engine, oracle_bones):
    """sh_set_params refuses values the kernels (and the reference) cannot index with: a groove window beyond half a turn used
    to run the local-minimum search off the 512-sample row (a GPU memory fault at 1e9 degrees); NaN, negatives, unknown
    enumerators and cut-offs outside [0, 1] are argument errors.  Legal extremes run."""
    h = oracle_bones("humerus_left")
    engine.upload([(h.verts, h.faces)])
    try:
        for bad in (dict(groove_deg_window=1e9), dict(groove_deg_window=float("nan")), dict(groove_deg_window=-1.0), dict(groove_deg_window=181.0),
                    dict(unet_dtype=7), dict(bone_kind=3), dict(canal_cutoff=(float("nan"), 0.75)), dict(canal_cutoff=(-0.1, 0.75)),
                    dict(canal_cutoff=(0.75, 0.35)), dict(groove_cutoff=(0.1, 0.9))):
            with pytest.raises(ShoulderHipError) as err:
                engine.set_params(**bad)
            assert err.value.code == -1, bad
        for ok in (dict(groove_deg_window=0.0), dict(groove_deg_window=180.0), dict(canal_cutoff=(0.0, 1.0)), dict(canal_cutoff=(0.5, 0.52))):
            engine.set_params(**ok)
            assert engine.run(_lib.STAGE_ALL)["status"][0] == 0, ok
    finally:
        engine.reset_params()


def test_non_finite_coordinates_are_refused(engine, oracle_bones, tmp_path):
    """NaN / inf vertices never reach the kernels: sh_upload_meshes checks the host array, sh_upload_stl flags them while parsing."""
    import bench
    h = oracle_bones("humerus_left")
    for badval in (np.nan, np.inf, -np.inf):
        v = h.verts.copy()
        v[1234, 1] = badval
        with pytest.raises(ShoulderHipError) as err:
            engine.upload([(v, h.faces)])
        assert err.value.code == -1
        p = tmp_path / "bad.stl"
        p.write_bytes(bench.stl_bytes(v, h.faces))
        with pytest.raises(ShoulderHipError) as err:
            engine.upload_stl([os.path.join(os.path.dirname(__file__), "golden", "bones", "humerus_right.stl"), str(p)])
        assert err.value.code == -1 and "NaN" in str(err.value)
    engine.upload([(h.verts, h.faces)])
    assert engine.run(_lib.STAGE_ALL)["status"][0] == 0


def test_malformed_forest_is_refused(tmp_path):
    """The forest walk on the device loops until it meets a leaf, so sh_load_rfc must not accept tables with a cycle or a shared
    subtree; the shipped tables load."""
    from shoulder_amd.engine import Engine
    models = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "shoulder_amd", "models", "rfc_bg3.npz")
    z = dict(np.load(models))
    e = Engine(0)
    try:
        e.load_rfc()
        branch = np.nonzero(z["true_idx"] >= 0)[0]
        bad = dict(z)
        bad["true_idx"] = z["true_idx"].copy()
        bad["true_idx"][branch[5]] = z["roots"][0]                      # a branch points back at its root: a cycle
        p = tmp_path / "cyclic.npz"
        np.savez(p, **bad)
        with pytest.raises(ShoulderHipError) as err:
            e.load_rfc(str(p))
        assert err.value.code == -1 and "forest" in str(err.value)
        bad["true_idx"] = z["true_idx"].copy()
        bad["true_idx"][branch[7]] = len(z["feat"]) + 3                  # out of range
        np.savez(p, **bad)
        with pytest.raises(ShoulderHipError):
            e.load_rfc(str(p))
    finally:
        e.close()


def test_wrapper_argument_checks(engine, oracle_bones):
    """The Python side hands the C-ABI only arrays of the size it will read: a 3x3 'transform' is the reference's ValueError, not an
    out-of-bounds host read; an empty mesh list and a collect() with nothing in flight are errors too."""
    h = oracle_bones("humerus_left")
    engine.upload([(h.verts, h.faces)])
    with pytest.raises(ValueError, match="Invalid transformation matrix shape"):
        engine.mesh_transformed(0, np.zeros((3, 3)))
    with pytest.raises(ValueError, match="Invalid transformation matrix shape"):
        engine.transform_points(np.zeros((4, 3)), np.identity(3))
    with pytest.raises(IndexError):
        engine.mesh_transformed(3, np.identity(4))
    with pytest.raises(ValueError):
        engine.upload([])
    with pytest.raises(ShoulderHipError) as err:
        engine.collect()
    assert err.value.code == -3
    assert engine.transform_points(np.zeros((0, 3)), np.identity(4)).shape == (0, 3)
    assert engine.run(_lib.STAGE_ALL)["status"][0] == 0


def test_rejected_upload_leaves_the_resident_batch_usable(engine, oracle_bones, tmp_path):
    """A rejected sh_upload_meshes / sh_upload_stl must not leave the context half way between two batches (the rejected
    batch's offsets over the old batch's device buffers): the resident batch (here TWO meshes, rejected ones hold ONE) still
    runs and gives the same records, section and mesh transforms as before."""
    import bench
    a, b = oracle_bones("humerus_left"), oracle_bones("humerus_right")
    engine.upload([(a.verts, a.faces), (b.verts, b.faces)])
    before = engine.run(_lib.STAGE_ALL).copy()
    assert (before["status"] == 0).all()
    bad_faces = a.faces.copy(); bad_faces[11, 2] = -4
    nan_verts = a.verts.copy(); nan_verts[77, 0] = np.nan
    p = tmp_path / "nan.stl"
    p.write_bytes(bench.stl_bytes(nan_verts, a.faces))
    for attempt in (lambda: engine.upload([(a.verts, bad_faces)]), lambda: engine.upload([(nan_verts, a.faces)]),
                    lambda: engine.upload([(a.verts[:3], a.faces[:1] * 0 + np.array([[0, 1, 2]], dtype=np.int32))]),
                    lambda: engine.upload_stl([str(p)])):
        with pytest.raises(ShoulderHipError) as err:
            attempt()
        assert err.value.code == -1
        assert engine.B == 2
        after = engine.run(_lib.STAGE_ALL)
        for key in ("obb_transform", "canal_axis", "te_axis", "groove_axis", "anp_axis_central", "csys", "n_anp", "status"):
            np.testing.assert_array_equal(after[key], before[key], err_msg=key)
        assert len(engine.section_plane(1, before["anp_plane_point"][1], before["anp_plane_normal"][1])) > 50
        assert engine.mesh_transformed(1, np.identity(4)).shape == (len(b.verts), 3)


def test_rejected_unet_load_keeps_the_loaded_network(oracle_bones):
    """sh_load_unet with a wrong parameter count (another base / depth) used to swap the layer table before the size check:
    the next forward then read weights past the parameter block.  Now the loaded network stays in force."""
    from conftest import _teacher_weights
    from shoulder_amd import unet_spec
    from shoulder_amd.engine import Engine, unet_pack_order
    h = oracle_bones("humerus_left")
    w = _teacher_weights()
    e = Engine(0)
    try:
        e.load_rfc(); e.load_unet(w, unet_spec.BASE, unet_spec.DEPTH)
        e.upload([(h.verts, h.faces)])
        before = e.run(_lib.STAGE_ALL).copy()
        img = np.random.default_rng(3).random((1, 256, 256), dtype=np.float32)
        lg = e.unet_infer(img)
        for base, depth in ((64, 4), (32, 3), (32, 5)):
            with pytest.raises(ShoulderHipError, match="expected") as err:
                e.load_unet(w, base, depth) if depth == unet_spec.DEPTH else e.load_unet({k: w.get(k, np.zeros(1, np.float32)) for k in unet_pack_order(depth)}, base, depth)
            assert err.value.code == -1
            np.testing.assert_array_equal(e.unet_infer(img), lg)
        bad = dict(w); bad["dec1a_b"] = w["dec1a_b"].copy(); bad["dec1a_b"][3] = np.nan
        with pytest.raises(ShoulderHipError, match="NaN"):
            e.load_unet(bad, unet_spec.BASE, unet_spec.DEPTH)
        after = e.run(_lib.STAGE_ALL)
        for key in ("anp_axis_central", "anp_plane_point", "n_anp", "csys"):
            np.testing.assert_array_equal(after[key], before[key], err_msg=key)
    finally:
        e.close()


def test_proximal_csys_needs_the_anatomic_neck_stage(engine, oracle_bones):
    """a proximal humerus' frame is canal / articular; SH_STAGE_CSYS without SH_STAGE_ANP in the same run would pack a
    stale matrix (and SH_STAGE_APPLY would move the mesh with it) -- refused as an argument error."""
    h = oracle_bones("humerus_left")
    engine.upload([(h.verts, h.faces)])
    engine.set_params(bone_kind=_lib.BONE_PROXIMAL)
    try:
        early = _lib.STAGE_OBB | _lib.STAGE_FULL | _lib.STAGE_NECK | _lib.STAGE_CANAL
        engine.run(early)
        with pytest.raises(ShoulderHipError) as err:
            engine.run(_lib.STAGE_PROXIMAL | _lib.STAGE_GROOVE | _lib.STAGE_CSYS | _lib.STAGE_APPLY)
        assert err.value.code == -1 and "SH_STAGE_ANP" in str(err.value)
    finally:
        engine.set_params(bone_kind=_lib.BONE_HUMERUS)
