"""A bounded slice of the randomized parity sweeps (tools/parity_sweep*.py, which only the builder ran) under `-m gpu`, so
that the driver sees device-vs-oracle agreement OFF the fixtures too: seeded similarity copies (rotation, translation up to
+-500 mm, isotropic scale 0.85-1.15) of all four reference STLs in ONE ragged, shuffled batch, streamed through
sh_submit / sh_collect with the host-hull overlap on, plus the two multi-component meshes (a detached fragment beside the
shaft, a closed cavity inside the head: inner loops in the sections).  f32 UNet = the exact path: integer decisions equal,
every landmark and metric within 1e-6 mm / 1e-6 deg of the oracle (north star: 1e-4 mm).  ~17 humeri, ~60 s of oracle time."""
import os

import numpy as np
import pytest

from conftest import BONES
from oracle.humerus import OracleHumerus
from shoulder_amd import _lib, synth
from shoulder_amd.stl import load_stl

pytestmark = pytest.mark.gpu
NAMES = ["humerus_left", "humerus_right", "humerus_left_trab", "humerus_left_flipped"]
KEYS = ("canal_axis", "te_axis", "groove_axis", "anp_plane_point", "anp_axis_normal", "anp_axis_central", "csys")
NPER, SEED0 = 4, 20261004       # (seeds the builder's sweeps did not use)


def _check(r, h, tag):
    L, M = h.landmarks(), h.metrics()
    assert int(r["status"]) == 0, tag
    assert float(r["bg_theta"]) == L["bg_theta"] and int(r["n_anp"]) == len(L["anp_points"]) and bool(r["flipped"]) == h.obb["flipped"], tag
    assert int(r["neck_index"]) == h.neck["bkp"] and ("left", "right")[int(r["side"])] == M["side"], tag
    for k in KEYS:
        np.testing.assert_allclose(np.asarray(r[k]).reshape(np.shape(L[k])), L[k], rtol=0, atol=1e-6, err_msg=f"{tag}: {k}")
    np.testing.assert_allclose(r["obb_transform"].reshape(4, 4), L["T_obb"], rtol=0, atol=1e-9, err_msg=tag)
    np.testing.assert_allclose(r["groove_points"].reshape(-1, 3), L["groove_points"], rtol=0, atol=1e-6, err_msg=tag)
    n = int(r["n_anp"])
    np.testing.assert_allclose(r["anp_points"].reshape(-1, 3)[:min(n, 4096)], L["anp_points"][:4096], rtol=0, atol=1e-6, err_msg=tag)
    assert abs(float(r["neckshaft"]) - M["neckshaft"]) < 1e-6 and abs(float(r["retroversion"]) - M["retroversion"]) < 1e-6, tag
    return max(float(np.abs(np.asarray(r[k]).reshape(np.shape(L[k])) - L[k]).max()) for k in KEYS)


def test_similarity_copies_of_all_fixtures_in_one_ragged_streamed_batch(engine, rfc_tables, unet_weights):
    meshes = []
    for bi, name in enumerate(NAMES):
        v, f = load_stl(os.path.join(BONES, name + ".stl"))
        T = synth.similarity_transforms(NPER, v, seed=SEED0 + bi)
        meshes += [(name, synth.apply_similarity(T[i], v), f) for i in range(NPER)]
    order = np.random.default_rng(SEED0).permutation(len(meshes))
    meshes = [meshes[i] for i in order]
    engine.reset_params()
    engine.set_params(unet_dtype=_lib.UNET_F32)
    engine.upload([(mv, mf) for _, mv, mf in meshes])
    engine.set_overlap(True)
    try:
        engine.submit(_lib.STAGE_ALL); first = engine.collect().copy()
        engine.submit(_lib.STAGE_ALL); engine.submit(_lib.STAGE_ALL)            # (the second of these uses hulls prepared during the first)
        second = engine.collect().copy(); third = engine.collect().copy()
    finally:
        engine.set_overlap(False)
    assert first.tobytes() == second.tobytes() == third.tobytes()
    worst = 0.0
    for i, (name, mv, mf) in enumerate(meshes):
        worst = max(worst, _check(third[i], OracleHumerus(mv, mf, rfc_tables, unet_weights, unet_eval="chain"), f"{name} copy at slot {i}"))
    print(f"ragged streamed batch of {len(meshes)}: worst landmark deviation {worst:.2e} mm")


def _box(center_obb, half, Tinv, inward=False):
    c = np.array([[x, y, z] for x in (-1, 1) for y in (-1, 1) for z in (-1, 1)], dtype=np.float64) * half + center_obb
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    t = np.array([tri for a, b, cc, d in quads for tri in ((a, b, cc), (a, cc, d))], dtype=np.int32)
    if inward:
        t = t[:, ::-1]
    return ((np.c_[c, np.ones(8)] @ Tinv.T)[:, :3]).astype(np.float32), t


@pytest.mark.parametrize("case", ["fragment beside the shaft", "cavity inside the head"])
def test_meshes_with_more_than_one_component(engine, oracle_bones, rfc_tables, unet_weights, case):
    h0 = oracle_bones("humerus_left")
    Tinv = np.linalg.inv(h0.T_obb)
    zmax = h0.verts_obb[:, 2].max()
    bv, bf = (_box(np.array([45.0, 3.0, -20.0]), np.array([2.5, 2.0, 3.0]), Tinv) if case.startswith("fragment")
              else _box(np.array([0.0, 0.0, zmax - 22.0]), np.array([3.0, 3.5, 4.0]), Tinv, inward=True))
    mv = np.concatenate([h0.verts, bv]).astype(np.float32)
    mf = np.concatenate([h0.faces, bf + len(h0.verts)]).astype(np.int32)
    engine.reset_params()
    engine.set_params(unet_dtype=_lib.UNET_F32)
    engine.upload([(mv, mf)])
    r = engine.run(_lib.STAGE_ALL)[0]
    h = OracleHumerus(mv, mf, rfc_tables, unet_weights, unet_eval="chain")
    d = _check(r, h, case)
    nl = engine.fetch("prox.nloops", np.int32, (600,))
    print(f"{case}: worst landmark deviation {d:.2e} mm, sections with more than one loop: {int((nl > 1).sum())}")
