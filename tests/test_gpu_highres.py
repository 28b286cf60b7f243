"""Meshes denser than the fixtures: one midpoint subdivision of humerus_left.stl (4x the triangles) doubles the crossing
segments per plane (~300-650), which takes the sections through the large-capacity instantiations (k_slice_link_large,
k_resample_polar_large, k_te_rows<1024>) that the fixture meshes never reach; two subdivisions exceed the 1024-segment
capacity and must end in SH_ERR_CAPACITY for that humerus, not in a wrong answer."""
import os

import numpy as np
import pytest

from conftest import BONES
from oracle.humerus import OracleHumerus
from shoulder_amd import _lib
from shoulder_amd.engine import ShoulderHipError
from shoulder_amd.stl import load_stl

pytestmark = pytest.mark.gpu


def subdivide(v, f):
    """every triangle -> 4 (edge midpoints shared between neighbours); float32 coordinates like an STL"""
    v = v.astype(np.float64)
    e = np.sort(np.concatenate([f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]]), axis=1)
    ue, inv = np.unique(e, axis=0, return_inverse=True)
    mid = (0.5 * (v[ue[:, 0]] + v[ue[:, 1]]))
    m = len(v) + inv.reshape(3, -1)                                  # midpoint ids of edges 01, 12, 20 per face
    a, b, c = f[:, 0], f[:, 1], f[:, 2]
    nf = np.concatenate([np.c_[a, m[0], m[2]], np.c_[m[0], b, m[1]], np.c_[m[2], m[1], c], np.c_[m[0], m[1], m[2]]])
    return np.concatenate([v, mid]).astype(np.float32), nf.astype(np.int32)


def test_large_capacity_tier_matches_oracle(engine, rfc_tables, unet_weights):
    v, f = load_stl(os.path.join(BONES, "humerus_left.stl"))
    v2, f2 = subdivide(v, f)
    assert len(f2) == 4 * len(f)
    engine.set_params(unet_dtype=_lib.UNET_F32)
    engine.upload([(v2, f2), (v, f)])                                # ragged: the dense mesh beside the original
    lm = engine.run(_lib.STAGE_ALL)
    cnt = engine.fetch("prox.seg_count", np.int32, (2, 600))
    assert cnt[0].max() > 384 and cnt[1].max() <= 384                 # the dense mesh really is in the large tier
    assert (lm["status"] == 0).all()
    h = OracleHumerus(v2, f2, rfc_tables, unet_weights, unet_eval="chain")
    L = h.landmarks()
    r = lm[0]
    assert bool(r["flipped"]) == h.obb["flipped"] and float(r["bg_theta"]) == L["bg_theta"] and int(r["n_anp"]) == len(L["anp_points"])
    for k in ("canal_axis", "te_axis", "groove_axis", "anp_plane_point", "anp_axis_normal", "anp_axis_central", "csys"):
        np.testing.assert_allclose(np.asarray(r[k]).reshape(np.shape(L[k])), L[k], rtol=0, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(r["anp_points"].reshape(-1, 3)[: int(r["n_anp"])], L["anp_points"], rtol=0, atol=1e-6)


def test_over_capacity_is_an_error(engine):
    v, f = load_stl(os.path.join(BONES, "humerus_left.stl"))
    v3, f3 = subdivide(*subdivide(v, f))
    engine.upload([(v3, f3)])
    with pytest.raises(ShoulderHipError) as err:
        engine.run(_lib.STAGE_ALL)
    assert err.value.code == -4 and "mesh 0" in str(err.value)


def test_slice_capacity_overflow_is_an_error(engine, oracle_bones):
    """The 16x mesh past the hull stage (box frame injected): ~1 300 crossing segments per plane exceed SH_MAXSEG = 1024; the slice
    stage must flag the humerus (SH_ERR_CAPACITY), and the engine must still work afterwards."""
    h = oracle_bones("humerus_left")
    v3, f3 = subdivide(*subdivide(h.verts, h.faces))
    engine.upload([(v3, f3)])
    engine.store("obb_transform", h.T_obb[None])
    with pytest.raises(ShoulderHipError) as err:
        engine.run(_lib.STAGE_FULL | _lib.STAGE_PROXIMAL | _lib.STAGE_NECK)
    assert err.value.code == -4
    engine.upload([(h.verts, h.faces)])
    assert engine.run(_lib.STAGE_ALL)["status"][0] == 0


def test_plane_cuts_of_the_dense_mesh(engine):
    """sh_slice_mesh_planes on the 130 k-triangle mesh, 16 planes in one pass: bit-identical to oracle/clip.py (the midpoints make
    many vertices lie within the 1e-8 on-plane tolerance of axis-aligned planes: the sign-0 branches get exercised)."""
    from oracle import clip
    v, f = load_stl(os.path.join(BONES, "humerus_left.stl"))
    v2, f2 = subdivide(v, f)
    v2 = v2.astype(np.float64)
    rng = np.random.default_rng(31)
    P = 16
    normals = rng.normal(size=(P, 3))
    normals[:3] = np.identity(3)                                   # axis-aligned planes through existing vertices
    origins = v2[rng.integers(0, len(v2), P)].copy()
    origins[3:] += rng.normal(scale=1.0, size=(P - 3, 3))
    got = engine.slice_mesh_planes(v2, f2, origins, normals, edges=True)
    for (gv, gf, ge), o, n in zip(got, origins, normals):
        wv, wf, we = clip.slice_plane(v2, f2, o, n)
        assert gv.shape == wv.shape and np.array_equal(gv.view(np.int64), wv.view(np.int64))
        assert np.array_equal(gf, wf) and np.array_equal(ge, we)


def test_stl_ingest_of_the_dense_mesh(engine, tmp_path):
    """sh_upload_stl on a 130 k-triangle file beside a fixture file (ragged, 4x larger hash table): vertices and faces bit-identical
    to the host loader."""
    import bench
    v, f = load_stl(os.path.join(BONES, "humerus_left.stl"))
    v2, f2 = subdivide(v, f)
    p = tmp_path / "dense.stl"
    p.write_bytes(bench.stl_bytes(v2, f2))
    hv, hf = load_stl(str(p))
    assert len(hf) == len(f2) and len(hv) == len(v2)
    engine.upload_stl([str(p), os.path.join(BONES, "humerus_right.stl")])
    gv = engine.fetch("verts", np.float32, (int(engine.voff[-1]), 3))
    gf = engine.fetch("faces", np.int32, (int(engine.foff[-1]), 3))
    assert np.array_equal(gv[: engine.voff[1]].view(np.uint32), hv.view(np.uint32)) and np.array_equal(gf[: engine.foff[1]], hf)
    rv, rf = load_stl(os.path.join(BONES, "humerus_right.stl"))
    assert np.array_equal(gv[engine.voff[1]:].view(np.uint32), rv.view(np.uint32)) and np.array_equal(gf[engine.foff[1]:], rf)
    assert (engine.run(_lib.STAGE_ALL)["status"] == 0).all()
