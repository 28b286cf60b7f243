"""GPU parity, slice layer + neck + canal: HIP path (through the C-ABI) vs the oracle on the
reference's STL fixtures, with the oracle's OBB transform injected so that this stage is judged
on its own.  Integer results exact; coordinates to 1e-9 mm (budget: 1e-4 mm)."""
import numpy as np
import pytest

from shoulder_amd import _lib

pytestmark = pytest.mark.gpu
NAMES = ["humerus_left", "humerus_left_trab", "humerus_right"]
TOL = 1e-9


@pytest.fixture(scope="module")
def ran(engine, oracle_bones):
    engine.reset_params()      # (whatever an earlier test file left: bone kind, UNet element type, cut-offs)
    hs = [oracle_bones(n) for n in NAMES]
    engine.upload([(h.verts, h.faces) for h in hs])
    engine.store("obb_transform", np.stack([h.T_obb for h in hs]))
    engine.set_keep_products(True)      # every plane's resampled contour and polar rows (a production run writes the rows its later stages read)
    engine.run(_lib.STAGE_FULL | _lib.STAGE_DISTAL | _lib.STAGE_NECK | _lib.STAGE_CANAL | _lib.STAGE_PROXIMAL, fetch=False)
    engine.set_keep_products(False)
    return hs


def test_verts_obb_and_bounds(engine, ran):
    vo = engine.fetch("verts_obb", np.float64).reshape(-1, 3)
    zb = engine.fetch("z_bounds", np.float64, (len(ran), 2))
    for b, h in enumerate(ran):
        got = vo[engine.voff[b]:engine.voff[b + 1]]
        np.testing.assert_allclose(got, h.verts_obb, rtol=0, atol=1e-10)
        assert abs(zb[b, 0] - h.verts_obb[:, 2].min()) < 1e-10 and abs(zb[b, 1] - h.verts_obb[:, 2].max()) < 1e-10


@pytest.mark.parametrize("pfx,attr,N", [("full", "full", 200), ("distal", "distal", 200), ("prox", "proximal", 600)])
def test_slice_sets(engine, ran, pfx, attr, N):
    B = len(ran)
    zs = engine.fetch(pfx + ".zs", np.float64, (B, N))
    cen = engine.fetch(pfx + ".centroids", np.float64, (B, N, 2))
    areas = engine.fetch(pfx + ".areas", np.float64, (B, N))
    nl = engine.fetch(pfx + ".nloops", np.int32, (B, N))
    cnt = engine.fetch(pfx + ".seg_count", np.int32, (B, N))
    ring_n = engine.fetch(pfx + ".ring_n", np.int32, (B, N))
    for b, h in enumerate(ran):
        s = getattr(h, attr)
        np.testing.assert_allclose(zs[b], s.zs_all, rtol=0, atol=1e-12)
        np.testing.assert_array_equal(nl[b], s.n_loops)                                   # loop count: exact
        np.testing.assert_array_equal(cnt[b], [sum(len(r) - 1 for r in rings) for rings in s.loops])   # crossing triangles: exact
        np.testing.assert_array_equal(ring_n[b], [len(r) - 1 for r in s.largest])
        np.testing.assert_allclose(cen[b], s.centroids_all, rtol=0, atol=TOL)
        np.testing.assert_allclose(areas[b], s.areas1_all, rtol=1e-12, atol=1e-9)


@pytest.mark.parametrize("pfx,attr,N", [("distal", "distal", 200), ("prox", "proximal", 600)])
def test_rings_canonical(engine, ran, pfx, attr, N):
    B = len(ran)
    ring = engine.fetch(pfx + ".ring", np.float64, (B, N, 1025, 2))
    for b, h in enumerate(ran):
        s = getattr(h, attr)
        for k in range(0, N, 7):
            r = s.largest[k]
            np.testing.assert_allclose(ring[b, k, :len(r)], r, rtol=0, atol=TOL)          # same start, same direction


def test_resample_and_polar(engine, ran):
    B = len(ran)
    ixy = engine.fetch("prox.ixy", np.float64, (B, 600, 2, 512))
    its = engine.fetch("prox.itr_start", np.float64, (B, 600, 2, 512))
    itc = engine.fetch("prox.itr_centered_start", np.float64, (B, 600, 2, 512))
    for b, h in enumerate(ran):
        p = h.proximal
        np.testing.assert_allclose(ixy[b], p.ixy_all, rtol=0, atol=TOL)
        for got, exp in ((its[b], p.itr_start_all), (itc[b], p.itr_centered_start_all)):
            # theta rows: the roll index (argmin theta) is an integer decision -> must match exactly
            np.testing.assert_array_equal(np.argmin(np.abs(got[:, 0, :] - exp[:, 0, :1]), axis=1), 0)
            np.testing.assert_allclose(got, exp, rtol=0, atol=TOL)


def test_neck_and_canal(engine, ran):
    B = len(ran)
    nz = engine.fetch("neck_z", np.float64, (B,))
    ni = engine.fetch("neck_index", np.int32, (B,))
    ax = engine.fetch("canal.axis_ct", np.float64, (B, 2, 3))
    pts = engine.fetch("canal.points_obb", np.float64, (B, 200, 3))[:, :80]      # capacity 200 rows per humerus, 80 used with the default cut-offs
    for b, h in enumerate(ran):
        assert ni[b] == h.neck["bkp"]
        assert nz[b] == pytest.approx(h.neck["neck_z"], abs=1e-12)
        np.testing.assert_allclose(pts[b], h.canal["points_obb"], rtol=0, atol=TOL)
        np.testing.assert_allclose(ax[b], h.canal["axis_ct"], rtol=0, atol=1e-8)


def test_merged_slice_sets_equal_separate_launches(engine, oracle_bones):
    """Round 4: full + distal and neck contour + proximal go through the set launches together (one plane-height launch, one pass
    over the mesh, one join grid per pair).  Against one launch group per set (SHOULDER_SLICE_MERGE=0): the same records and the
    same slice-layer buffers byte for byte, for a batch and for a single humerus (which keeps the distal set on its side stream)."""
    from conftest import engine_with_env
    names = ["humerus_left", "humerus_left_flipped", "humerus_left_trab", "humerus_right"]
    hs = [oracle_bones(n) for n in names]
    engine.reset_params()
    engine.set_params(unet_dtype=_lib.UNET_BF16)
    with engine_with_env(SHOULDER_SLICE_MERGE=0) as e_sep:      # (a switch of the context, read when it is created)
      e_sep.set_params(unet_dtype=_lib.UNET_BF16)
      try:
        for meshes in ([(h.verts, h.faces) for h in hs] * 5, [(hs[2].verts, hs[2].faces)]):      # 20 humeri (> 16: merged), one humerus
            B = len(meshes)
            engine.upload(meshes)
            e_sep.upload(meshes)
            out = {}
            for mode in ("1", "0", "1"):
                eng_ = engine if mode == "1" else e_sep
                lm = eng_.run(_lib.STAGE_ALL).copy()
                bufs = {k: eng_.fetch(k, dt, shp).copy() for k, dt, shp in (
                    ("full.areas", np.float64, (B, 200)), ("full.centroids", np.float64, (B, 200, 2)), ("distal.ring_n", np.int32, (B, 200)),
                    ("distal.ring", np.float64, (B, 200, 1025, 2)), ("prox.seg_count", np.int32, (B, 600)), ("prox.itr_start", np.float64, (B, 600, 2, 512)),
                    ("neckc.ring_n", np.int32, (B, 1)), ("neckc.centroids", np.float64, (B, 1, 2)))}
                if mode in out:
                    assert out[mode][0].tobytes() == lm.tobytes()
                out[mode] = (lm, bufs)
            assert (out["1"][0]["status"] == 0).all()
            assert out["1"][0].tobytes() == out["0"][0].tobytes()
            for k in out["1"][1]:
                rn = out["1"][1].get("distal.ring_n") if k == "distal.ring" else None
                if k == "prox.itr_start":      # (a run writes the rows its later stages read: from plane 88 on)
                    np.testing.assert_array_equal(out["1"][1][k][:, 88:], out["0"][1][k][:, 88:], err_msg=k)
                elif rn is None:
                    np.testing.assert_array_equal(out["1"][1][k], out["0"][1][k], err_msg=k)
                else:      # (ring slots behind a ring's end keep whatever an earlier run left there)
                    for b in range(B):
                        for p in range(0, 200, 9):
                            n = int(rn[b, p]) + 1
                            np.testing.assert_array_equal(out["1"][1][k][b, p, :n], out["0"][1][k][b, p, :n])
      finally:
        engine.set_params(unet_dtype=_lib.UNET_F32)


def test_a_run_writes_the_polar_rows_its_stages_read(engine, oracle_bones):
    """k_resample_polar (RsWant): by default the polar rows about the origin leave the kernel from plane 88 on, the centred ones inside
    the groove's cut-off range, the resampled contour not at all -- bit for bit the rows of a run that keeps everything, and the same
    records.  SH_STAGE_GROOVE alone after the cut-off moved is refused (its rows were not written)."""
    from shoulder_amd.engine import ShoulderHipError
    h = oracle_bones(NAMES[0])
    engine.reset_params()
    engine.upload([(h.verts, h.faces)])
    engine.set_keep_products(True)
    lm_all = engine.run(_lib.STAGE_ALL).copy()
    keep = {k: engine.fetch(k, np.float64, (1, 600, 2, 512)).copy() for k in ("prox.ixy", "prox.itr_start", "prox.itr_centered_start")}
    engine.set_keep_products(False)
    for k in keep:
        engine.store(k, np.full((1, 600, 2, 512), -7.0))
    lm = engine.run(_lib.STAGE_ALL).copy()
    assert lm.tobytes() == lm_all.tobytes()
    got = {k: engine.fetch(k, np.float64, (1, 600, 2, 512)) for k in keep}
    assert (got["prox.ixy"] == -7.0).all()
    assert (got["prox.itr_start"][:, :88] == -7.0).all()
    np.testing.assert_array_equal(got["prox.itr_start"][:, 88:], keep["prox.itr_start"][:, 88:])
    cs = got["prox.itr_centered_start"][0]
    written = np.flatnonzero((cs != -7.0).any(axis=(1, 2)))
    assert len(written) == 330 and written[-1] - written[0] == 329
    np.testing.assert_array_equal(cs[written], keep["prox.itr_centered_start"][0][written])
    # another cut-off range: the groove stage alone has no rows to read
    engine.set_params(groove_cutoff=(0.3, 0.85))
    with pytest.raises(ShoulderHipError, match="SH_STAGE_PROXIMAL"):
        engine.run(_lib.STAGE_GROOVE)
    engine.run(_lib.STAGE_PROXIMAL | _lib.STAGE_GROOVE)
    engine.reset_params()
