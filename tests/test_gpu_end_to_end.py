"""GPU parity, the whole hot path (SH_STAGE_ALL incl. the OBB stage) through the C-ABI vs the oracle:
the reference's four STL fixtures, and BASELINE config 3 (synthetic similarity copies) through the
equivariance property landmarks(T mesh) = T landmarks(mesh)."""
import numpy as np
import pytest

from oracle.stl import load_stl
from shoulder_amd import _lib, synth

pytestmark = pytest.mark.gpu
NAMES = ["humerus_left", "humerus_left_flipped", "humerus_left_trab", "humerus_right"]
MM = 1e-4      # north-star tolerance, mm


@pytest.fixture(scope="module")
def ran(engine, oracle_bones):
    engine.reset_params()      # (whatever an earlier test file left: bone kind, UNet element type, cut-offs)
    hs = [oracle_bones(n) for n in NAMES]
    engine.upload([(h.verts, h.faces) for h in hs])
    lm = engine.run(_lib.STAGE_ALL)
    return hs, lm


def test_obb_frame(engine, ran):
    hs, lm = ran
    for b, h in enumerate(hs):
        assert lm["status"][b] == 0
        assert bool(lm["flipped"][b]) == h.obb["flipped"]                    # head-end decision: exact
        np.testing.assert_allclose(lm["obb_transform"][b][:3, :3], h.T_obb[:3, :3], rtol=0, atol=1e-9)
        np.testing.assert_allclose(lm["obb_transform"][b][:3, 3], h.T_obb[:3, 3], rtol=0, atol=1e-6)
        assert abs(lm["z_length"][b] - h.obb["z_length"]) < 1e-7


def test_all_landmarks(engine, ran):
    hs, lm = ran
    for b, h in enumerate(hs):
        L = h.landmarks()
        assert lm["neck_index"][b] == h.neck["bkp"]
        assert lm["bg_theta"][b] == L["bg_theta"]
        assert lm["n_anp"][b] == len(L["anp_points"])
        for key, ref in (("canal_axis", L["canal_axis"]), ("te_axis", L["te_axis"]), ("groove_axis", L["groove_axis"]),
                         ("groove_points", L["groove_points"]), ("anp_plane_point", L["anp_plane_point"]),
                         ("anp_axis_normal", L["anp_axis_normal"]), ("anp_axis_central", L["anp_axis_central"])):
            np.testing.assert_allclose(lm[key][b], ref, rtol=0, atol=MM, err_msg=f"{NAMES[b]}:{key}")
        K = min(len(L["anp_points"]), 4096)
        np.testing.assert_allclose(lm["anp_points"][b][:K], L["anp_points"][:K], rtol=0, atol=MM)
        np.testing.assert_allclose(lm["csys"][b], L["csys"], rtol=0, atol=1e-6)


def test_batch_apply_csys_and_affine_apply(engine, ran):
    """bone.py:155 for the batch (SH_STAGE_APPLY, part of SH_STAGE_ALL): "verts_csys" = csys[b] * vertices of mesh b, against the
    oracle's transform_pts; then sh_affine_apply (utils.transform_pts for B device-resident point sets, ragged offsets) takes
    the OBB-frame vertices back to CT with the inverse box transforms."""
    from oracle import xform
    hs, lm = ran
    voff = engine.voff
    vc = engine.fetch("verts_csys", np.float64).reshape(-1, 3)[: voff[-1]]
    for b, h in enumerate(hs):
        got = vc[voff[b]:voff[b + 1]]
        np.testing.assert_allclose(got, xform.transform_pts(h.verts.astype(np.float64), lm["csys"][b]), rtol=0, atol=1e-9)     # same matrix: rounding only
        np.testing.assert_allclose(got, xform.transform_pts(h.verts.astype(np.float64), h.landmarks()["csys"]), rtol=0, atol=MM)
    p_in, n_in = engine.buffer_device("verts_obb")
    p_out, n_out = engine.buffer_device("verts_csys")
    assert n_in >= voff[-1] * 24 and n_out >= voff[-1] * 24
    Tinv = np.stack([xform.inv_transform(lm["obb_transform"][b]) for b in range(len(hs))])
    engine.affine_apply(Tinv, p_in, p_out, voff)
    back = engine.fetch("verts_csys", np.float64).reshape(-1, 3)[: voff[-1]]
    for b, h in enumerate(hs):
        np.testing.assert_allclose(back[voff[b]:voff[b + 1]], h.verts.astype(np.float64), rtol=0, atol=1e-9)
    with pytest.raises(ValueError, match="Invalid transformation matrix shape"):
        engine.affine_apply(np.zeros((4, 3, 3)), p_in, p_out, voff)
    from shoulder_amd.engine import ShoulderHipError
    with pytest.raises(ShoulderHipError, match="SH_STAGE_CSYS"):
        engine.run(_lib.STAGE_APPLY | _lib.STAGE_FULL)


def test_left_vs_flipped_file(ran):
    """humerus_left_flipped.stl is humerus_left.stl rotated pi about y (x -> -x, z -> -z)."""
    _, lm = ran
    S = np.diag([-1.0, 1.0, -1.0])
    for key in ("canal_axis", "te_axis", "groove_axis"):
        np.testing.assert_allclose(lm[key][0] @ S, lm[key][1], rtol=0, atol=1e-6)


def test_windowed_run_matches_single_window(engine, oracle_bones):
    """sh_run walks batches larger than 16 in windows (host hull of window k+1 overlaps the device work of
    window k): a mesh must get the same landmarks whichever window it falls in."""
    h = oracle_bones("humerus_left")
    T = synth.similarity_transforms(20, h.verts)
    from conftest import engine_with_env
    with engine_with_env(SHOULDER_WINDOW=8) as ew:      # 20 meshes -> windows of 8 + 8 + 4 (default window: 64)
        ew.upload([(h.verts, h.faces)])
        ew.synth_batch(T)
        big = ew.run(_lib.STAGE_ALL).copy()
    assert (big["status"] == 0).all()
    engine.upload([(h.verts, h.faces)])
    engine.synth_batch(T[12:20])                # the same last 8 meshes in one window
    small = engine.run(_lib.STAGE_ALL)
    for key in ("obb_transform", "canal_axis", "te_axis", "groove_axis", "anp_axis_central", "csys", "anp_plane_point"):
        np.testing.assert_allclose(big[key][12:20], small[key], rtol=0, atol=1e-9, err_msg=key)
    for key in ("neck_index", "n_anp", "n_articular", "flipped", "side"):
        np.testing.assert_array_equal(big[key][12:20], small[key])
    np.testing.assert_array_equal(big["bg_theta"][12:20], small["bg_theta"])


ORACLE_PICKS = (5, 23, 41, 63)      # humeri of the 64-batch that also go through the oracle (~6 s each on one core)
# anatomic-neck bounds per UNet element type, asserted below on the bench's own batch (mm; f32 is the north-star 1e-4):
# the mask boundary moves by single pixels of the 512 x 512 polar image (~0.3 mm) where a 16-bit logit rounds across zero,
# and the reference's plane / ellipse fits and ray casts amplify that
# `deg`: the ANP-derived angles (neckshaft, retroversion), `radius`: radius_curvature in mm -- ADVICE r2: bounded for the 16-bit types too
ANP_MM = {"f32": dict(plane=MM, axes=MM, edge_pts=0, deg=1e-6, radius=1e-6), "f32x": dict(plane=MM, axes=MM, edge_pts=0, deg=1e-6, radius=1e-6),
          "bf16": dict(plane=0.3, axes=1.5, edge_pts=60, deg=2.5, radius=0.05), "f16": dict(plane=0.06, axes=0.4, edge_pts=12, deg=0.6, radius=0.02)}


@pytest.fixture(scope="module")
def oracle_picks(oracle_bones, rfc_tables, unet_weights):
    """Oracle landmarks of ORACLE_PICKS of the bench batch (host-generated float32 vertices = the device generator's, bit for bit)."""
    from conftest import _ensure_chain_lib
    from oracle.humerus import OracleHumerus
    _ensure_chain_lib()
    h = oracle_bones("humerus_left")
    T = synth.similarity_transforms(64, h.verts, seed=1234)
    out = {}
    for b in ORACLE_PICKS:
        o = OracleHumerus(synth.apply_similarity(T[b], h.verts), h.faces, rfc_tables, unet_weights, unet_eval="chain")
        out[b] = (o, o.landmarks(), o.metrics())
    return out


@pytest.mark.parametrize("unet", ["f32", "f32x", "bf16", "f16"])
def test_full_size_batch_equivariance(engine, oracle_bones, oracle_picks, unet):
    """BASELINE configs[2] at full size, in every UNet element type: 64 synthetic humeri (the bench's batch, same seed).
    (1) oracle parity on ORACLE_PICKS: everything that does not read the mask within 1e-4 mm and integer decisions exact in all
    three types; the anatomic-neck landmarks within 1e-4 mm for f32 and within the stated ANP_MM bound for bf16 / f16.
    (2) Size-independent property over all 64: every mesh's landmarks are the similarity transform of the template's."""
    h = oracle_bones("humerus_left")
    B = 64
    T = synth.similarity_transforms(B, h.verts, seed=1234)
    engine.upload([(h.verts, h.faces)])
    engine.synth_batch(np.concatenate([np.identity(4)[None], T[1:]]))
    engine.set_params(unet_dtype={"f32": _lib.UNET_F32, "f32x": _lib.UNET_F32X, "bf16": _lib.UNET_BF16, "f16": _lib.UNET_F16}[unet])
    try:
        lm = engine.run(_lib.STAGE_ALL)
    finally:
        engine.set_params(unet_dtype=_lib.UNET_F32)
    assert (lm["status"] == 0).all()
    bound = ANP_MM[unet]
    seen = dict(plane=0.0, axes=0.0, edge_pts=0, geom=0.0, deg=0.0, radius=0.0)
    for b, (o, L, M) in oracle_picks.items():
        assert ("left", "right")[int(lm["side"][b])] == M["side"]
        seen["deg"] = max(seen["deg"], abs(float(lm["neckshaft"][b]) - M["neckshaft"]), abs(float(lm["retroversion"][b]) - M["retroversion"]))
        seen["radius"] = max(seen["radius"], abs(float(lm["radius_curvature"][b]) - M["radius_curvature"]))
        assert lm["neck_index"][b] == o.neck["bkp"] and bool(lm["flipped"][b]) == o.obb["flipped"]
        assert lm["bg_theta"][b] == L["bg_theta"]
        for key in ("canal_axis", "groove_axis", "groove_points", "te_axis"):
            d = float(np.abs(lm[key][b] - L[key]).max())
            seen["geom"] = max(seen["geom"], d)
            assert d < MM, (unet, b, key, d)
        np.testing.assert_allclose(lm["csys"][b], L["csys"], rtol=0, atol=1e-6, err_msg=f"{unet}:{b}:csys")
        seen["plane"] = max(seen["plane"], float(np.abs(lm["anp_plane_point"][b] - L["anp_plane_point"]).max()))
        seen["axes"] = max(seen["axes"], float(np.abs(lm["anp_axis_normal"][b] - L["anp_axis_normal"]).max()),
                           float(np.abs(lm["anp_axis_central"][b] - L["anp_axis_central"]).max()))
        seen["edge_pts"] = max(seen["edge_pts"], abs(int(lm["n_anp"][b]) - len(L["anp_points"])))
        if unet in ("f32", "f32x"):
            K = min(len(L["anp_points"]), 4096)
            np.testing.assert_allclose(lm["anp_points"][b][:K], L["anp_points"][:K], rtol=0, atol=MM)
    print(f"{unet}: vs oracle on {ORACLE_PICKS}: geometry {seen['geom']:.2e} mm, neck plane point {seen['plane']:.2e} mm, "
          f"neck axes {seen['axes']:.2e} mm, edge-point count differs by <= {seen['edge_pts']}, neck-shaft / retroversion angle {seen['deg']:.2e} deg, "
          f"radius of curvature {seen['radius']:.2e} mm")
    assert seen["plane"] <= bound["plane"] and seen["axes"] <= bound["axes"] and seen["edge_pts"] <= bound["edge_pts"]
    assert seen["deg"] <= bound["deg"] and seen["radius"] <= bound["radius"]
    base = lm[0]
    worst = {}
    for key in ("canal_axis", "te_axis", "groove_axis", "anp_plane_point"):
        dev = []
        for b in range(1, B):
            exp = np.atleast_2d(base[key]) @ T[b][:3, :3].T + T[b][:3, 3]
            dev.append(np.abs(np.atleast_2d(lm[key][b]) - exp).max() / np.cbrt(abs(np.linalg.det(T[b][:3, :3]))))
        worst[key] = max(dev)
    print("max deviation from equivariance (mm, scale-normalised):", {k: round(v, 5) for k, v in worst.items()})
    # vertices are re-rounded to float32 at |coord| up to ~1.5e3 mm (6e-5 mm each) before every stage
    assert worst["canal_axis"] < 2e-2 and worst["groove_axis"] < 5e-2 and worst["te_axis"] < 5e-2
    assert (lm["neck_index"] == base["neck_index"]).all() and (lm["flipped"] == base["flipped"]).all()


def test_synthetic_batch_equivariance(engine, oracle_bones):
    """BASELINE config 3: device-side similarity copies of the template; landmarks must follow."""
    h = oracle_bones("humerus_left")
    B = 6
    T = synth.similarity_transforms(B, h.verts)
    T[0] = np.identity(4)
    engine.upload([(h.verts, h.faces)])
    engine.synth_batch(T)
    lm = engine.run(_lib.STAGE_ALL)
    assert (lm["status"] == 0).all()
    base = lm[0]
    ref = h.landmarks()
    np.testing.assert_allclose(base["canal_axis"], ref["canal_axis"], rtol=0, atol=MM)
    # synthetic vertices are re-rounded to float32 after the transform (|coord| up to ~1500 mm ->
    # 6e-5 mm per vertex), so copies are compared at 5e-3 mm / exact integer decisions
    for b in range(1, B):
        tf = lambda p: p @ T[b][:3, :3].T + T[b][:3, 3]
        assert lm["neck_index"][b] == base["neck_index"]
        for key in ("canal_axis", "te_axis", "groove_axis"):
            np.testing.assert_allclose(lm[key][b], tf(base[key]), rtol=0, atol=5e-3 * 3, err_msg=key)
    # the device-generated vertices themselves: float64 arithmetic, float32 storage
    v = engine.fetch("verts", np.float32, (B, len(h.verts), 3))
    for b in range(B):
        np.testing.assert_array_equal(v[b], synth.apply_similarity(T[b], h.verts))


def test_forked_trans_epicondylar_part_gives_the_same_records(engine, oracle_bones):
    """The rectangles of the distal rows and the ends of the widest one (k_te_rows, k_te_ends: they need the distal set only) run in
    front of the UNet pass by default, on the side stream beside the proximal set, the groove and the UNet pass with SHOULDER_TE_EARLY=1,
    behind the UNet with =0 (switches of a context, read when it is created); k_te_orient (medial end first: needs the head's central
    axis) follows in front of the record.  Same kernels on the same inputs: the records are the sequential run's bit for bit, run after
    run (the second run of a batch is the first that forks: the overflow tier is known to be idle by then)."""
    from conftest import engine_with_env
    h = oracle_bones("humerus_left")
    B = 24
    T = synth.similarity_transforms(B, h.verts, seed=5)
    with engine_with_env(SHOULDER_TE_EARLY=0) as e0:      # behind the UNet (the order of the reference's accessors)
        e0.upload([(h.verts, h.faces)])
        e0.synth_batch(T)
        e0.run(_lib.STAGE_ALL)
        a = e0.run(_lib.STAGE_ALL).copy()
    engine.reset_params()
    engine.upload([(h.verts, h.faces)])
    engine.synth_batch(T)
    for _ in range(3):      # in the chain in front of the UNet (the default)
        b = engine.run(_lib.STAGE_ALL)
        assert (b["status"] == 0).all() and b.tobytes() == a.tobytes()
    with engine_with_env(SHOULDER_TE_EARLY=1) as e1:      # forked beside the UNet
        e1.upload([(h.verts, h.faces)])
        e1.synth_batch(T)
        for _ in range(3):
            b = e1.run(_lib.STAGE_ALL)
            assert (b["status"] == 0).all() and b.tobytes() == a.tobytes()


def test_overlapped_hulls_identical_and_invalidated(engine, oracle_bones):
    """sh_set_overlap: hulls prepared in the background during run k give run k+1 bit-identical records, and a new
    batch (other transforms) voids them."""
    h = oracle_bones("humerus_left")
    verts, faces = h.verts, h.faces
    B = 6
    T1 = synth.similarity_transforms(B, verts, seed=7)
    T2 = synth.similarity_transforms(B, verts, seed=8)
    def batch(T):      # sh_synth_batch copies mesh 0 of the CURRENT batch: start from the template every time
        engine.upload([(verts, faces)])
        engine.synth_batch(T)
    batch(T1)
    ref1 = engine.run(_lib.STAGE_ALL).copy()
    batch(T2)
    ref2 = engine.run(_lib.STAGE_ALL).copy()
    batch(T1)
    engine.set_overlap(True)
    try:
        a = engine.run(_lib.STAGE_ALL).copy()       # inline hulls, prepares the next run
        b = engine.run(_lib.STAGE_ALL).copy()       # uses the prepared hulls
        c = engine.run(_lib.STAGE_ALL).copy()
        batch(T2)                                   # prepared hulls belong to the old batch
        d = engine.run(_lib.STAGE_ALL).copy()
        e = engine.run(_lib.STAGE_ALL).copy()
    finally:
        engine.set_overlap(False)
        engine.discard_prepared()
    for r in (a, b, c):
        assert r.tobytes() == ref1.tobytes()
    for r in (d, e):
        assert r.tobytes() == ref2.tobytes()


def test_submit_collect_pipeline(engine, oracle_bones):
    """sh_submit / sh_collect: two runs in flight (with and without prepared hulls) return the same records as sh_run;
    a third submit without a collect and sh_run with runs in flight are refused."""
    h = oracle_bones("humerus_left")
    B = 5
    T = synth.similarity_transforms(B, h.verts, seed=3)
    engine.upload([(h.verts, h.faces)])
    engine.synth_batch(T)
    ref = engine.run(_lib.STAGE_ALL).copy()
    for overlap in (False, True):
        engine.set_overlap(overlap)
        try:
            engine.submit(_lib.STAGE_ALL)
            engine.submit(_lib.STAGE_ALL)
            with pytest.raises(Exception):
                engine.submit(_lib.STAGE_ALL)
            with pytest.raises(Exception):
                engine.run(_lib.STAGE_ALL)
            a = engine.collect().copy()
            engine.submit(_lib.STAGE_ALL)
            b = engine.collect().copy()
            c = engine.collect().copy()
        finally:
            engine.set_overlap(False)
            engine.discard_prepared()
        for r in (a, b, c):
            assert r.tobytes() == ref.tobytes()


def test_two_contexts_on_one_device(engine, oracle_bones):
    """Lanes (sh_set_unet_turns): two contexts on one device, runs interleaved through sh_submit / sh_collect, UNet passes
    chained by events; every run returns the records a single context returns, with different batches per lane."""
    from shoulder_amd import unet_spec
    from shoulder_amd.engine import Engine
    from conftest import _teacher_weights
    h = oracle_bones("humerus_left")
    B = 4
    Ts = [synth.similarity_transforms(B, h.verts, seed=s) for s in (5, 6)]
    refs = []
    for T in Ts:
        engine.upload([(h.verts, h.faces)])
        engine.synth_batch(T)
        refs.append(engine.run(_lib.STAGE_ALL).copy())
    assert refs[0].tobytes() != refs[1].tobytes()
    lanes = []
    try:
        for T in Ts:
            e = Engine(0)
            e.load_rfc()
            e.load_unet(_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
            e.upload([(h.verts, h.faces)])
            e.synth_batch(T)
            e.set_unet_turns(True)
            e.set_overlap(True)
            lanes.append(e)
        got = [[], []]
        pend = []
        for s in range(6):
            if len(pend) >= 2:
                k = pend.pop(0)
                got[k].append(lanes[k].collect().copy())
            lanes[s % 2].submit(_lib.STAGE_ALL)
            pend.append(s % 2)
        for k in pend:
            got[k].append(lanes[k].collect().copy())
        for k in range(2):
            assert len(got[k]) == 3
            for r in got[k]:
                assert r.tobytes() == refs[k].tobytes()
        lanes[0].set_unet_turns(False)                 # leaving the chain is allowed at any time
        assert lanes[0].run(_lib.STAGE_ALL).tobytes() == refs[0].tobytes()
        assert lanes[1].run(_lib.STAGE_ALL).tobytes() == refs[1].tobytes()
    finally:
        for e in lanes:
            e.close()


def test_contexts_driven_from_different_host_threads(oracle_bones):
    """Contexts are independent: two engines driven concurrently from two Python threads (ctypes releases the GIL; one with UNet
    turns, one without) return what they return alone."""
    import threading
    from conftest import _teacher_weights
    from shoulder_amd import unet_spec
    from shoulder_amd.engine import Engine
    h = oracle_bones("humerus_left")
    engs, refs, bad = [], [], []
    try:
        for seed, turns in ((41, True), (42, False)):
            e = Engine(0)
            e.load_rfc()
            e.load_unet(_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
            e.set_params(unet_dtype=_lib.UNET_BF16)
            e.upload([(h.verts, h.faces)])
            e.synth_batch(synth.similarity_transforms(6, h.verts, seed=seed))
            e.set_overlap(True)
            e.set_unet_turns(turns)
            engs.append(e)
            refs.append(e.run(_lib.STAGE_ALL).copy())

        def worker(i):
            for k in range(12):
                if k % 3 == 0:
                    out = engs[i].run(_lib.STAGE_ALL)
                else:
                    engs[i].submit(_lib.STAGE_ALL)
                    out = engs[i].collect()
                if out.tobytes() != refs[i].tobytes():
                    bad.append((i, k))
        ths = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        assert bad == []
    finally:
        for e in engs:
            e.close()


def test_packed_records_carry_the_full_records_fields(engine, oracle_bones):
    """the wire format of sh_set_record_rows(R) -- 8 680 + 24 R bytes instead of 104 KB per humerus -- holds
    every field of the full record and the first min(n_anp, R) anatomic-neck rows; sh_anp_points returns all of them; the same
    through sh_submit / sh_collect and into a device buffer (a gather's send buffer)."""
    hs = [oracle_bones(n) for n in ("humerus_left", "humerus_right", "humerus_left_trab")]
    engine.reset_params()
    engine.set_params(unet_dtype=_lib.UNET_F32)
    engine.upload([(h.verts, h.faces) for h in hs])
    try:
        engine.set_record_rows(0)
        full = engine.run(_lib.STAGE_ALL).copy()
        n = full["n_anp"]
        assert (full["status"] == 0).all() and (n > 600).all()
        for R in (2560, 512):
            engine.set_record_rows(R)
            pk = engine.run(_lib.STAGE_ALL).copy()
            assert pk.dtype == _lib.record_dtype(R) and pk.dtype.itemsize == 8680 + 24 * R
            for name in full.dtype.names:
                if name != "anp_points":
                    np.testing.assert_array_equal(pk[name], full[name], err_msg=name)
            for b in range(3):
                k = min(int(n[b]), R)
                np.testing.assert_array_equal(pk["anp_points"][b, :k], full["anp_points"][b, :k])
                assert not pk["anp_points"][b, k:].any()
                np.testing.assert_array_equal(engine.anp_points(b), full["anp_points"][b, : int(n[b])])
            engine.submit(_lib.STAGE_ALL)                                     # page-locked host view
            assert engine.collect().tobytes() == pk.tobytes()
            # device memory as `out` (the send buffer of a gather): a scratch buffer of the engine that the run is done with when the
            # records are written (the network's input image)
            ptr, nbytes = engine.buffer_device("anp.image")
            assert nbytes >= 3 * pk.dtype.itemsize
            engine.submit(_lib.STAGE_ALL, fetch=False, out_ptr=ptr)
            engine.collect()
            assert engine.fetch("anp.image", np.uint8, (3 * pk.dtype.itemsize,)).tobytes() == pk.tobytes()
        assert (n > 512).all() and (n > 2560).any() and (n <= 2560).any()     # 512 rows cut every list, 2 560 one of the three
    finally:
        engine.set_record_rows(0)

