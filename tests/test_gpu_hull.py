"""The hull of SH_STAGE_OBB on the device (k_hull.h, round-based quickhull) against the host quickhull (sh_hull.h) through
the C-ABI: the same triangles, hence -- normals being written canonically by both -- the same record bits that matter and
bit-identical box transforms and landmarks; the device path's fall-back to the host hull for inputs it gives up."""
import os

import numpy as np
import pytest

from conftest import BONES
from shoulder_amd import _lib, synth

pytestmark = pytest.mark.gpu
NAMES = ["humerus_left", "humerus_left_flipped", "humerus_left_trab", "humerus_right"]
OBB_ONLY = _lib.STAGE_OBB


def _engine(mode, weights):
    from shoulder_amd import unet_spec
    from shoulder_amd.engine import Engine
    e = Engine(0)
    e.set_hull_mode(mode)
    assert e.hull_mode == mode
    e.load_rfc()
    e.load_unet(weights, unet_spec.BASE, unet_spec.DEPTH)
    return e


@pytest.fixture(scope="module")
def engines(unet_weights):
    dev, host = _engine("device", unet_weights), _engine("host", unet_weights)
    yield dev, host
    dev.close(); host.close()


def _record(e, B):
    nv, nf, ne = (e.fetch(f"hull.{k}", np.int32, (B,)) for k in ("nv", "nf", "ne"))
    hv = e.fetch("hull.hv", np.float64).reshape(B, 16384, 3)
    nr = e.fetch("hull.normals", np.float64).reshape(B, 32768, 3)
    ed = e.fetch("hull.edges", np.int32).reshape(B, 49152, 4)
    return nv, nf, ne, hv, nr, ed


def _rows(a):
    return a[np.lexsort(tuple(a[:, k] for k in reversed(range(a.shape[1]))))]


def test_fixture_hulls_equal_host_hulls(engines, oracle_bones):
    dev, host = engines
    hs = [oracle_bones(n) for n in NAMES]
    for e in (dev, host):
        e.upload([(h.verts, h.faces) for h in hs])
        e.run(OBB_ONLY, fetch=False)
    B = len(hs)
    rd, rh = _record(dev, B), _record(host, B)
    assert int(dev.fetch("hulld.fail", np.int32, (B,)).max()) == 0
    rounds = dev.fetch("hulld.rounds", np.int32, (B,))
    print("device hull rounds per humerus:", rounds.tolist(), " faces:", rd[1].tolist())
    assert (rounds > 20).all() and (rounds < 400).all()
    for b in range(B):
        nv, nf, ne = int(rd[0][b]), int(rd[1][b]), int(rd[2][b])
        assert (nv, nf, ne) == (int(rh[0][b]), int(rh[1][b]), int(rh[2][b])) and nv - ne + nf == 2
        # same vertex set; same triangles: a triangle is its canonical normal + the set of its three edges, compared through the
        # sorted (normal) rows and the sorted geometric edges
        assert np.array_equal(_rows(rd[3][b][:nv]), _rows(rh[3][b][:nv]))
        assert np.array_equal(_rows(rd[4][b][:nf]), _rows(rh[4][b][:nf]))           # bit for bit: both paths write canonical normals
        def geo_edges(rec):
            hv, ed = rec[3][b], rec[5][b][:ne]
            a, c = hv[ed[:, 0]], hv[ed[:, 1]]
            lo = np.where((a < c).any(axis=1)[:, None] & (np.argmax(a != c, axis=1) >= 0)[:, None], a, a)      # keep orientation-free key below
            key = np.sort(np.stack([a, c], axis=1), axis=1).reshape(ne, 6)
            return _rows(key)
        assert np.array_equal(geo_edges(rd), geo_edges(rh))
        # every edge's two faces are different, valid faces
        ed = rd[5][b][:ne]
        assert (ed[:, 2] != ed[:, 3]).all() and ed[:, 2:].min() >= 0 and ed[:, 2:].max() < nf and ed[:, :2].max() < nv
    np.testing.assert_array_equal(dev.fetch("obb.T_pre", np.float64, (B, 16)), host.fetch("obb.T_pre", np.float64, (B, 16)))


def test_landmarks_identical_on_the_bench_batch(engines, oracle_bones):
    """64 synthetic humeri: every record field the two hull paths can influence is bit-identical."""
    dev, host = engines
    h = oracle_bones("humerus_left")
    T = synth.similarity_transforms(64, h.verts, seed=1234)
    out = []
    for e in (dev, host):
        e.upload([(h.verts, h.faces)])
        e.synth_batch(T)
        out.append(e.run(_lib.STAGE_ALL).copy())
    assert int(dev.fetch("hulld.fail", np.int32, (64,)).max()) == 0
    assert (out[0]["status"] == 0).all()
    for key in ("obb_transform", "z_length", "canal_axis", "te_axis", "groove_axis", "anp_plane_point", "anp_axis_central", "csys", "neck_index", "flipped", "n_anp"):
        np.testing.assert_array_equal(out[0][key], out[1][key], err_msg=key)


def test_proximal_and_streaming(engines, oracle_bones):
    """The cut humerus (a planar cap: hundreds of nearly coplanar points) and the submit / collect path."""
    dev, host = engines
    from oracle.stl import load_stl
    v, f = load_stl(os.path.join(BONES, "proximal_left_cut.stl"))
    res = []
    for e in (dev, host):
        e.set_params(bone_kind=_lib.BONE_PROXIMAL)
        try:
            e.upload([(v, f)])
            res.append(e.run(_lib.STAGE_OBB | _lib.STAGE_FULL | _lib.STAGE_NECK | _lib.STAGE_CANAL).copy())
        finally:
            e.reset_params()
    np.testing.assert_array_equal(res[0]["obb_transform"], res[1]["obb_transform"])
    h = oracle_bones("humerus_left")
    T = synth.similarity_transforms(6, h.verts, seed=9)
    dev.upload([(h.verts, h.faces)]); dev.synth_batch(T)
    ref = dev.run(_lib.STAGE_ALL).copy()
    dev.submit(_lib.STAGE_ALL); dev.submit(_lib.STAGE_ALL)
    a = dev.collect().copy(); b = dev.collect().copy()
    assert a.tobytes() == ref.tobytes() and b.tobytes() == ref.tobytes()


def test_gives_up_cleanly_and_falls_back(engines):
    """An input the device hull does not take -- the midpoint-subdivided humerus: 65 k vertices of which far more than the
    kernel's 8 192 survive the prefilter, thousands of them coplanar with hull facets -- is reported per humerus (reason 40)
    and THAT humerus is re-done by sh_run itself with the host quickhull (redo_given_up): same result as the host-hull engine,
    the other humerus of the batch included; the next batch is back on the device hull."""
    dev, host = engines
    from oracle.stl import load_stl
    from test_gpu_highres import subdivide
    hv, hf = load_stl(os.path.join(BONES, "humerus_right.stl"))
    V, F = subdivide(*load_stl(os.path.join(BONES, "humerus_left.stl")))
    res = []
    for e in (dev, host):
        e.upload([(hv, hf), (V, F)])
        res.append(e.run(OBB_ONLY).copy())
    fail = dev.fetch("hulld.fail", np.int32, (2,))
    assert fail[0] == 0 and fail[1] != 0, fail
    print("device hull gave up with reason", int(fail[1]))
    np.testing.assert_array_equal(res[0]["obb_transform"], res[1]["obb_transform"])
    assert (res[0]["status"] == 0).all()
    dev.upload([(hv, hf)])
    dev.run(OBB_ONLY, fetch=False)
    assert int(dev.fetch("hulld.fail", np.int32, (1,))[0]) == 0 and int(dev.fetch("hulld.rounds", np.int32, (1,))[0]) > 20


def test_given_up_humerus_is_redone_alone_also_with_a_run_in_flight(engines):
    """SH_STAGE_ALL, streaming: two submits in flight on the device-hull engine over a ragged batch whose middle humerus the
    device hull gives up (the midpoint-subdivided mesh).  Each collect re-does only that humerus -- host quickhull, its record
    patched into hull.*, its stages re-run as a window of one behind the other run -- and hands back records that equal the
    host-hull engine's for EVERY humerus; `hulld.skip` then keeps the device hull off it (a third run needs no redo and the
    device hull still ran for the others).  ADVICE r2: the decision comes from the hull's own per-humerus word, not from the
    status word later stages may overwrite."""
    dev, host = engines
    from oracle.stl import load_stl
    from test_gpu_highres import subdivide
    a = load_stl(os.path.join(BONES, "humerus_right.stl"))
    bad = subdivide(*load_stl(os.path.join(BONES, "humerus_left.stl")))
    c = load_stl(os.path.join(BONES, "humerus_left_trab.stl"))
    batch = [a, bad, c]
    for e in (dev, host):
        e.set_params(unet_dtype=_lib.UNET_F32)
        e.upload(batch)
    ref = host.run(_lib.STAGE_ALL).copy()
    assert (ref["status"] == 0).all()
    dev.submit(_lib.STAGE_ALL); dev.submit(_lib.STAGE_ALL)
    r1 = dev.collect().copy(); r2 = dev.collect().copy()
    assert dev.fetch("hulld.skip", np.int32, (3,)).tolist() == [0, 1, 0]
    for r in (r1, r2):
        assert (r["status"] == 0).all()
        assert r.tobytes() == ref.tobytes()
    r3 = dev.run(_lib.STAGE_ALL).copy()                      # skip in force: right the first time
    assert r3.tobytes() == ref.tobytes()
    assert dev.fetch("hulld.fail", np.int32, (3,)).tolist() == [0, 0, 0] and (dev.fetch("hulld.rounds", np.int32, (3,))[[0, 2]] > 20).all()
    dev.upload([a])                                          # a new batch clears the skips
    assert dev.run(_lib.STAGE_ALL)["status"][0] == 0 and dev.fetch("hulld.skip", np.int32, (1,))[0] == 0
