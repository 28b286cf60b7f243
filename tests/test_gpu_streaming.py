"""The staging side of the mesh slot (sh_stage_meshes / sh_stage_stl / sh_commit_staged): a stream of DIFFERENT batches through two
lanes, each batch staged while the lane's previous run executes, gives the records of the synchronous path (sh_upload_* + sh_run)
byte for byte -- the reference's unit of work is a new STL (mesh.py:22-27, bone.py:110-131).  VERDICT r3 item 2."""
import os

import numpy as np
import pytest

from conftest import BONES
from shoulder_amd import _lib, synth, unet_spec
from shoulder_amd.engine import Engine, ShoulderHipError
from shoulder_amd.stl import load_stl

pytestmark = pytest.mark.gpu
NAMES = ["humerus_left", "humerus_left_flipped", "humerus_left_trab", "humerus_right"]


def stl_image(verts, faces):
    import struct
    tri = np.asarray(verts, dtype="<f4")[np.asarray(faces)]
    rec = np.zeros(len(tri), dtype=np.dtype([("n", "<f4", (3,)), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    rec["v"] = tri
    return b"\0" * 80 + struct.pack("<I", len(tri)) + rec.tobytes()


@pytest.fixture(scope="module")
def batches():
    """six different ragged batches of five humeri: similarity copies (seeded) of the four fixture meshes, shuffled"""
    base = [load_stl(os.path.join(BONES, n + ".stl")) for n in NAMES]
    rng = np.random.default_rng(77)
    out = []
    for k in range(6):
        pick = rng.integers(0, 4, 5)
        T = synth.similarity_transforms(5, base[0][0], seed=500 + k)
        out.append([(synth.apply_similarity(T[i], base[j][0]), base[j][1]) for i, j in enumerate(pick)])
    return out


def make_lane(unet_weights):
    e = Engine(0)
    e.load_rfc()
    e.load_unet(unet_weights, unet_spec.BASE, unet_spec.DEPTH)
    e.set_params(unet_dtype=_lib.UNET_BF16)
    e.set_hull_mode("host")
    return e


@pytest.mark.parametrize("kind", ["arrays", "stl"])
def test_streamed_batches_equal_the_synchronous_path(batches, unet_weights, kind):
    ref_eng = make_lane(unet_weights)
    lanes = [make_lane(unet_weights), make_lane(unet_weights)]
    try:
        for e in lanes:
            e.set_unet_turns(True)
        feed = [[stl_image(v, f) for v, f in b] for b in batches] if kind == "stl" else [Engine.pack_meshes(b) for b in batches]
        want = []
        for k, b in enumerate(batches):                     # synchronous path: hand over, run, wait
            if kind == "stl":
                ref_eng.upload_stl(feed[k])
            else:
                ref_eng.upload(b)
            want.append(ref_eng.run(_lib.STAGE_ALL).copy())
            assert (want[-1]["status"] == 0).all()
        stage = lambda e, k: (e.stage_stl if kind == "stl" else e.stage)(feed[k])
        got = [None] * len(batches)
        pend = []
        for s in range(len(batches)):
            e = lanes[s % 2]
            if len(pend) >= 2:
                k0, e0 = pend.pop(0)
                got[k0] = e0.collect().copy()
            if not e.staged:
                stage(e, s)
            e.commit_staged()
            e.submit(_lib.STAGE_ALL)
            pend.append((s, e))
            if s + 2 < len(batches):
                stage(e, s + 2)                             # beside the run just submitted
        for k0, e0 in pend:
            got[k0] = e0.collect().copy()
        for k in range(len(batches)):
            assert got[k].tobytes() == want[k].tobytes(), f"batch {k}"
        if kind == "stl":                                   # the offsets made on the device are the synchronous path's
            np.testing.assert_array_equal(lanes[(len(batches) - 1) % 2].voff, ref_eng.voff)
    finally:
        for e in lanes + [ref_eng]:
            e.close()


def test_a_rejected_staged_batch_leaves_the_resident_one(batches, unet_weights):
    e = make_lane(unet_weights)
    try:
        e.upload(batches[0])
        want = e.run(_lib.STAGE_ALL).copy()
        v, f, vo, fo = Engine.pack_meshes(batches[1])
        bad_f = f.copy()
        bad_f[int(fo[2]) + 7, 1] = int(vo[3] - vo[2])        # one index past its mesh
        e.stage((v, bad_f, vo, fo))
        with pytest.raises(ShoulderHipError) as ei:
            e.commit_staged()
        assert ei.value.code == -1 and "face index" in str(ei.value)
        assert not e.staged and e.B == 5
        bad_v = v.copy()
        bad_v[int(vo[4]) + 11, 2] = np.nan
        e.stage((bad_v, f, vo, fo))
        with pytest.raises(ShoulderHipError) as ei:
            e.commit_staged()
        assert ei.value.code == -1 and "NaN" in str(ei.value)
        e.stage_stl([stl_image(*batches[1][0]), stl_image(*batches[1][1])[:-50] + b"\0" * 50])      # a file with a degenerate last record is fine ...
        e.commit_staged()
        assert e.B == 2
        with pytest.raises(ShoulderHipError):                 # ... a truncated one is refused at once
            e.stage_stl([stl_image(*batches[1][0])[:-3]])
        e.upload(batches[0])
        assert e.run(_lib.STAGE_ALL).tobytes() == want.tobytes()
    finally:
        e.close()


def test_a_run_between_stage_and_commit_voids_only_the_prepared_hulls(batches, unet_weights):
    e = make_lane(unet_weights)
    ref = make_lane(unet_weights)
    try:
        ref.upload(batches[2])
        want2 = ref.run(_lib.STAGE_ALL).copy()
        ref.upload(batches[3])
        want3 = ref.run(_lib.STAGE_ALL).copy()
        e.upload(batches[2])
        e.stage(batches[3])
        assert e.run(_lib.STAGE_ALL).tobytes() == want2.tobytes()      # the RESIDENT batch; takes the pinned slot the staged hulls sat in
        e.commit_staged()
        assert e.run(_lib.STAGE_ALL).tobytes() == want3.tobytes()
        # and the device hull: nothing is prepared on the host, the commit is all there is
        e.set_hull_mode("device")
        ref.set_hull_mode("device")
        ref.upload(batches[4])
        want4 = ref.run(_lib.STAGE_ALL).copy()
        e.stage(batches[4])
        e.commit_staged()
        assert e.run(_lib.STAGE_ALL).tobytes() == want4.tobytes()
    finally:
        e.close()
        ref.close()
