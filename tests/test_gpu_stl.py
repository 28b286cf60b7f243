"""Binary STL ingest on the device (sh_upload_stl, k_stl.h; `trimesh.load_mesh` of mesh.py:22-27 with the canonical merge rule)
against the host routine (shoulder_amd/stl.py) and the oracle (oracle/stl.py): bit-identical vertices and faces."""
import os
import struct
import time

import numpy as np
import pytest

from conftest import BONES
from oracle import stl as o_stl
from shoulder_amd import _lib
from shoulder_amd.stl import load_stl

pytestmark = pytest.mark.gpu
FILES = ["humerus_left", "humerus_left_flipped", "humerus_left_trab", "humerus_right", "proximal_left_cut"]


def _fetch_meshes(engine):
    B = engine.B
    v = engine.fetch("verts", np.float32, (int(engine.voff[-1]), 3))
    f = engine.fetch("faces", np.int32, (int(engine.foff[-1]), 3))
    return [(v[engine.voff[b]:engine.voff[b + 1]], f[engine.foff[b]:engine.foff[b + 1]]) for b in range(B)]


def test_fixture_files_bit_identical(engine):
    paths = [os.path.join(BONES, n + ".stl") for n in FILES]
    engine.upload_stl(paths)
    got = _fetch_meshes(engine)
    for p, (v, f) in zip(paths, got):
        hv, hf = load_stl(p)
        ov, of = o_stl.load_stl(p)
        assert np.array_equal(hv.view(np.uint32), ov.view(np.uint32)) and np.array_equal(hf, of)
        assert v.shape == hv.shape and f.shape == hf.shape
        assert np.array_equal(v.view(np.uint32), hv.view(np.uint32))
        assert np.array_equal(f, hf)


def _stl_bytes(tris):
    tris = np.asarray(tris, dtype=np.float32)
    out = bytearray(b"x" * 80) + struct.pack("<I", len(tris))
    for t in tris:
        out += struct.pack("<12fH", 0, 0, 0, *t.reshape(-1), 0)
    return bytes(out)


def test_merge_edge_cases(engine, tmp_path):
    """-0.0 / +0.0 are one vertex, first-appearance numbering, triangles that use a vertex twice are dropped and the rest
    keeps file order; odd triangle counts exercise the 2-byte alignment of the records."""
    a, b, c, d, e = [0.0, 0, 0], [1, 0, 0], [0, 1, 0], [0, 0, 1], [1, 1, 1]
    nz = [-0.0, 0.0, -0.0]
    tris = [[a, b, c], [nz, c, d], [b, b, e], [a, d, b], [c, b, e], [e, d, c], [d, e, b]]      # third one is degenerate
    blob = _stl_bytes(tris)
    p = tmp_path / "t.stl"
    p.write_bytes(blob)
    hv, hf = load_stl(p)
    engine.upload_stl([blob, str(p)])
    for v, f in _fetch_meshes(engine):
        assert np.array_equal(v.view(np.uint32), hv.view(np.uint32)) and np.array_equal(f, hf)
    assert len(hv) == 5 and len(hf) == 6 and hf[1].tolist() == [0, 2, 3]


def test_rejects_bad_files(engine):
    good = open(os.path.join(BONES, "humerus_right.stl"), "rb").read()
    for bad in (good[:50], good[:-1], good[:84]):
        with pytest.raises(Exception):
            engine.upload_stl([bad])


def test_landmarks_from_device_ingest(engine, oracle_bones):
    """The whole path fed by sh_upload_stl equals the path fed by host-merged meshes."""
    p = os.path.join(BONES, "humerus_left.stl")
    v, f = load_stl(p)
    engine.upload([(v, f)])
    ref = engine.run(_lib.STAGE_ALL).copy()
    engine.upload_stl([p])
    got = engine.run(_lib.STAGE_ALL).copy()
    assert got.tobytes() == ref.tobytes()


def test_ingest_speed(engine):
    paths = [os.path.join(BONES, n + ".stl") for n in FILES[:4]] * 16
    blobs = [open(p, "rb").read() for p in paths]
    engine.upload_stl(blobs)
    t0 = time.perf_counter()
    engine.upload_stl(blobs)
    dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    for p in paths[:4]:
        load_stl(p)
    host = (time.perf_counter() - t0) * 16
    print(f"\n64 STL files: device ingest {dev * 1e3:.1f} ms (incl. H2D of {sum(map(len, blobs)) / 1e6:.0f} MB), host NumPy merge {host * 1e3:.0f} ms")
    assert dev < host
