"""sh_slice_mesh_planes (k_clip.h) against oracle/clip.py: float64 coordinates bit-for-bit, faces / cut edges identical
(the oracle's canonical numbering is the library's)."""
import os

import numpy as np
import pytest

from conftest import BONES
from oracle import clip
from shoulder_amd.stl import load_stl
from test_oracle_clip import area, cube, volume_about

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from shoulder_amd.engine import Engine
    return Engine(0)


def check(eng, v, f, origins, normals):
    got = eng.slice_mesh_planes(v, f, origins, normals, edges=True)
    assert len(got) == len(origins)
    for (gv, gf, ge), o, n in zip(got, origins, normals):
        wv, wf, we = clip.slice_plane(v, f, o, n)
        assert gv.shape == wv.shape and gf.shape == wf.shape and ge.shape == we.shape
        assert np.array_equal(gv.view(np.int64), wv.view(np.int64))
        assert np.array_equal(gf, wf) and np.array_equal(ge, we)
    return got


def test_cube_cases(eng):
    v, f = cube()
    origins = [[0, 0, 0.5], [0, 0, 1], [0, 0, 1], [0, 0, 0], [0, 0, 2], [0, 0, 2], [0.2, 0.3, 0.4]]
    normals = [[0, 0, 1], [0, 0, 1], [0, 0, -1], [1, -1, 0], [0, 0, 1], [0, 0, -1], [0.3, -0.2, 0.9]]
    check(eng, v, f, origins, normals)


def test_humerus_sweep(eng):
    v, f = load_stl(os.path.join(BONES, "humerus_left.stl"))
    v = v.astype(np.float64)
    rng = np.random.default_rng(9)
    P = 12
    normals = rng.normal(size=(P, 3))
    normals[0] = (0, 0, 1)
    origins = v[rng.integers(0, len(v), P)] + rng.normal(scale=2.0, size=(P, 3))
    origins[1] = v[100]                       # a plane exactly through a vertex
    got = check(eng, v, f, origins, normals)
    # and the two halves of one plane re-assemble the surface
    (hv, hf, _), (rv, rf, _) = eng.slice_mesh_planes(v, f, [origins[2], origins[2]], [normals[2], -normals[2]], edges=True)
    A = area(v, f)
    assert abs(area(hv, hf) + area(rv, rf) - A) < 1e-9 * A
    V = volume_about(v, f, origins[2])
    assert abs(volume_about(hv, hf, origins[2]) + volume_about(rv, rf, origins[2]) - V) < 1e-9 * abs(V)
    assert sum(len(g[1]) for g in got) > 0


def test_bad_arguments(eng):
    from shoulder_amd.engine import ShoulderHipError
    v, f = cube()
    with pytest.raises(ShoulderHipError):
        eng.slice_mesh_planes(v, f, [[0, 0, 0]], [[0, 0, 0]])
    bad = f.copy()
    bad[0, 0] = 99
    with pytest.raises(ShoulderHipError):
        eng.slice_mesh_planes(v, bad, [[0, 0, 0.5]], [[0, 0, 1]])
