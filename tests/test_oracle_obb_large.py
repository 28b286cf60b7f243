"""oracle/obb.py: oriented_bounds_large (silhouette edges off the 3-D hull, a chunk of directions per matrix product) is what the GPU
tests of hulls above 16 384 vertices are held to; here it is pinned against oriented_bounds (one qhull call per face normal) on a
fixture and on a strictly convex surface small enough for both."""
import os

import numpy as np
import pytest

from conftest import BONES, convex_surface
from oracle import obb
from shoulder_amd.stl import load_stl


def _same(a, b):
    Ta, ea, va = a
    Tb, eb, vb = b
    assert abs(va - vb) <= 1e-9 * va
    np.testing.assert_allclose(ea, eb, rtol=0, atol=1e-9)
    np.testing.assert_allclose(Ta, Tb, rtol=0, atol=1e-9)


@pytest.mark.parametrize("name", ["humerus_left", "humerus_right", "humerus_left_flipped", "humerus_left_trab"])
def test_on_the_fixtures(name):
    path = os.path.join(BONES, name + ".stl")
    if not os.path.exists(path):
        pytest.skip(name + ".stl is not among the fixtures")
    v, _ = load_stl(path)
    _same(obb.oriented_bounds_large(v.astype(np.float64)), obb.oriented_bounds(v.astype(np.float64)))


@pytest.mark.parametrize("n,seed", [(600, 3), (1500, 4), (2500, 5)])
def test_on_strictly_convex_surfaces(n, seed):
    v, f = convex_surface(n, seed=seed)
    hv_ids, tri, _ = obb.hull(v.astype(np.float64))
    assert len(hv_ids) == len(v) and len(tri) == len(f) == 2 * len(v) - 4      # every vertex is on the hull
    _same(obb.oriented_bounds_large(v.astype(np.float64)), obb.oriented_bounds(v.astype(np.float64)))
