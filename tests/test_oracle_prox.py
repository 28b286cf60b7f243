"""Oracle restatement of `shoulder.ProximalHumerus` (oracle/prox.py) on the cut fixture: invariants the reference's
algorithm implies (no reference vectors exist for this class -- parity unpinned, see oracle/prox.py)."""
import os

import numpy as np
import pytest

from conftest import BONES
from oracle import obb as o_obb
from oracle import prox
from oracle.section import ZSlicer, ring_area
from oracle.stl import load_stl


@pytest.fixture(scope="module")
def cut():
    v, f = load_stl(os.path.join(BONES, "proximal_left_cut.stl"))
    return v.astype(np.float64), f, prox.prox_obb(v.astype(np.float64), f)


def test_fixture_is_watertight_and_outward(cut):
    v, f, _ = cut
    e = np.sort(np.r_[f[:, [0, 1]], f[:, [1, 2]], f[:, [2, 0]]], axis=1)
    _, counts = np.unique(e, axis=0, return_counts=True)
    assert (counts == 2).all()
    a, b, c = v[f[:, 0]], v[f[:, 1]], v[f[:, 2]]
    assert np.einsum("ij,ij->i", a, np.cross(b, c)).sum() > 0


def test_prox_obb_head_up_and_canal_range(cut):
    v, f, o = cut
    a = o["z_area"]
    assert int(np.argmax(a)) > 50                       # after the flip the head (largest section) is in the upper half
    lo, hi = o["canal_zs"]
    assert 0 <= lo < 10 and 50 < hi < 80                # the shaft, from just above the cut to below the head
    assert o["cutoff_pcts"] == [lo / 100, hi / 100] and o["cutoff_bot"] == lo
    assert (o["grad"][lo:hi + 1] < 10).all() and (hi + 1 == 100 or o["grad"][hi + 1] >= 10)
    # the long axis of the cut piece stays within ten degrees of the whole bone's
    vw, fw = load_stl(os.path.join(BONES, "humerus_left.stl"))
    zw = o_obb.full_obb(vw.astype(np.float64), fw)["transform"][2, :3]
    assert abs(float(np.dot(zw, o["transform"][2, :3]))) > np.cos(np.deg2rad(10.0))


def test_total_area_is_signed_loop_sum(cut):
    v, f, o = cut
    sl = ZSlicer(o["verts_obb"], f)
    for z in (-60.0, 0.0, 40.0, 70.0):
        rings = sl.loops(z)
        if len(rings) == 1:
            assert prox.total_area(sl, z) == pytest.approx(ring_area(rings[0]), rel=1e-12)
    assert prox.total_area(sl, 1e6) == 0.0


def test_consecutive_first_longest():
    assert list(prox.consecutive(np.array([0, 1, 2, 5, 6, 7, 9]))) == [0, 1, 2]
    assert list(prox.consecutive(np.array([3, 7, 8, 9, 10, 20, 21]))) == [7, 8, 9, 10]
