"""The pruning bound in front of k_obb_candidates (k_obb.h: k_obb_face_area2 / k_obb_bounds / k_obb_select) against the
evaluation of every direction (SHOULDER_OBB_PRUNE=0), through the C-ABI: the bound never exceeds the exact volume of its
direction, the directions it drops cannot be the minimum, and the box frame (mesh.py:63-125) is the same bit for bit."""
import os

import numpy as np
import pytest

from shoulder_amd import _lib, synth

pytestmark = pytest.mark.gpu
HF = 32768      # SH_HF: stride of the per-face arrays


def _obb(engine, B):
    engine.run(_lib.STAGE_OBB)
    nf = engine.fetch("hull.nf", np.int32, (B,))
    vol = engine.fetch("obb.cand_vol", np.float64, (B, HF))
    edge = engine.fetch("obb.cand_edge", np.int32, (B, HF))
    lb = engine.fetch("obb.lb", np.float64, (B, HF))
    T = engine.fetch("obb.T_pre", np.float64, (B, 4, 4))
    return nf, vol, edge, lb, T


@pytest.mark.parametrize("case", ["fixtures", "batch64"])
def test_pruned_candidates_give_the_same_box(engine, oracle_bones, case):
    from conftest import engine_with_env

    def load(e):
        if case == "fixtures":
            bones = [oracle_bones(n) for n in ("humerus_left", "humerus_left_flipped", "humerus_left_trab", "humerus_right")]
            e.upload([(b.verts, b.faces) for b in bones])
            return len(bones)
        h = oracle_bones("humerus_left")
        e.upload([(h.verts, h.faces)])
        e.synth_batch(synth.similarity_transforms(64, h.verts, seed=1234))
        return 64
    with engine_with_env(SHOULDER_OBB_PRUNE=0) as e_all:      # (a switch of the context, read when it is created)
        B = load(e_all)
        nf, vol_all, edge_all, lb, T_all = _obb(e_all, B)
    B = load(engine)
    nf2, vol, edge, lb2, T = _obb(engine, B)
    np.testing.assert_array_equal(nf, nf2)
    np.testing.assert_array_equal(T, T_all)
    for b in range(B):
        n = int(nf[b])
        va, v = vol_all[b, :n], vol[b, :n]
        done_all = va < 1e299                                   # (directions with a degenerate silhouette edge stay 1e300 in both)
        done = v < 1e299
        assert done_all.sum() > 0.99 * n
        assert done.sum() < 0.5 * n, (b, int(done.sum()), n)    # the bound does prune
        # what was evaluated is what the full evaluation found for that direction (in-kernel skips aside: they only drop
        # directions whose own bound already exceeds the best volume)
        np.testing.assert_array_equal(v[done], va[done])
        np.testing.assert_array_equal(edge[b, :n][done], edge_all[b, :n][done])
        assert v.min() == va.min() and int(np.argmin(v)) == int(np.argmin(va))
        # the bound is a lower bound of its direction's exact volume, up to the rounding the selection allows for
        assert (lb[b, :n][done_all] * (1 - 1e-9) <= va[done_all]).all()
        # dropped directions could not have won
        assert (va[~done & done_all] > va.min()).all()
