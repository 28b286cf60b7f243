"""Canonical rule B-8 (flat maximum of the linear-kernel KDE, bicipital_groove.py:184-188).  Found by a randomized
device-vs-oracle sweep: similarity copy 56 of seed 300 of humerus_left_trab.stl has a maximum plateau 11 grid points wide;
before the rule the device and sklearn picked different ends of it (bg_theta -2.171 vs -2.104) and everything downstream
differed.  Both sides must now pick the lowest grid index and agree on every landmark."""
import os

import numpy as np
import pytest
import sklearn.neighbors

from conftest import BONES
from oracle import groove as o_groove
from oracle.humerus import OracleHumerus
from shoulder_amd import _lib, synth
from shoulder_amd.stl import load_stl

pytestmark = pytest.mark.gpu


def test_plateau_case_agrees(engine, rfc_tables, unet_weights):
    v, f = load_stl(os.path.join(BONES, "humerus_left_trab.stl"))
    T = synth.similarity_transforms(64, v, seed=300)
    mv = synth.apply_similarity(T[56], v)
    engine.set_params(unet_dtype=_lib.UNET_F32)
    engine.upload([(mv, f)])
    lm = engine.run(_lib.STAGE_ALL)[0]
    h = OracleHumerus(mv, f, rfc_tables, unet_weights, unet_eval="chain")
    g = h.groove
    # the case really is a plateau: several grid points within the tie tolerance of the maximum
    sel = g["peak_theta"][g["proba"] > 0.4]
    kde = sklearn.neighbors.KernelDensity(kernel="linear").fit(sel.reshape(-1, 1))
    p = np.exp(kde.score_samples(np.linspace(-np.pi, np.pi, 1024).reshape(-1, 1)))
    tied = np.nonzero(p >= p.max() * (1 - o_groove.KDE_TIE))[0]
    assert len(tied) >= 5 and tied[-1] - tied[0] == len(tied) - 1
    assert float(lm["bg_theta"]) == g["bg_theta"] == np.linspace(-np.pi, np.pi, 1024)[tied[0]]
    L = h.landmarks()
    assert int(lm["n_anp"]) == len(L["anp_points"]) and lm["status"] == 0
    for k in ("groove_axis", "anp_plane_point", "anp_axis_normal", "anp_axis_central", "te_axis", "csys"):
        np.testing.assert_allclose(np.asarray(lm[k]).reshape(np.shape(L[k])), L[k], rtol=0, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(lm["anp_points"].reshape(-1, 3)[: int(lm["n_anp"])], L["anp_points"], rtol=0, atol=1e-6)
