"""HumeralHeadOsteotomy plane bookkeeping against vectors produced by the reference's OWN code
(tests/golden/make_osteotomy_golden.py ran src/shoulder/arthroplasty.py:13-175 on a stand-in humerus): both the oracle
restatement (oracle/osteotomy.py) and the product facade (shoulder_amd/arthroplasty.py; host code, no GPU needed for the
plane algebra) replay the recorded script."""
import ast
import os

import numpy as np
import pytest

from conftest import GOLDEN
from oracle import xform
from oracle.osteotomy import OracleOsteotomy
from shoulder_amd.arthroplasty import HumeralHeadOsteotomy
from shoulder_amd.base import Plane, Transform
from shoulder_amd.csys import transform_plane_pn

G = np.load(os.path.join(GOLDEN, "osteotomy_golden.npz"))
SCRIPT = [ast.literal_eval(str(s)) for s in G["script"]]


class _Neck:
    def __init__(self, bone, point_ct, normal_ct):
        self._b, self._p, self._n = bone, point_ct, normal_ct

    def plane(self):
        return Plane(*transform_plane_pn(self._p, self._n, self._b._tfrm.matrix))


class FakeHumerus:
    """What the facade class touches of a Humerus, without an engine."""

    def __init__(self, T_start, T_anp, point_ct, normal_ct, side):
        self._tfrm = Transform()
        self._tfrm.matrix = T_start
        self._T_anp, self._side = T_anp, side
        self.anatomic_neck = _Neck(self, point_ct, normal_ct)

    def side(self):
        return self._side

    def apply_csys_canal_articular(self):
        self._tfrm.matrix = self._T_anp.copy()

    def apply_csys_ct(self):
        self._tfrm.reset()

    def apply_csys_custom(self, T, from_ct=True):
        self._tfrm.matrix = T


def close(row, point, normal, ns):
    np.testing.assert_allclose(point, row[:3], rtol=0, atol=1e-9)
    np.testing.assert_allclose(normal, row[3:6], rtol=0, atol=1e-12)
    assert ns == pytest.approx(row[6], abs=1e-9)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_facade_replays_the_reference(case):
    g = lambda k: G[f"c{case}_{k}"]
    hum = FakeHumerus(g("T_start").copy(), g("T_anp"), g("point_ct"), g("normal_ct"), str(g("side")))
    ost = HumeralHeadOsteotomy(hum)
    np.testing.assert_allclose(hum._tfrm.matrix, g("transform_after_init"), rtol=0, atol=1e-12)
    rows, retro = g("rows"), []
    close(rows[0], ost.plane.point, ost.plane.normal, ost.neckshaft_rel)
    for i, step in enumerate(SCRIPT):
        if step[0] == "read_retroversion_rel":
            retro.append(ost.retroversion_rel)
        elif step[0] == "move":
            hum.apply_csys_custom(g("T_move").copy())
        else:
            getattr(ost, step[0])(*step[1:])
        close(rows[i + 1], ost.plane.point, ost.plane.normal, ost.neckshaft_rel)
    np.testing.assert_allclose(retro, g("retro"), rtol=0, atol=1e-9)


@pytest.mark.parametrize("case", [0, 1, 2])
def test_oracle_replays_the_reference(case):
    g = lambda k: G[f"c{case}_{k}"]
    O = OracleOsteotomy(g("T_anp"), g("point_ct"), g("normal_ct"), str(g("side")))
    T = g("T_start").copy()
    rows, retro = g("rows"), []
    close(rows[0], *O.plane(T), O.neckshaft_rel())
    for i, step in enumerate(SCRIPT):
        if step[0] == "read_retroversion_rel":
            retro.append(O.retroversion_rel())
        elif step[0] == "move":
            T = g("T_move").copy()
        else:
            getattr(O, step[0])(*step[1:])
        close(rows[i + 1], *O.plane(T), O.neckshaft_rel())
    np.testing.assert_allclose(retro, g("retro"), rtol=0, atol=1e-9)
