"""shoulder_amd.Plot (reference src/shoulder/plotting.py:45-128): traces and names as the reference builds them."""
import os

import numpy as np
import pytest

from conftest import BONES

pytestmark = pytest.mark.gpu


def test_plot_landmarks_and_surgery(engine):
    pytest.importorskip("plotly")
    import shoulder_amd as shoulder
    hum = shoulder.Humerus(os.path.join(BONES, "humerus_left.stl"), engine=engine)
    fig0 = shoulder.Plot(hum).figure
    names0 = [t.name for t in fig0.data[1:]]
    assert fig0.data[0].type == "mesh3d" and fig0.data[0].opacity == 0.7 and fig0.data[0].color == "#DFDAC0"
    assert names0 == ["Surgical Neck"]                     # the only landmark the constructor computes (bone.py:120)
    hum.apply_csys_canal_transepiconylar()
    hum.bicipital_groove.axis()
    hum.canal.points()
    p = shoulder.Plot(hum, opacity=0.9)
    fig = p.figure
    assert fig.layout.title.text == "humerus_left.stl" and fig.layout.scene.aspectmode == "data"
    assert fig.data[0].opacity == 0.9 and len(fig.data[0].x) == len(hum.mesh.vertices)
    # landmarks in the alphabetical attribute order of Bone._list_landmarks; the anatomic neck contributes two traces
    assert [t.name for t in fig.data[1:]] == ["Anatomic Neck", "Anatomic Neck Plane", "Bicipital Groove", "Canal Axis", "Surgical Neck",
                                              "Transverse Epicondylar Axis"]
    np.testing.assert_allclose(np.c_[fig.data[-1].x, fig.data[-1].y, fig.data[-1].z], hum.trans_epiconylar.axis())
    ost = shoulder.HumeralHeadOsteotomy(hum)
    fs = shoulder.Plot(ost, opacity=0.5).figure
    assert [t.type for t in fs.data] == ["mesh3d", "mesh3d"] and fs.data[0].opacity == 0.5 and fs.data[1].opacity is None
    with pytest.raises(ValueError, match="Bone or HumeralHead"):
        shoulder.Plot(object())
