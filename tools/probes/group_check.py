import os, sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
from shoulder_amd import _lib, unet_spec
from shoulder_amd.engine import Engine
from conftest import _teacher_weights
e = Engine(0); e.load_rfc(); e.load_unet(_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH); e.set_params(unet_dtype=_lib.UNET_BF16)
rng = np.random.default_rng(3)
for n in (20, 64):
    img = rng.random((n, 512, 512), dtype=np.float32)
    os.environ["SHOULDER_UNET_GROUP"] = "0"; a = e.unet_infer(img)
    for g in ("8", "16", "7"):
        os.environ["SHOULDER_UNET_GROUP"] = g; b = e.unet_infer(img)
        print(n, g, "equal" if np.array_equal(a, b) else "DIFFERENT %g" % np.abs(a - b).max())
