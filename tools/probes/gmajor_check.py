import os, sys, numpy as np
sys.path.insert(0, os.getcwd())
from shoulder_amd import _lib, unet_spec
from shoulder_amd.engine import Engine
e = Engine(0); e.load_unet(unet_spec.make_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH); e.set_params(unet_dtype=_lib.UNET_BF16)
img = np.random.default_rng(5).random((5, 512, 512), dtype=np.float32)
os.environ["SHOULDER_GMAJOR"] = "0"; a = e.unet_infer(img)
os.environ["SHOULDER_GMAJOR"] = "1"; b = e.unet_infer(img)
print("gmajor bit-identical:", np.array_equal(a, b))
