"""Ping-pong level-0 kernels (k_unet16_pp.h) against the first generation: logits compared bit for bit / by max deviation on random
images, several shapes and batch sizes, run after run.  python tools/probes/pp_check.py [VAR ...]   (VAR: environment switches that
select the OLD path when set to 0, default SHOULDER_L0_PP)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from shoulder_amd import _lib, unet_spec
from shoulder_amd.engine import Engine

switches = sys.argv[1:] or ["SHOULDER_L0_PP"]
eng = Engine(0)
eng.load_unet(unet_spec.make_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
rng = np.random.default_rng(5)
bad = 0
for dt, nm in ((_lib.UNET_BF16, "bf16"), (_lib.UNET_F16, "f16")):
    eng.set_params(unet_dtype=dt)
    for H, W, n in ((256, 512, 1), (256, 512, 3), (512, 512, 5), (512, 512, 16), (512, 512, 64)):
        img = rng.random((n, H, W), dtype=np.float32)
        for s in switches: os.environ[s] = "0"
        a = eng.unet_infer(img)
        for s in switches: os.environ.pop(s, None)
        for rep in range(3):
            b = eng.unet_infer(img)
            same = np.array_equal(a, b)
            d = float(np.abs(a - b).max())
            flips = int(((a > 0) != (b > 0)).sum())
            if not same: bad += 1
            print(f"{nm} {H}x{W} n={n} rep={rep}: identical={same} max|d|={d:.3e} mask flips={flips}", flush=True)
print("PP_CHECK", "OK" if bad == 0 else f"DIFFERENT in {bad} runs")
