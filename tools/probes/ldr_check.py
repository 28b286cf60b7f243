"""k_conv3_ldr16 against k_conv3_dma16 (SHOULDER_DMA_LDR=0): logits and every intermediate tensor of the 16-bit network compared."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from shoulder_amd import _lib, unet_spec
from shoulder_amd.engine import Engine
eng = Engine(0)
eng.load_unet(unet_spec.make_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
rng = np.random.default_rng(3)
for dt, nm in ((_lib.UNET_BF16, "bf16"), (_lib.UNET_F16, "f16")):
    tof = (lambda u: (u.astype(np.uint32) << 16).view(np.float32)) if nm == "bf16" else (lambda u: u.view(np.float16).astype(np.float32))
    eng.set_params(unet_dtype=dt)
    img = rng.random((3, 256, 512), dtype=np.float32)
    os.environ["SHOULDER_DMA_LDR"] = "0"
    a = eng.unet_infer(img)
    ta = {k: eng.fetch(k, np.uint16).copy() for k in ("unet16.skip1", "unet16.skip2", "unet16.skip3", "unet16.a", "unet16.b")}
    os.environ.pop("SHOULDER_DMA_LDR")
    b = eng.unet_infer(img)
    tb = {k: eng.fetch(k, np.uint16).copy() for k in ta}
    print(nm, "logits identical", np.array_equal(a, b), "max|d|", float(np.abs(a - b).max()))
    for k in ta:
        x, y = ta[k], tb[k]
        n = min(len(x), len(y))
        neq = int((x[:n] != y[:n]).sum())
        d = np.abs(tof(x[:n]) - tof(y[:n]))
        negz = int(((x[:n] == 0x8000) | (y[:n] == 0x8000)).sum())
        print(f"   {k}: differing elements {neq} of {n}, max|d| {float(d.max()):.3e}, -0.0 patterns {negz}")
