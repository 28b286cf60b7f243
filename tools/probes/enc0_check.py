"""k_enc0_pp against k_enc0_fused16 (SHOULDER_ENC0_PP=0): skip0 and the pooled tensor agree to an ulp or two of the element type (the
first conv's products are summed in another order), the logits far inside the type's tolerance; borders and ragged batches included."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from shoulder_amd import _lib, unet_spec
from shoulder_amd.engine import Engine
eng = Engine(0)
eng.load_unet(unet_spec.make_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
rng = np.random.default_rng(7)
bad = 0
for dt, nm, ulp in ((_lib.UNET_BF16, "bf16", 2.0 ** -8), (_lib.UNET_F16, "f16", 2.0 ** -11)):
    tof = (lambda u: (u.astype(np.uint32) << 16).view(np.float32)) if nm == "bf16" else (lambda u: u.view(np.float16).astype(np.float32))
    eng.set_params(unet_dtype=dt)
    for shape in ((5, 256, 256), (2, 256, 512), (3, 512, 512), (64, 512, 512)):
        img = rng.random(shape, dtype=np.float32)
        n0 = shape[0] * shape[1] * shape[2] * 32
        os.environ["SHOULDER_ENC0_PP"] = "0"
        a = eng.unet_infer(img)
        s0a = tof(eng.fetch("unet16.skip0", np.uint16)[:n0].copy()).reshape(shape + (32,))
        os.environ.pop("SHOULDER_ENC0_PP")
        for rep in range(2):
            b = eng.unet_infer(img)
            s0b = tof(eng.fetch("unet16.skip0", np.uint16)[:n0].copy()).reshape(shape + (32,))
            d = np.abs(s0a - s0b)
            scale = max(1.0, float(np.abs(s0a).max()))
            ok = float(d.max()) <= 8 * ulp * scale and float(d.mean()) < ulp * scale / 8 and np.isfinite(s0b).all()
            dl = float(np.abs(a - b).max())
            if not ok: bad += 1
            print(f"{nm} {shape} rep {rep}: skip0 max|d| {float(d.max()):.3e} mean {float(d.mean()):.3e} (ulp*scale {ulp * scale:.3e}) logits max|d| {dl:.3e} flips {int(((a > 0) != (b > 0)).sum())} {'ok' if ok else 'BAD'}", flush=True)
print("ENC0_CHECK", "OK" if bad == 0 else f"BAD in {bad}")
