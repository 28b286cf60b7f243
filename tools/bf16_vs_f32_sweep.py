"""bf16 UNet path vs f32 UNet path on 4 x 64 similarity copies (GPU box): per-landmark worst deviation."""
import sys, os, numpy as np
sys.path.insert(0, os.getcwd())
from shoulder_amd import _lib, synth, unet_spec
from shoulder_amd.engine import Engine
from shoulder_amd.stl import load_stl
e = Engine(0); e.load_rfc(); e.load_unet(unet_spec.make_teacher_weights(), 32, 4)
worst = {}
for bi, name in enumerate(["humerus_left", "humerus_right", "humerus_left_trab", "humerus_left_flipped"]):
    v, f = load_stl(f"tests/golden/bones/{name}.stl")
    e.upload([(v, f)]); e.synth_batch(synth.similarity_transforms(64, v, seed=6000 + bi))
    e.set_params(unet_dtype=_lib.UNET_F32); a = e.run(_lib.STAGE_ALL).copy()
    e.set_params(unet_dtype=_lib.UNET_BF16); b = e.run(_lib.STAGE_ALL).copy()
    assert (a["status"] == 0).all() and (b["status"] == 0).all()
    row = {}
    for k in ("anp_plane_point", "anp_axis_normal", "anp_axis_central", "neckshaft", "retroversion", "radius_curvature", "te_axis", "canal_axis", "groove_axis"):
        d = np.abs(np.asarray(a[k], dtype=np.float64) - np.asarray(b[k], dtype=np.float64)).reshape(64, -1).max(axis=1)
        row[k] = d.max(); worst[k] = max(worst.get(k, 0), d.max())
    print(name, "n_anp diff max", int(np.abs(a["n_anp"] - b["n_anp"]).max()), {k: "%.2e" % x for k, x in row.items()}, flush=True)
print("WORST bf16 vs f32 over 256 humeri:", {k: "%.3g" % x for k, x in worst.items()})
