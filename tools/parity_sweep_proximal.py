"""One-off randomized parity sweep of the ProximalHumerus path: device (f32 UNet) vs oracle/prox.py on similarity copies of the cut fixture."""
import sys, os, time, numpy as np, subprocess
sys.path.insert(0, os.getcwd())
from shoulder_amd import _lib, synth, unet_spec
from shoulder_amd.engine import Engine
from shoulder_amd.stl import load_stl
from oracle.prox import OracleProximalHumerus
from oracle import rfc
subprocess.run(["make", "-C", "oracle"], capture_output=True)
tables = rfc.load_tables("shoulder_amd/models/rfc_bg3.npz")
W = unet_spec.make_teacher_weights()
e = Engine(0); e.load_rfc(); e.load_unet(W, 32, 4); e.set_params(unet_dtype=_lib.UNET_F32, bone_kind=_lib.BONE_PROXIMAL)
MASK = _lib.STAGE_ALL & ~(_lib.STAGE_DISTAL | _lib.STAGE_TE)
N = int(os.environ.get("NPER", "16"))
v, f = load_stl("tests/golden/bones/proximal_left_cut.stl")
T = synth.similarity_transforms(N, v, seed=int(os.environ.get("SEED0", "900")))
meshes = [(synth.apply_similarity(T[i], v), f) for i in range(N)]
e.upload(meshes)
lm = e.run(MASK)
worst = {}
for i, (mv, mf) in enumerate(meshes):
    t0 = time.time()
    h = OracleProximalHumerus(mv, mf, tables, W, unet_eval="chain")
    L = h.landmarks(); r = lm[i]
    ok = r["status"] == 0 and bool(r["flipped"]) == h.obb["flipped"] and int(r["neck_index"]) == h.neck["bkp"] and float(r["bg_theta"]) == L["bg_theta"] and int(r["n_anp"]) == len(L["anp_points"]) \
        and list(r["canal_cutoff"]) == list(h.obb["cutoff_pcts"])
    d = {}
    if ok:
        for k in ("canal_axis", "groove_axis", "anp_plane_point", "anp_axis_normal", "anp_axis_central"):
            d[k] = float(np.abs(np.asarray(r[k]).reshape(np.shape(L[k])) - L[k]).max())
        d["anp_points"] = float(np.abs(r["anp_points"].reshape(-1, 3)[: int(r["n_anp"])] - L["anp_points"]).max())
        for k, x in d.items(): worst[k] = max(worst.get(k, 0.0), x)
    print(i, "OK" if ok else "MISMATCH", "max diff %.2e" % (max(d.values()) if d else -1), "status", int(r["status"]), "(%.1fs)" % (time.time() - t0),
          "" if ok else (bool(r["flipped"]), h.obb["flipped"], int(r["neck_index"]), h.neck["bkp"], float(r["bg_theta"]), L["bg_theta"], int(r["n_anp"]), len(L["anp_points"]), list(r["canal_cutoff"]), h.obb["cutoff_pcts"]), flush=True)
print("WORST", {k: "%.2e" % x for k, x in worst.items()})
