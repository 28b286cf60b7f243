"""Randomized parity sweep (GPU box): device (f32 UNet, the exact path) vs oracle on seeded similarity copies of the fixtures.
    SEED0=500 NPER=32 BONES=humerus_left,humerus_right python tools/parity_sweep.py > gpurun_out/sweep.log
Round 1: ~400 copies, one KDE-plateau case (canonical rule B-8, tests/test_gpu_kde_plateau.py), everything else <= 3e-11 mm."""
import sys, os, time, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from shoulder_amd import _lib, synth, unet_spec
from shoulder_amd.engine import Engine
from shoulder_amd.stl import load_stl
from oracle.humerus import OracleHumerus
from oracle import rfc
import subprocess
subprocess.run(["make", "-C", "oracle"], capture_output=True)
tables = rfc.load_tables("shoulder_amd/models/rfc_bg3.npz")
W = unet_spec.make_teacher_weights()
e = Engine(0); e.load_rfc(); e.load_unet(W, 32, 4); e.set_params(unet_dtype=_lib.UNET_F32)
NPER = int(os.environ.get("NPER", "5"))
worst = {}
BONES_ = os.environ.get("BONES", "humerus_left_trab,humerus_left_flipped,humerus_left,humerus_right").split(",")
for bi, name in enumerate(BONES_):
    v, f = load_stl(f"tests/golden/bones/{name}.stl")
    seed = int(os.environ.get("SEED0", "100")) + bi
    T = synth.similarity_transforms(NPER, v, seed=seed)
    meshes = [(synth.apply_similarity(T[i], v), f) for i in range(NPER)]
    e.upload(meshes)
    lm = e.run(_lib.STAGE_ALL)
    for i, (mv, mf) in enumerate(meshes):
        t0 = time.time()
        h = OracleHumerus(mv, mf, tables, W, unet_eval="chain")
        L = h.landmarks(); M = h.metrics()
        r = lm[i]
        assert r["status"] == 0
        d = {}
        for k in ("canal_axis", "te_axis", "groove_axis", "anp_plane_point", "anp_axis_normal", "anp_axis_central", "csys"):
            d[k] = float(np.abs(np.asarray(r[k]).reshape(np.shape(L[k])) - L[k]).max())
        d["obb"] = float(np.abs(r["obb_transform"].reshape(4, 4) - L["T_obb"]).max())
        d["groove_points"] = float(np.abs(r["groove_points"].reshape(-1, 3) - L["groove_points"]).max())
        n = int(r["n_anp"])
        if n != len(L["anp_points"]):
            print("MISMATCH", name, "seed", seed, "index", i, "n_anp", n, len(L["anp_points"]), flush=True)
            img_d = e.fetch("anp.image", np.float32).reshape(NPER, 512, 512)[i]
            img_o = h.anp_input["image"].astype(np.float32)
            neq = img_d != img_o
            print("  image pixels differing:", int(neq.sum()), "max abs", float(np.abs(img_d - img_o).max()), "where", np.argwhere(neq)[:5].tolist())
            lg_d = e.fetch("anp.logits", np.float32).reshape(NPER, 512, 512)[i]
            lg_o = np.asarray(h.logits(), dtype=np.float32)
            print("  logits differing:", int((lg_d != lg_o).sum()), "max abs", float(np.abs(lg_d - lg_o).max()), "mask flips", int(((lg_d > 0) != (lg_o > 0)).sum()))
            roll_d = e.fetch("anp.roll", np.int32)[i * 512:(i + 1) * 512] if False else None
            print("  bg_theta dev/oracle", float(r["bg_theta"]), h.groove["bg_theta"], "neck_index", int(r["neck_index"]), "n_articular", int(r["n_articular"]))
            continue
        d["anp_points"] = float(np.abs(r["anp_points"].reshape(-1, 3)[:n] - L["anp_points"]).max())
        d["neckshaft"] = abs(float(r["neckshaft"]) - M["neckshaft"]); d["retro"] = abs(float(r["retroversion"]) - M["retroversion"])
        assert ("left", "right")[int(r["side"])] == M["side"] and bool(r["flipped"]) == h.obb["flipped"] and int(r["neck_index"]) == h.neck["index"] if "index" in h.neck else True
        for k, x in d.items(): worst[k] = max(worst.get(k, 0.0), x)
        print(name, i, "max diff %.2e" % max(d.values()), "(%.1fs oracle)" % (time.time() - t0), flush=True)
print("WORST", {k: "%.2e" % x for k, x in worst.items()})
