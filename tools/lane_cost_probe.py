"""What a geometry stage costs the UNet lane when both run at once: lane A loops the anatomic-neck stage (UNet), lane B loops one stage mask.  python tools/lane_cost_probe.py (on the GPU box)"""
import time, sys, os, threading, numpy as np, torch
sys.path.insert(0, os.getcwd())
from shoulder_amd import _lib, synth, unet_spec
from shoulder_amd.engine import Engine
from shoulder_amd.stl import load_stl
verts, faces = load_stl("tests/golden/bones/humerus_left.stl")
B = 64
W = unet_spec.make_teacher_weights()
def mk():
    e = Engine(0); e.load_rfc(); e.load_unet(W, unet_spec.BASE, unet_spec.DEPTH)
    e.set_params(unet_dtype=_lib.UNET_BF16)
    e.upload([(verts, faces)]); e.synth_batch(synth.similarity_transforms(B, verts, seed=1234))
    e.run(_lib.STAGE_ALL)
    return e
A, Bq = mk(), mk()
KA = 30
def loopA():
    t0 = time.perf_counter()
    for _ in range(KA): A.run(_lib.STAGE_ANP)
    return (time.perf_counter() - t0) / KA * 1e3
for _ in range(3): A.run(_lib.STAGE_ANP)
a0 = loopA()
print(f"A alone: {a0:.3f} ms per UNet stage")
for name, mask in (("FULL+DISTAL", 0x82), ("PROXIMAL", 0x10), ("OBB", 0x1), ("GROOVE", 0x20), ("TE", 0x100), ("NECK+CANAL", 0xC)):
    for _ in range(3): Bq.run(mask)
    t0 = time.perf_counter()
    for _ in range(20): Bq.run(mask)
    b0 = (time.perf_counter() - t0) / 20 * 1e3
    stop = [False]; cnt = [0]
    def loopB():
        while not stop[0]:
            Bq.run(mask); cnt[0] += 1
    th = threading.Thread(target=loopB); th.start()
    time.sleep(0.05); c0 = cnt[0]
    a1 = loopA()
    nb = cnt[0] - c0
    stop[0] = True; th.join()
    print(f"{name:12s} B alone {b0:.3f} ms/iter | A with B: {a1:.3f} ms (+{a1 - a0:.3f}) | B iters during A: {nb} -> {KA * a1 / max(nb, 1):.3f} ms/iter | cost to A per B iter: {(a1 - a0) * KA / max(nb, 1):.3f} ms")
