"""A/B of the step schedule on one GPU box, one process: resident batch vs a stream of new batches (arrays / STL files), two lanes,
`--rounds` interleaved rounds of `--steps` steps each.  Prints ms per step per variant and round.
    python tools/stream_ab.py [--steps 20] [--rounds 3] [--variants resident,host,stl]
"""
import argparse, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
import bench
from shoulder_amd import _lib, synth, unet_spec
from shoulder_amd.engine import Engine
from shoulder_amd.stl import load_stl

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=20)
ap.add_argument("--rounds", type=int, default=3)
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--variants", default="resident,host,stl")
args = ap.parse_args()
B = args.batch
verts, faces = load_stl(os.path.join(ROOT, "tests", "golden", "bones", "humerus_left.stl"))
w = unet_spec.make_teacher_weights()
T = synth.similarity_transforms(B, verts, seed=1234)
variants = args.variants.split(",")
streams = {k: bench.make_stream(k, verts, faces, B, 4) for k in ("host", "stl") if k in variants}


def lanes_for(kind):
    es = [Engine(0) for _ in range(2)]
    for e in es:
        e.load_rfc(); e.load_unet(w, unet_spec.BASE, unet_spec.DEPTH); e.set_params(unet_dtype=_lib.UNET_BF16)
        e.set_hull_mode("host"); e.set_unet_turns(True)
        if kind == "stl":
            e.upload_stl(streams["stl"]["batches"][-1])
        elif kind == "host":
            e.upload_packed(streams["host"]["batches"][-1])
        else:
            e.upload([(verts, faces)]); e.synth_batch(T)
        e.run(_lib.STAGE_ALL, fetch=False)
    return es


def leg(es, kind, steps):
    strm = streams.get(kind)
    for e in es:
        e.discard_prepared(); e.set_overlap(strm is None)
    t0 = time.perf_counter()
    pend = []
    for s in range(steps):
        e = es[s % 2]
        if steps - s <= 2:
            e.set_overlap(False)
        if len(pend) >= 2:
            pend.pop(0).collect()
        if strm:
            if not e.staged:
                (e.stage_stl if kind == "stl" else e.stage)(strm["batches"][s % 4])
            e.commit_staged()
        e.submit(_lib.STAGE_ALL, fetch="view")
        pend.append(e)
        if strm and s + 2 < steps:
            (e.stage_stl if kind == "stl" else e.stage)(strm["batches"][(s + 2) % 4])
    for e in pend:
        e.collect()
    return 1e3 * (time.perf_counter() - t0) / steps


engs = {k: lanes_for(k) for k in variants}
for r in range(args.rounds):
    for k in variants:
        print(f"round {r} {k:9s} {leg(engs[k], k, args.steps):7.3f} ms/step", flush=True)
