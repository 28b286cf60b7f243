"""Time the kernels of one stage mask on the synthetic batch: python tools/time_stage.py [mask] [B]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from shoulder_amd import _lib, synth, unet_spec
from shoulder_amd.engine import Engine
from shoulder_amd.stl import load_stl
mask = int(sys.argv[1], 0) if len(sys.argv) > 1 else _lib.STAGE_OBB
B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
verts, faces = load_stl(os.path.join(ROOT, "tests", "golden", "bones", "humerus_left.stl"))
eng = Engine(0); eng.load_rfc(); eng.load_unet(unet_spec.make_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
eng.set_params(unet_dtype=_lib.UNET_BF16)
eng.upload([(verts, faces)]); eng.synth_batch(synth.similarity_transforms(B, verts, seed=1234))
def run():
    try: eng.run(mask)
    except Exception as e: print("run error:", str(e)[:100])
run(); eng.enable_timing(True); eng.reset_timers()
for _ in range(5): run()
import bench
names = list(bench.GEOM_KERNELS) + ["host.hull", "host.verts_d2h", "unet.pool", "unet.enc0a", "unet.head"] + list(bench.unet_layers(unet_spec.BASE, unet_spec.DEPTH, 512, 512, eb=2))
rows = [(k,) + eng.kernel_time_ms(k) for k in names]
for k, ms, n in sorted(rows, key=lambda r: -r[1] * r[2])[:int(os.environ.get("TOPN", "14"))]:
    if n: print("%-28s %8.3f ms x %.1f per run" % (k, ms, n / 5))
