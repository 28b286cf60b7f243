"""the anatomic-neck network alone.  Inputs [B,1,256,512] and [B,1,512,512], U(0,1), seed 1234;
`--unet f16` = BASELINE configs[4] (fp16 MFMA conv, f32 accumulate), `bf16` the bench's throughput dtype, `f32` the parity path.
Prints one JSON line per input shape: images/s, ms per forward, TFLOP/s over the conv layers (HIP-event kernel times).
  python tools/bench_unet.py [--batch 64] [--unet bf16|f16|f32] [--reps 5]
"""
import argparse, json, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from shoulder_amd import _lib, unet_spec
from shoulder_amd.engine import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--unet", choices=["bf16", "f16", "f32", "f32x"], default="bf16")
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--layers", action="store_true", help="also print the per-layer device times (ms per forward)")
args = ap.parse_args()
eng = Engine(0)
eng.load_unet(unet_spec.make_teacher_weights(), unet_spec.BASE, unet_spec.DEPTH)
eng.set_params(unet_dtype={"bf16": _lib.UNET_BF16, "f16": _lib.UNET_F16, "f32": _lib.UNET_F32, "f32x": _lib.UNET_F32X}[args.unet])
rng = np.random.default_rng(1234)
for H, W in ((256, 512), (512, 512)):
    img = rng.random((args.batch, H, W), dtype=np.float32)
    eng.unet_infer(img)
    eng.enable_timing(True); eng.reset_timers()
    for _ in range(args.reps):
        eng.unet_infer(img)
    layers = bench.unet_layers(unet_spec.BASE, unet_spec.DEPTH, H, W, eb=4 if args.unet in ("f32", "f32x") else 2)
    tot_ms = 0.0; conv_ms = 0.0; conv_fl = 0.0
    for name in list(layers) + ["unet.pool", "k_pack_w_bf16"]:
        ms, n = eng.kernel_time_ms(name)
        tot_ms += ms * n / args.reps
        if name in layers and name not in ("unet.enc0a", "unet.head") and n:      # the MFMA layers
            conv_ms += ms * n / args.reps; conv_fl += layers[name][0] * args.batch
    if args.layers:
        per = {name: round(eng.kernel_time_ms(name)[0] * eng.kernel_time_ms(name)[1] / args.reps, 4) for name in list(layers) + ["unet.pool", "k_pack_w_bf16"] if eng.kernel_time_ms(name)[1]}
        print(json.dumps({"layers_ms": per}))
    eng.enable_timing(False)
    print(json.dumps({"workload": f"UNet alone f32[{args.batch},1,{H},{W}] U(0,1) seed 1234", "dtype": args.unet,
                      "device_ms_per_forward": round(tot_ms, 3), "images_per_s": round(args.batch / tot_ms * 1e3, 1),
                      "conv_TFLOP_per_s": round(conv_fl / conv_ms * 1e-9, 1), "conv_frac_of_peak": round(conv_fl / conv_ms * 1e-9 / bench.mfma_peak_tf(args.unet), 3)}))
