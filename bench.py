#!/usr/bin/env python3
"""bench.py -- humerus meshes/s end-to-end (all four landmarks) on N MI355X of one node.

    python bench.py --gpus 1 --steps 5 --warmup 1
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

One process per GPU.  A step = one pass of the whole hot path (SH_STAGE_ALL: OBB -> slices -> neck ->
canal -> groove -> anatomic neck incl. UNet -> trans-epicondylar -> csys -> landmark records) over
one batch of --batch synthetic humeri per GPU (BASELINE config 3 at N=1, config 4 at N=8: similarity
copies of tests/golden/bones/humerus_left.stl, seed 1234, contiguous shards), with the batch already
resident in HBM, followed by the gather of the landmark records to rank 0.  Parameters (UNet + random
forest) are loaded on rank 0 and broadcast over RCCL before the timed region.  No data-path
collective: meshes are independent (weak scaling).

The JSON line also carries
  roofline      for the kernel with the largest share of device time (HIP events on the engine's
                stream around every launch of the timed steps), algorithmic flops/bytes per launch
                (DESIGN.md "Kernels") / average launch duration vs the gfx950 peak;
  cpu_baseline  the oracle (NumPy/SciPy restatement of the reference, oracle/) on the same synthetic
                meshes, one thread, rank 0, N=1 only, a bounded sample.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_HBM_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8.0 TB/s
PEAK_MFMA_F32_TF = 157.3       # dense f32 MFMA (= vector rate)
PEAK_MFMA_BF16_TF = 2500.0
ROUNDS = ("r05", "r04", "r03", "r02", "r01")      # committed profile sets, newest first


def mfma_peak_tf(unet):
    """Peak of USEFUL flops per UNet arithmetic: f32 MFMA = the vector rate; f32x issues three 16-bit MFMAs per product (k_unet_x3.h),
    so its algorithmic flops are priced against a third of the dense 16-bit peak (shared with tools/bench_unet.py)."""
    return {"f32": PEAK_MFMA_F32_TF, "f32x": PEAK_MFMA_BF16_TF / 3.0}.get(unet, PEAK_MFMA_BF16_TF)

GEOM_KERNELS = ["k_hull_rounds", "k_hull_flag", "k_apply_csys", "k_obb_face_area2", "k_obb_bounds", "k_obb_select", "k_obb_seed", "k_obb_candidates", "k_obb_pick", "k_obb_end_points", "k_obb_ends", "k_transform_verts", "k_make_planes",
                "k_slice_emit", "k_slice_link", "k_slice_link_large", "k_resample_polar", "k_resample_polar_large", "k_te_rows_large", "k_rfc_pack", "k_obb_seed", "k_neck", "k_canal", "k_groove_rows", "k_groove_scale",
                "k_groove_rfc", "k_groove_tail", "k_anp_rows", "k_anp_edge_count", "k_anp_edges",
                "k_anp_plane", "k_rays_hit", "k_tail", "k_te_rows", "k_te_ends", "k_sphere_partial", "k_anp_scale", "fill", "k_stage_status", "k_te_rows_huge", "k_slice_link_huge", "k_resample_polar_huge", "k_ovf_plan", "k_slice_emit_ovf",
                "k_decode_bounds", "k_section_points"]


def unet_layers(base, depth, H, W, eb=4):
    """name -> (flops per image, bytes per image: activations in+out + weights); eb = bytes per activation/weight element."""
    ch = [base << i for i in range(depth + 1)]
    out = {}
    h, w = H, W

    def conv(name, cin, cout, h, w, taps=9, up=False):
        oh, ow = (2 * h, 2 * w) if up else (h, w)
        fl = 2 * (4 if up else taps) * h * w * cin * cout
        by = eb * (h * w * cin + oh * ow * cout + (4 if up else taps) * cin * cout)
        out["unet." + name] = (fl, by)

    cin = 1
    for i in range(depth):
        conv(f"enc{i}a", cin, ch[i], h, w)
        conv(f"enc{i}b", ch[i], ch[i], h, w)
        cin = ch[i]
        h, w = h // 2, w // 2
    conv("bota", ch[depth - 1], ch[depth], h, w)
    conv("botb", ch[depth], ch[depth], h, w)
    for i in reversed(range(depth)):
        conv(f"up{i}", ch[i + 1], ch[i], h, w, up=True)
        h, w = 2 * h, 2 * w
        conv(f"dec{i}a", 2 * ch[i], ch[i], h, w)
        conv(f"dec{i}b", ch[i], ch[i], h, w)
    out["unet.head"] = (2 * H * W * ch[0], H * W * (eb * ch[0] + 4))
    return out


def unet_couts(base, depth):
    ch = [base << i for i in range(depth + 1)]
    out = {}
    for i in range(depth):
        out[f"unet.enc{i}a"] = out[f"unet.enc{i}b"] = out[f"unet.up{i}"] = out[f"unet.dec{i}a"] = out[f"unet.dec{i}b"] = ch[i]
    out["unet.bota"] = out["unet.botb"] = ch[depth]
    return out


def unet_cins(base, depth):
    ch = [base << i for i in range(depth + 1)]
    out = {}
    for i in range(depth):
        out[f"unet.enc{i}a"] = ch[i - 1] if i else 1
        out[f"unet.enc{i}b"] = out[f"unet.dec{i}b"] = ch[i]
        out[f"unet.up{i}"] = ch[i + 1]
        out[f"unet.dec{i}a"] = 2 * ch[i]
    out["unet.bota"], out["unet.botb"] = ch[depth - 1], ch[depth]
    return out


def geom_bytes(B, V, F):
    """Per STEP (all launches of a kernel name in one SH_STAGE_ALL pass over B humeri): name -> (algorithmic bytes in this
    implementation's layout -- f64 geometry, 32-byte segments --, SURVEY 8(d) bytes for the same work in the survey's f32
    layout or None where the survey names no figure).  Plane sets per step: full 200, distal 200, proximal 600, neck contour 1;
    ~145 crossing segments per plane (28.5 k + 28.5 k + 88 k segments per humerus)."""
    s_full, s_dist, s_prox, s_neck = 28471.0, 28500.0, 88100.0, 150.0
    S = s_full + s_dist + s_prox + s_neck
    seg = 32
    return {
        "k_transform_verts": (B * V * (12 + 24), B * 2 * 12 * V),
        # four passes over the mesh (one per plane set) + the segments written (read back by the link kernels)
        "k_slice_emit": (B * (4 * (24 * V + 12 * F) + seg * S), B * (3 * (12 * V + 12 * F) + 16 * S)),
        # segments read; closed rings written for the distal / proximal / neck sets
        "k_slice_link": (B * (seg * S + 16 * (s_dist + s_prox + s_neck)), B * 16 * S),
        "k_slice_link_large": (0, None),
        # proximal rings read; written: the polar rows the later stages read (about the origin from plane 88 on, centred ones in the groove's
        # 330-plane range; round 3 wrote contour + both images of all 600 planes: 944 MB at B = 64) -- k_slices.h, RsWant
        "k_resample_polar": (B * (16 * s_prox + (512 + 330) * 2 * 512 * 8), B * (600 * 512 * 2 * 4 * 3)),
        "k_resample_polar_large": (0, None),
        "k_groove_rows": (B * 330 * (2 * 512 * 8 + 512 * 8), B * 330 * 2 * 512 * 4),
        "k_anp_rows": (B * 512 * (2 * 512 * 8 + 512 * 8), B * 512 * 2 * 512 * 4 * 2),      # (theta, r) rows read, the raw image written (round 3: + the shifted-theta image)
        "k_anp_scale": (B * 512 * 512 * (8 + 4), B * 512 * 512 * 4),
        "k_anp_edge_count": (B * 512 * 512 * 4, None),
        # the mask as bits (2 MB, written by k_anp_edge_count) instead of the logits again; the radii of the mask's pixels (about 0.6 of the image)
        "k_anp_edges": (B * 512 * 64, B * 512 * 512 * 4),
        "k_sphere_partial": (B * (512 * 64 + int(0.6 * 512 * 512) * 8), None),
        "k_rays_hit": (B * (24 * V + 12 * F), B * (12 * V + 12 * F)),      # every triangle once for the four rays
        "k_apply_csys": (B * V * (12 + 24), B * 2 * 12 * V),
        "k_obb_end_points": (B * (12 * V + 12 * F), None),
        "k_obb_bounds": (B * (1368 * 24 + 2732 * 32), None),            # the hull record once: compute bound, listed for completeness
        "k_obb_candidates": (B * (1368 * 24 + 2732 * 24 + 4098 * 16), None),
        "k_te_rows": (B * 37 * 300 * 16, None),
        "k_te_rows_large": (0, None),
    }


# kernel (timer) name -> stage of the hot path, for the per-stage device times of the single-humerus leg
def stage_of(name):
    if name.startswith("unet.") or name in ("k_pack_w_bf16",):
        return "anatomic_neck.unet"
    for pfx, st in (("k_hull", "obb"), ("k_obb", "obb"), ("host.", "obb.host"), ("k_transform_verts", "slices"), ("k_make_planes", "slices"),
                    ("k_init_bounds", "slices"), ("k_decode_bounds", "slices"), ("k_slice", "slices"), ("k_resample", "slices"),
                    ("k_neck", "surgical_neck"), ("k_canal", "canal"), ("k_groove", "bicipital_groove"), ("k_rfc", "bicipital_groove"),
                    ("k_anp", "anatomic_neck.geometry"), ("k_rays", "anatomic_neck.geometry"), ("k_te", "trans_epicondylar"),
                    ("k_tail", "csys"), ("fill", "other"), ("k_stage", "csys"), ("k_ovf", "slices"), ("k_sphere", "csys"), ("k_apply", "csys"), ("k_section", "csys")):
        if name.startswith(pfx):
            return st
    return "other"


# what the test suite asserts for the UNet arithmetic of a leg (tests/test_gpu_end_to_end.py::test_full_size_batch_equivariance
# on this bench's own batch, test_gpu_landmarks.py on the fixtures): every landmark that does not read the mask is within
# 1e-4 mm of the oracle in all three; the anatomic-neck landmarks move by single mask pixels on the 16-bit paths
PARITY_NOTE = {
    "f32": "every landmark within 1e-4 mm of the oracle, integer decisions equal, f32 logits bit-exact (north-star tolerance met)",
    "bf16": "mask-independent landmarks (OBB, canal, groove, TE, csys) within 1e-4 mm of the oracle; anatomic-neck landmarks within the bf16 "
            "bound asserted on this batch: plane point 0.3 mm, axis ends 1.5 mm, edge-point count +-60, neck-shaft / retroversion 2.5 deg, radius of curvature 0.05 mm (1e-4 mm is met by the f32_unet and f32x_unet legs)",
    "f16": "mask-independent landmarks within 1e-4 mm of the oracle; anatomic-neck landmarks within the f16 bound asserted on this batch: "
           "plane point 0.06 mm, axis ends 0.4 mm, edge-point count +-12, neck-shaft / retroversion 0.6 deg, radius of curvature 0.02 mm",
    "f32x": "f32 tensors, MFMA layers on split-f16 operands (3 MFMAs per product): logits within 3e-6 of the exact f32 path, mask and every landmark "
            "IDENTICAL to the f32 path's on the four fixtures and the 64 humeri of this batch (tests/test_gpu_unet_x3.py), hence within 1e-4 mm of the oracle",
}

UNET_ENUM = {"f32": 0, "bf16": 1, "f16": 2, "f32x": 3}


def sym_key(name, unet, cout, fused_net=True, raw_image=True):
    """Kernel symbol (as rocprofv3 --stats groups launches) of UNet layer `name`; element type spelled bf16 / f16 / f32."""
    if name in ("unet.head", "unet.enc0a", "unet.pool"):
        return name
    nt = 4 if cout % 64 == 0 else 2
    up = name.startswith("unet.up")
    from shoulder_amd import unet_spec
    cin = unet_cins(unet_spec.BASE, unet_spec.DEPTH).get(name, 0)
    if unet == "f32":
        return "k_conv_mfma_f32<%d,%d>" % (1 if up else 9, nt)
    if unet == "f32x":
        if up and cin in (64, 128, 256, 512):      # source pixels resident in registers (k_unet_x3.h)
            return "k_upconv_x3r<%d,%d>" % (cin // 32, {64: 4, 128: 4, 256: 2, 512: 1}[cin])
        fuse = 4 if name in ("unet.enc0b", "unet.enc1b", "unet.enc2b", "unet.enc3b") else 0
        return "k_conv_mfma_x3<%d,%d,%d>" % (1 if up else 9, nt, fuse)
    if fused_net:      # the production 16-bit network (shoulder_hip.hip: unet_forward16)
        if name == "unet.enc0b":
            return "k_enc0_pp<%s,%s>" % (unet, "true" if raw_image else "false")      # image -> enc0a -> enc0b -> skip0 + pool (k_unet16_pp.h)
        if name == "unet.dec0a":
            return "k_dec0a_up_pp<%s>" % unet      # up0 computed inside dec0a; no unet.up0 launch
        if name == "unet.dec0b":
            return "k_dec0b_head_pp<%s>" % unet      # dec0b + head: only the logits leave the kernel
        if up:
            if cin in (128, 256, 512):      # source pixels resident in MFMA fragments, a group's weights by LDS-DMA (k_unet16_up.h)
                return "k_upconv16g<%s,%d,%d,%d,%s>" % (unet, cin // 32, 4 if cin == 128 else 2, 2 if cin == 512 else 4, "false" if cin == 512 else "true")
            return "k_upconv16<%s>" % unet                           # 2x2 transposed conv, both column phases per workgroup
        if cout % 64 == 0:      # persistent LDS-DMA conv with loader waves (k_unet16_ldr.h): <EK, FUSE (UF_POOL = 4), weights resident>
            fuse = 4 if name in ("unet.enc1b", "unet.enc2b", "unet.enc3b") else 0
            wres = int(cout == 64 and (cin // 32) * 64 <= 128)
            return "k_conv3_ldr16<%s,%d,%d>" % (unet, fuse, wres)
    if up:
        return "k_conv_mfma16<%s,1,%d,0>" % (unet, nt)
    return "k_conv_mfma16<%s,9,%d,0>" % (unet, nt)


def rocprof_name(key):
    """bench symbol key -> the kernel name rocprofv3 prints (profiles/*.json are keyed by it).  The 16-bit kernels are
    templates on an element-kind integer (k_unet_bf16.h: 0 = __bf16, 1 = _Float16), spelled bf16 / f16 in bench keys."""
    return "sh::" + key.replace("<bf16", "<0").replace("<f16", "<1").replace(",", ", ")      # (bool template arguments print as true / false)


def host_info():
    model = ""
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                model = ln.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return {"cpu_model": model, "os_cpu_count": os.cpu_count(), "affinity_cores": len(os.sched_getaffinity(0)), "usable_cores": usable_cores()}


def usable_cores():
    """Hardware threads this process may use: affinity mask, capped by a cgroup-v2 CPU quota (as usable_threads() in shoulder_hip.hip)."""
    n = len(os.sched_getaffinity(0))
    try:
        q, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if q != "max" and int(period) > 0:
            n = min(n, max(1, -(-int(q) // int(period))))
    except (OSError, ValueError):
        pass
    return n


def hull_threads(world_local):
    """The worker pool of the host hull phase (HullPool, shoulder_hip.hip), one per process shared by its lanes: hardware threads / LOCAL_WORLD_SIZE, at most 32."""
    if os.environ.get("SHOULDER_HULL_THREADS"):
        return int(os.environ["SHOULDER_HULL_THREADS"])
    return max(1, min(32, len(os.sched_getaffinity(0)) // max(1, world_local)))      # (pool size: affinity mask; the hull-mode decision counts a CPU quota too)


def default_lanes(hull_mode, pipelined=True):
    """Engine contexts per GPU: two with the hull on the host, three with the hull on the device (its lanes are serial chains with a
    5.6 ms hull kernel on 64 CUs at their head) -- what eight ranks on one host get (sh_auto_hull_mode: 48 usable threads per rank needed)."""
    return (3 if hull_mode == "device" else 2) if pipelined else 1


def launch_ranks(n, argv, worker=None, timeout=None):
    """`python bench.py --gpus N` without a launcher around it (WORLD_SIZE unset): start the N rank processes of one node
    ourselves -- fresh children, one per GPU, with the environment torch.distributed.run would give them (RANK, LOCAL_RANK,
    WORLD_SIZE, LOCAL_WORLD_SIZE, MASTER_ADDR = 127.0.0.1, a free MASTER_PORT) -- BEFORE anything in this process touches
    the GPU (this process never does: it only waits).  Rank 0's stdout is forwarded as it is (the one JSON line); the other
    ranks' stdout goes to stderr.  Returns the exit code: 0 if every rank ended with 0, else the first non-zero one (the
    remaining ranks are terminated, so a failed rank does not leave the others waiting in a collective)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = list(worker) if worker else [sys.executable, os.path.abspath(__file__)]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "SH_BENCH_SELF_LAUNCHED": "1"})
        procs.append(subprocess.Popen(cmd + list(argv), env=env, stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr))
    rc = 0
    t_end = None if timeout is None else time.time() + timeout
    out0 = None
    import threading
    buf = []
    th = threading.Thread(target=lambda: buf.append(procs[0].stdout.read()), daemon=True)      # (a full pipe must not block rank 0)
    th.start()
    live = set(range(n))
    while live:
        for r in sorted(live):
            c = procs[r].poll()
            if c is not None:
                live.discard(r)
                if c != 0 and rc == 0:
                    rc = c
                    print(f"bench.py: rank {r} ended with exit code {c}; stopping the other ranks", file=sys.stderr)
                    for q in live:
                        procs[q].terminate()
        if t_end is not None and time.time() > t_end and live:
            rc = rc or 124
            for q in live:
                procs[q].kill()
        time.sleep(0.05)
    th.join(timeout=10)
    out0 = buf[0] if buf else b""
    if rc == 0:
        sys.stdout.write(out0.decode(errors="replace"))
        sys.stdout.flush()
    else:
        sys.stderr.write(out0.decode(errors="replace"))
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=64, help="humeri per GPU per step")
    ap.add_argument("--unet", choices=["f32", "bf16", "f16", "f32x"], default="bf16",
                    help="UNet arithmetic of the headline leg: bf16 = BASELINE configs[2]/[3] (throughput), f16 = configs[4]'s element type, "
                         "f32 = configs[1] parity path (bit-exact vs the oracle)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the f32-UNet leg and the one-lane leg that follow the headline leg at N=1")
    ap.add_argument("--cpu-meshes", type=int, default=4, help="humeri of the batch the single-core oracle leg processes")
    ap.add_argument("--cpu-pool", type=int, default=-1, help="worker processes of the pooled oracle leg (one humerus each); -1 = host cores of this process, at most 16; 0 = skip")
    ap.add_argument("--no-pipeline", action="store_true", help="wait for every step before enqueueing the next one (sh_run instead of sh_submit / sh_collect)")
    ap.add_argument("--lanes", type=int, default=0, help="engine contexts per GPU; steps alternate between them and their streams overlap on the device "
                    "(the small geometry kernels of one step fill the gaps of the other step's UNet); 1 = a single context; "
                    "0 (default) = 2 with the hull on the host, 3 with the hull on the device (its lanes start with a 5.6 ms hull kernel on 64 CUs)")
    ap.add_argument("--no-overlap", action="store_true", help="do not compute the host hulls of step k+1 during the device work of step k")
    ap.add_argument("--from-host", action="store_true", help="re-upload the batch from host memory every step (PCIe-inclusive rate)")
    ap.add_argument("--from-stl", action="store_true", help="every step starts from the bytes of B binary STL files (device-side parse + vertex merge, sh_upload_stl)")
    ap.add_argument("--sync-upload", action="store_true", help="with --from-host / --from-stl: hand every batch over with the synchronous sh_upload_* (round 2's schedule) instead of the staging side (sh_stage_* / sh_commit_staged)")
    ap.add_argument("--stream-batches", type=int, default=4, help="distinct batches the streaming legs cycle through (every step gets a batch different from the one its lane holds)")
    ap.add_argument("--record-rows", type=int, default=2560, help="anatomic-neck rows per record on the wire (sh_set_record_rows): records of 8 680 + 24 R bytes instead of the full "
                    "104 KB (4 096 padded rows; these humeri have ~2 300 edge points); 0 = full records.  Every other field is in both formats; n_anp keeps the true count")
    ap.add_argument("--hull", choices=["auto", "host", "device"], default="auto", help="where the convex hull of the OBB stage runs (sh_set_hull_mode); "
                    "auto = host quickhull where the rank has enough usable hardware threads (16 alone on its host, 48 per rank otherwise), else the device hull")
    ap.add_argument("--check-gather", action="store_true", help="after the timed region rank 0 compares the records the last gather delivered for its own shard with a run of its own engine")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under torch.distributed.run: this process becomes the launcher of its own N ranks and never initialises the GPU
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # HIP spreads a process's streams over GPU_MAX_HW_QUEUES hardware queues (default 4); two streams on one queue run in
    # order.  The lanes below need their streams on different queues next to torch's and RCCL's streams, so: more queues,
    # and the engine contexts are created before anything else makes streams.  (Must be set before HIP initialises.)
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    import torch
    # SH_BENCH_FORCE_DIST=1 exercises the collective code path with a single rank (1-GPU rehearsal)
    use_dist = world > 1 or os.environ.get("SH_BENCH_FORCE_DIST") == "1"
    if args.gpus != world and rank == 0:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}: measuring {world} rank(s)", file=sys.stderr)

    from shoulder_amd import _lib, synth, unet_spec
    from shoulder_amd import dist as shd
    from shoulder_amd.engine import Engine
    from shoulder_amd.stl import load_stl
    verts, faces = load_stl(os.path.join(ROOT, "tests", "golden", "bones", "humerus_left.stl"))   # host I/O, outside the timed region
    V, F, B = len(verts), len(faces), args.batch

    weights = unet_spec.make_teacher_weights()
    cpu = None
    if world == 1 and not args.no_cpu_baseline:
        # before anything touches the GPU: the pooled leg starts worker processes
        cpu = cpu_baseline(verts, faces, synth.similarity_transforms(B, verts, seed=1234), weights, args.cpu_meshes, args.cpu_pool)
    # Lanes: independent engine contexts (own stream, own scratch) on this GPU.  Step s runs on lane s % lanes, so the launch-
    # and latency-bound geometry kernels of one step execute beside the chip-filling UNet kernels of the previous step.
    eng = Engine(local)
    if args.hull != "auto":      # (auto: what the engine chose at creation -- SHOULDER_HULL, else by the host's threads per rank)
        eng.set_hull_mode(args.hull)
    lanes = args.lanes if (args.lanes > 0 and not args.no_pipeline) else default_lanes(eng.hull_mode, not args.no_pipeline)
    engs = [eng] + [Engine(local) for _ in range(lanes - 1)]
    REC = _lib.record_dtype(args.record_rows)
    for e in engs:
        e.set_record_rows(args.record_rows)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        torch.cuda.set_device(local)
        # RCCL prints a version banner on stdout when its first communicator comes up; stdout carries the one JSON line only
        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local))
            dist.barrier()
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    for e in engs:
        if rank == 0:
            e.load_rfc()
            e.load_unet(weights, unet_spec.BASE, unet_spec.DEPTH)
        else:   # allocate the parameter block with the right shapes, contents arrive by broadcast
            e.load_rfc()       # tree topology is needed host-side for validation; values are overwritten below
            e.load_unet({k: np.zeros_like(v) for k, v in weights.items()}, unet_spec.BASE, unet_spec.DEPTH)
    if use_dist:
        shd.broadcast_params(eng.param_block(), src=0, device=f"cuda:{local}")   # RCCL broadcast of UNet + forest parameters
        src_blk = shd.as_byte_tensor(eng.param_block(), device=f"cuda:{local}")
        for e in engs[1:]:                                                          # the other lanes copy the block device to device
            shd.as_byte_tensor(e.param_block(), device=f"cuda:{local}").copy_(src_blk)
        torch.cuda.synchronize()
        for e in engs:
            e.param_block_commit()      # host mirrors follow the device block (a later sh_load_* re-uploads from them)
    for e in engs:
        e.set_hull_mode(eng.hull_mode)
        if os.environ.get("SH_BENCH_KEEP_PRODUCTS") == "1":      # A/B: every proximal plane's resampled contour and polar rows written (round 3's traffic)
            e.set_keep_products(True)
        e.set_unet_turns((lanes > 1 and os.environ.get("SH_BENCH_NO_TURNS") != "1") or os.environ.get("SH_BENCH_FORCE_TURNS") == "1")      # (FORCE_TURNS: a single lane on the two-lane grid, i.e. without the CU reserve)

    T = synth.similarity_transforms(B, verts, seed=1234, start=rank * B)
    host_batch = None
    stl_blobs = None
    stream = None      # the PCIe-inclusive variants (never the headline `value`): a NEW batch is handed over for every step
    if args.from_stl or args.from_host:
        stream = make_stream("stl" if args.from_stl else "host", verts, faces, B, args.stream_batches, start=rank * B)
        if args.from_stl:
            stl_blobs = host_batch = stream["batches"][0]
            for e in engs:
                e.upload_stl(stl_blobs)
        else:
            host_batch = stream["batches"][0]
            for e in engs:
                e.upload_packed(host_batch)
        if args.sync_upload or args.no_pipeline:
            stream = None
    else:
        for e in engs:
            e.upload([(verts, faces)])
            e.synth_batch(T)                           # batch resident in HBM (of every lane) before the timed region

    # the records of a step go to rank 0 in the wire format (packed unless --record-rows 0): B x REC.itemsize bytes per rank and step
    lm_t = torch.zeros(B * REC.itemsize, dtype=torch.uint8, device=f"cuda:{local}") if use_dist else None
    gather_list = [torch.empty_like(lm_t) for _ in range(world)] if (use_dist and rank == 0) else None
    # pipelined schedule: every run writes its records (on its engine's stream) into one of the send buffers
    nsend = 2 * lanes
    send = [torch.empty_like(lm_t) for _ in range(nsend)] if use_dist else None
    send_free = [None] * nsend      # event recorded behind the gather that last read send[k]
    overlap = not args.no_overlap and host_batch is None

    def hand_over(e):      # the PCIe-inclusive variants give every step its batch again (a new batch voids prepared hulls)
        if stl_blobs is not None:
            e.upload_stl(stl_blobs)
        elif host_batch is not None:
            e.upload_packed(host_batch)

    leg_info = {"error_status": 0, "records": 0}

    def run_leg(leg_engs, unet, steps, warmup, pipelined, stream=None):
        """`warmup` untimed steps, then exactly `steps` timed steps of SH_STAGE_ALL on `leg_engs` (step s on lane s % len),
        bracketed by barrier + synchronize on both sides.  -> (seconds: max over ranks, per-layer HIP-event times of the region).
        `stream` (make_stream): step s works on batch s % n of the stream, handed over through the lane's staging side -- staged
        right behind the lane's previous submit (copies, device-side checks / STL merge and hulls run beside that step), committed
        when that step has been collected; every stage, commit, submit and collect of the K steps lies inside the timed region."""
        nl = len(leg_engs)
        nb_ = len(stream["batches"]) if stream else 0

        def stage_for(e, k):
            (e.stage_stl if stream["kind"] == "stl" else e.stage)(stream["batches"][k % nb_])
        for e in leg_engs:
            e.set_params(unet_dtype=UNET_ENUM[unet])
            e.set_overlap(False)
        for w_ in range(warmup):
            for e in (leg_engs if w_ == 0 else leg_engs[:1]):      # every lane's scratch is allocated before t0
                if stream:      # (the staging side's page-locked memory and device buffers as well)
                    stage_for(e, nb_ - 1 - w_)
                    e.commit_staged()
                else:
                    hand_over(e)
                e.run(_lib.STAGE_ALL, fetch=False)
            if use_dist:
                dist.gather(lm_t, gather_list, dst=0)
        # HIP events around the UNet layers only inside the timed region (level 2): the dominant kernel is one of them, and
        # events around all ~150 launches of a step stretch the step by ~0.7 ms.  The other kernels are timed in one extra
        # pass after the region (level 1).
        for e in leg_engs:
            e.enable_timing(2)
            e.reset_timers()
        # Streaming schedule: inside the timed region the hulls of step k+1 are computed by host threads while the device
        # works on step k.  Nothing is carried in from the warmup (discard) and nothing is prepared for a step K+1 (overlap
        # off before the last step): K hull passes and K device passes lie between t0 and t1.
        for e in leg_engs:
            e.discard_prepared()
            e.set_overlap(overlap and not stream)      # (a streamed batch runs once: its hulls are prepared by the staging side)
        leg_info.update(error_status=0, records=0)
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        if pipelined:
            # Steps in flight: step s is enqueued (sh_submit) on lane s % lanes while the steps before it still execute, so the
            # device goes from one step to the next without waiting for the host and the lanes' streams overlap; the records of a
            # step are collected (and gathered) once `depth` later steps are enqueued.  All K submits and all K collects lie
            # inside the timed region.
            depth = int(os.environ.get("SH_BENCH_DEPTH", "0")) or (nl if (nl > 1 or stream) else 2)      # steps in flight (an engine takes two; a lane that streams, one)
            trace_host = os.environ.get("SH_BENCH_TRACE") == "1" and rank == 0
            pend = []

            def finish(k, e):
                rec = e.collect()
                if rec is not None:
                    leg_info["error_status"] += int((rec["status"] != 0).sum())
                    leg_info["records"] += len(rec)
                if use_dist:
                    # asynchronous on the host: the gather is enqueued on torch's stream behind the finished run (collect waited
                    # for its event) and an event behind it says when send[k] may be overwritten -- the host goes straight on
                    # to the next submit, the ranks are coupled on the device side only
                    dist.gather(send[k % nsend], gather_list, dst=0)
                    ev = torch.cuda.Event()
                    ev.record()
                    send_free[k % nsend] = ev
            for s_ in range(steps):
                e = leg_engs[s_ % nl]
                if steps - s_ <= nl:
                    e.set_overlap(False)                           # the last step of a lane prepares nothing
                if len(pend) >= depth:
                    finish(*pend.pop(0))
                if stream:         # (the lane is idle here: its previous step was collected above)
                    if not e.staged:
                        stage_for(e, s_)      # the lane's first step of the region: nothing was staged ahead
                    e.commit_staged()
                else:
                    hand_over(e)
                ts_ = time.perf_counter()
                if use_dist:
                    if send_free[s_ % nsend] is not None:
                        send_free[s_ % nsend].synchronize()       # (2 * lanes steps old: long done)
                    e.submit(_lib.STAGE_ALL, fetch=False, out_ptr=send[s_ % nsend].data_ptr())
                else:
                    e.submit(_lib.STAGE_ALL, fetch="view")
                if trace_host:
                    print("[bench] step %d lane %d: submit returned after %.2f ms (at %.2f ms of the region)" % (s_, s_ % nl, 1e3 * (time.perf_counter() - ts_), 1e3 * (time.perf_counter() - t0)), file=sys.stderr)
                pend.append((s_, e))
                if stream and s_ + nl < steps:
                    stage_for(e, s_ + nl)      # the batch of this lane's NEXT step: copied, checked, merged and hulled beside this one
            for k_e in pend:
                finish(*k_e)
        else:
            e = leg_engs[0]
            for s_ in range(steps):
                if s_ == steps - 1:
                    e.set_overlap(False)
                hand_over(e)
                if use_dist:
                    e.submit(_lib.STAGE_ALL, fetch=False, out_ptr=lm_t.data_ptr())
                    e.collect()
                    dist.gather(lm_t, gather_list, dst=0)      # landmark records of every rank to rank 0 (device to device)
                else:
                    e.run(_lib.STAGE_ALL, fetch="view")        # records land in the engine's page-locked buffer
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        el = time.perf_counter() - t0
        if use_dist:
            tt = torch.tensor([el], dtype=torch.float64, device=f"cuda:{local}")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        ul_ = unet_layers(unet_spec.BASE, unet_spec.DEPTH, 512, 512, eb=4 if unet in ("f32", "f32x") else 2)

        def merged(name):       # (average ms, launches) over all lanes
            parts = [e.kernel_time_ms(name) for e in leg_engs]
            n = sum(p[1] for p in parts)
            return (sum(p[0] * p[1] for p in parts) / n if n else 0.0, n)
        region = {name: merged(name) for name in list(ul_) + ["unet.pool", "host.verts_d2h", "host.hull"]}
        for e in leg_engs:
            e.reset_timers()
            e.enable_timing(0)
        return el, region, ul_

    pipelined = not args.no_pipeline
    el, region_times, ul = run_leg(engs, args.unet, args.steps, args.warmup, pipelined, stream)

    eng.enable_timing(1)
    eng.run(_lib.STAGE_ALL, fetch=False)      # profiling pass outside the timed region, one lane alone: every kernel between events
    extra_times = {name: eng.kernel_time_ms(name) for name in GEOM_KERNELS + ["k_synth_batch", "k_pack_w_bf16"]}
    alone_times = {name: eng.kernel_time_ms(name) for name in ul}
    eng.enable_timing(0)

    gather_check = None
    if use_dist and rank == 0:      # what rank 0 holds after the last gather: every rank's records, in rank order
        lm = torch.cat(gather_list).cpu().numpy().view(REC)
        assert len(lm) == world * B
        if args.check_gather:
            own = eng.run(_lib.STAGE_ALL)
            gather_check = {"records": int(len(lm)), "own_shard_equal_to_local_run": bool(lm[:B].tobytes() == own.tobytes())}
    else:
        lm = eng.run(_lib.STAGE_ALL)
    n_bad = int((lm["status"] != 0).sum())

    # Extra legs, N=1 only (the driver's SCALE runs keep to the headline leg): the tolerance-conformant configuration -- f32
    # UNet, landmarks within 1e-4 mm of the oracle (tests/test_gpu_end_to_end.py) -- timed by the same protocol in the same
    # invocation, and the headline configuration on ONE lane (what a caller without the two-context schedule gets).
    extra = {}
    if world == 1 and not use_dist and not args.no_extra_legs and host_batch is None:
        k32 = max(3, min(args.steps, 10))
        el32, reg32, ul32 = run_leg(engs, "f32", k32, 1, pipelined)
        c32 = unet_couts(unet_spec.BASE, unet_spec.DEPTH)
        g = {}
        for name, (ms, n) in reg32.items():
            if name in ul32 and n and name not in ("unet.head", "unet.enc0a"):
                a = g.setdefault(sym_key(name, "f32", c32[name]), [0.0, 0.0, 0])
                a[0] += ms * n; a[1] += ul32[name][0] * B * n; a[2] += n
        dom32 = max(g, key=lambda k: g[k][0]) if g else None
        extra["f32_unet"] = {"value": round(B * k32 / el32, 3), "unit": "meshes/s", "steps": k32, "warmup": 1, "ms_per_step": round(1e3 * el32 / k32, 3),
                             "dtype": "f64 geometry + f32 UNet", "lanes": lanes,
                             "parity": "landmarks within 1e-4 mm of the oracle, f32 logits bit-exact (tests/test_gpu_end_to_end.py, test_gpu_landmarks.py)"}
        if dom32:
            ach = g[dom32][1] / (g[dom32][0] * 1e-3) / 1e12
            extra["f32_unet"]["roofline"] = {"bound": "mfma", "kernel": dom32, "achieved": round(ach, 2), "peak": PEAK_MFMA_F32_TF, "unit": "TFLOP/s",
                                             "frac": round(ach / PEAK_MFMA_F32_TF, 4), "avg_ms": round(g[dom32][0] / g[dom32][2], 4), "launches": g[dom32][2]}
        # SH_UNET_F32X: the tolerance-conformant configuration on the 16-bit matrix pipe (split-f16 operands, k_unet_x3.h), same protocol;
        # the records of its last step against the exact f32 path's, humerus by humerus (mask-derived counts and every coordinate)
        kx = max(3, min(args.steps, 10))
        elx, regx, ulx = run_leg(engs, "f32x", kx, 1, pipelined)
        lmx = engs[0].run(_lib.STAGE_ALL).copy()
        engs[0].set_params(unet_dtype=UNET_ENUM["f32"])
        lm32 = engs[0].run(_lib.STAGE_ALL).copy()
        gx = {}
        for name, (ms, n) in regx.items():
            if name in ulx and n and name not in ("unet.head", "unet.enc0a"):
                a = gx.setdefault(sym_key(name, "f32x", c32[name]), [0.0, 0.0, 0])
                a[0] += ms * n; a[1] += ulx[name][0] * B * n; a[2] += n
        domx = max(gx, key=lambda k: gx[k][0]) if gx else None
        dev = max(float(np.abs(lmx[k] - lm32[k]).max()) for k in ("anp_plane_point", "anp_axis_normal", "anp_axis_central", "te_axis", "csys", "canal_axis", "groove_axis"))
        extra["f32x_unet"] = {"value": round(B * kx / elx, 3), "unit": "meshes/s", "steps": kx, "warmup": 1, "ms_per_step": round(1e3 * elx / kx, 3),
                              "dtype": "f64 geometry + f32 tensors, split-f16 MFMA UNet", "lanes": lanes, "speedup_vs_f32_unet": round((B * kx / elx) / (B * k32 / el32), 3),
                              "humeri_with_the_f32_paths_edge_point_count": int((lmx["n_anp"] == lm32["n_anp"]).sum()), "humeri": int(B),
                              "max_landmark_deviation_from_f32_path_mm": dev, "parity": PARITY_NOTE["f32x"]}
        if domx:
            achx = gx[domx][1] / (gx[domx][0] * 1e-3) / 1e12      # algorithmic (f32-equivalent) flops: the kernel issues three 16-bit MFMAs per product
            extra["f32x_unet"]["roofline"] = {"bound": "mfma", "kernel": domx, "achieved": round(achx, 2), "peak": round(PEAK_MFMA_BF16_TF / 3.0, 1), "unit": "TFLOP/s",
                                              "frac": round(achx / (PEAK_MFMA_BF16_TF / 3.0), 4), "avg_ms": round(gx[domx][0] / gx[domx][2], 4), "launches": gx[domx][2],
                                              "peak_note": "dense f16 MFMA peak / 3: every product costs three MFMAs"}
        if args.unet == "bf16":      # the other 16-bit element type: same kernels and rate, 11 significant bits instead of 8
            k16 = max(3, min(args.steps, 10))
            el16, _, _ = run_leg(engs, "f16", k16, 1, pipelined)
            extra["f16_unet"] = {"value": round(B * k16 / el16, 3), "unit": "meshes/s", "steps": k16, "warmup": 1, "ms_per_step": round(1e3 * el16 / k16, 3),
                                 "dtype": "f64 geometry + f16 UNet", "lanes": lanes,
                                 "parity": "mask-independent landmarks within 1e-4 mm of the oracle; anatomic-neck landmarks within the f16 bound of tests/test_gpu_end_to_end.py (plane point 0.06 mm, axes 0.4 mm)"}
        if eng.hull_mode == "host" and pipelined:
            # the same step with the hull on the device (k_hull.h): nothing goes to the host or comes back from it.  Its lanes are
            # serial chains with a 6.5 ms hull kernel at the head, so three of them are needed to keep the chip busy.
            dev_engs = [Engine(local) for _ in range(3)]
            for e in dev_engs:
                e.set_record_rows(args.record_rows)
                e.load_rfc()
                e.load_unet(weights, unet_spec.BASE, unet_spec.DEPTH)
                e.set_hull_mode("device")
                e.set_unet_turns(True)
                e.upload([(verts, faces)])
                e.synth_batch(T)
            kd = 3 * max(8, min(args.steps, 20))      # (three lanes fill and drain inside the timed region: more steps than the other legs)
            eld, regd, _ = run_leg(dev_engs, args.unet, kd, 1, pipelined)
            lmd = dev_engs[0].run(_lib.STAGE_ALL)
            same = all(np.array_equal(lmd[k], lm[k]) for k in ("obb_transform", "canal_axis", "te_axis", "groove_axis", "csys", "n_anp"))
            extra["device_hull"] = {"value": round(B * kd / eld, 3), "unit": "meshes/s", "steps": kd, "warmup": 1, "ms_per_step": round(1e3 * eld / kd, 3),
                                    "lanes": 3, "host_ms_per_step": {k: round(regd[k][0] * regd[k][1] / kd, 3) for k in ("host.verts_d2h", "host.hull")},
                                    "records_equal_to_host_hull_run": bool(same)}
            for e in dev_engs:
                e.close()
        elif eng.hull_mode == "device" and pipelined:
            # the headline ran with the hull on the device (what `auto` picks when this rank has too few usable hardware
            # threads, e.g. eight ranks on one host or a small CPU quota): the same step with the host quickhull (two lanes, hulls of step k+1 computed by the
            # host's worker threads behind the device work of step k), and whether both modes give the same records
            host_engs = [Engine(local) for _ in range(2)]
            for e in host_engs:
                e.set_record_rows(args.record_rows)
                e.load_rfc()
                e.load_unet(weights, unet_spec.BASE, unet_spec.DEPTH)
                e.set_hull_mode("host")
                e.set_unet_turns(True)
                e.upload([(verts, faces)])
                e.synth_batch(T)
            kh = max(8, min(args.steps, 20))
            elh, regh, _ = run_leg(host_engs, args.unet, kh, 1, pipelined)
            lmh = host_engs[0].run(_lib.STAGE_ALL)
            same = all(np.array_equal(lmh[k], lm[k]) for k in ("obb_transform", "canal_axis", "te_axis", "groove_axis", "csys", "n_anp"))
            extra["host_hull"] = {"value": round(B * kh / elh, 3), "unit": "meshes/s", "steps": kh, "warmup": 1, "ms_per_step": round(1e3 * elh / kh, 3),
                                  "lanes": 2, "hull_threads_per_process": hull_threads(int(os.environ.get("LOCAL_WORLD_SIZE", "1"))),
                                  "host_ms_per_step": {k: round(regh[k][0] * regh[k][1] / kh, 3) for k in ("host.verts_d2h", "host.hull")},
                                  "records_equal_to_device_hull_run": bool(same)}
            for e in host_engs:
                e.close()
        extra["single_humerus_f32"] = single_humerus_leg(local, weights, verts, faces)
        if lanes > 1 and pipelined:
            # the headline configuration over a timed region ten times as long: what the lanes sustain once they are full.  `value`
            # above times exactly --steps steps from idle lanes -- the first hull phase (no device work to hide behind) and the
            # first geometry chain are inside it, ~7 ms that 20 steps carry as 0.35 ms each
            ks = 10 * args.steps
            els, _, _ = run_leg(engs, args.unet, ks, 2, pipelined)
            extra["steady_state"] = {"value": round(B * ks / els, 3), "unit": "meshes/s", "steps": ks, "warmup": 2, "ms_per_step": round(1e3 * els / ks, 3),
                                     "lanes": lanes, "note": "same configuration as the headline; 10 x its steps in one timed region"}
        if lanes > 1:
            # a context of its own, alone on the device: no turns to take, no CUs to leave to another lane (what `--lanes 1` runs)
            e1 = Engine(local)
            e1.set_record_rows(args.record_rows)
            e1.load_rfc()
            e1.load_unet(weights, unet_spec.BASE, unet_spec.DEPTH)
            e1.set_hull_mode(eng.hull_mode)
            e1.upload([(verts, faces)])
            e1.synth_batch(T)
            el1, _, _ = run_leg([e1], args.unet, args.steps, 1, pipelined)
            e1.close()
            extra["one_lane"] = {"value": round(B * args.steps / el1, 3), "unit": "meshes/s", "steps": args.steps, "warmup": 1,
                                 "ms_per_step": round(1e3 * el1 / args.steps, 3), "lanes": 1}

        if pipelined:
            # A stream of NEW batches (the reference's unit of work is a new STL): every step gets a batch its lane does not hold --
            # page-locked staging, H2D, device-side checks (from_host) or parse + vertex merge (from_stl), hull prefilter and host
            # hulls of step k + 2 all run beside step k of the same lane (sh_stage_* / sh_commit_staged); PCIe inclusive, never `value`.
            for kind in ("host", "stl"):
                st_engs = [Engine(local) for _ in range(2)]
                strm = make_stream(kind, verts, faces, B, args.stream_batches)
                for e in st_engs:
                    e.set_record_rows(args.record_rows)
                    e.load_rfc()
                    e.load_unet(weights, unet_spec.BASE, unet_spec.DEPTH)
                    e.set_hull_mode(eng.hull_mode)
                    e.set_unet_turns(True)
                    (e.upload_stl if kind == "stl" else lambda b: e.upload_packed(b))(strm["batches"][-1])
                el_s, _, _ = run_leg(st_engs, args.unet, args.steps, 1, True, strm)
                extra["from_" + kind] = {"value": round(B * args.steps / el_s, 3), "unit": "meshes/s", "steps": args.steps, "warmup": 1,
                                         "ms_per_step": round(1e3 * el_s / args.steps, 3), "lanes": 2, "distinct_batches": len(strm["batches"]),
                                         "records": leg_info["records"], "meshes_with_error_status": leg_info["error_status"],
                                         "input": ("B binary STL images per step (104 MB): page-locked staging, H2D, parse + vertex merge on the device" if kind == "stl" else
                                                   "packed float32 vertices + int32 faces per step (37 MB, pageable caller memory): page-locked staging, H2D, checks on the device")
                                                  + "; staged beside the lane's previous step, all inside the timed region"}
                for e in st_engs:
                    e.close()
                del strm

    if rank == 0:
        ul_cout = unet_couts(unet_spec.BASE, unet_spec.DEPTH)
        gb = geom_bytes(B, V, F)
        times = {}
        host_ms = {k: (lambda t: t[0] * t[1] / args.steps)(region_times[k]) for k in ("host.verts_d2h", "host.hull")}
        for name in list(ul) + ["unet.pool"]:                 # averaged over the launches of the timed steps
            ms, n = region_times[name]
            if n:
                times[name] = (ms, n)
        for name, (ms, n) in extra_times.items():              # one pass: scaled to the same per-step accounting
            if n:
                times[name] = (ms, n * args.steps)
        total_dev = sum(ms * n for ms, n in times.values())
        # group the launches by kernel symbol (what rocprofv3 --stats reports)
        sym, sym_of = {}, {}
        for name, (ms, n) in times.items():
            key = sym_key(name, args.unet, ul_cout[name]) if (name in ul and name in ul_cout) else name
            sym_of[name] = key
            g = sym.setdefault(key, dict(ms=0.0, n=0, flops=0.0, bytes=0.0))
            g["ms"] += ms * n
            g["n"] += n
            if name in ul:
                g["flops"] += ul[name][0] * B * n
                g["bytes"] += ul[name][1] * B * n
            else:
                g["bytes"] += gb.get(name, (0, None))[0] * (n / max(1, extra_times.get(name, (0, 1))[1]))      # (per-step bytes x steps)
        dom = max(sym, key=lambda k: sym[k]["ms"]) if sym else None
        roof = None
        if dom:
            g = sym[dom]
            common = dict(kernel=dom, avg_ms=round(g["ms"] / g["n"], 4), launches=g["n"], share_of_device_time=round(g["ms"] / total_dev, 3), traffic=None)
            if g["flops"] > 0:
                peak = mfma_peak_tf(args.unet)
                ach = g["flops"] / (g["ms"] * 1e-3) / 1e12
                roof = dict(bound="mfma", achieved=round(ach, 2), peak=peak, unit="TFLOP/s", frac=round(ach / peak, 4),
                            algorithmic_gbytes_per_launch=round(g["bytes"] / g["n"] / 1e9, 4), **common)
                if lanes > 1:
                    # the same launches with one lane alone on the device (profiling pass after the timed region): what the
                    # kernel does when no other stream shares the CUs with it
                    members = [nm for nm in ul if nm in times and sym_of.get(nm) == dom]
                    a_ms = sum(alone_times[nm][0] * alone_times[nm][1] for nm in members)
                    a_fl = sum(ul[nm][0] * B * alone_times[nm][1] for nm in members)
                    if a_ms > 0:
                        roof["one_lane_alone"] = dict(avg_ms=round(a_ms / sum(alone_times[nm][1] for nm in members), 4),
                                                      achieved=round(a_fl / (a_ms * 1e-3) / 1e12, 2), frac=round(a_fl / (a_ms * 1e-3) / 1e12 / peak, 4),
                                                      note="HIP events around every launch of ONE profiling pass on a grid of all 256 CUs; an event pair spans the launch's "
                                                           "dispatch as well (~10-20 us of a ~0.3 ms launch), so rocprofv3's kernel-trace duration of the same launches "
                                                           "(profiles/*_kernel_stats_*_one_lane.csv) reads 5-8 % shorter")
                        for rnd in ROUNDS:      # the same symbol in the committed one-lane kernel trace
                            csvp = os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats_b{B}_{args.unet}_one_lane.csv")
                            if os.path.exists(csvp):
                                import csv
                                want = rocprof_name(dom).replace("sh::", "")
                                for row in csv.DictReader(open(csvp)):
                                    if want in row["Name"]:
                                        r_ms = float(row["AverageNs"]) * 1e-6
                                        roof["one_lane_alone"].update(rocprof_avg_ms=round(r_ms, 4), rocprof_frac=round(a_fl / sum(alone_times[nm][1] for nm in members) / (r_ms * 1e-3) / 1e12 / peak, 4),
                                                                      rocprof_source=os.path.relpath(csvp, ROOT))
                                        break
                                break
            else:
                ach = g["bytes"] / (g["ms"] * 1e-3) / 1e9
                roof = dict(bound="hbm", achieved=round(ach, 2), peak=PEAK_HBM_GBS, unit="GB/s", frac=round(ach / PEAK_HBM_GBS, 4), **common)
        # HBM traffic and MFMA-busy fraction of the dominant kernel from the committed PMC profiles of this same command
        # (rocprofv3 cannot run inside the bench); null when no profile matches the configuration
        if roof:
            for rnd in ROUNDS:
                pmc_path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_b{B}_{args.unet}.json")
                if os.path.exists(pmc_path):
                    pk = json.load(open(pmc_path))["kernels"]
                    key = rocprof_name(dom)
                    if key in pk:
                        roof["traffic"] = pk[key]["hbm_bytes_per_launch"]
                        roof["traffic_source"] = os.path.relpath(pmc_path, ROOT)
                        break
            for rnd in ROUNDS:
                sq_path = os.path.join(ROOT, "profiles", f"{rnd}_pmc_sq_b{B}_{args.unet}.json")
                if os.path.exists(sq_path):
                    pk = json.load(open(sq_path))["kernels"]
                    key = rocprof_name(dom)
                    if key in pk and "mfma_busy_frac" in pk[key]:
                        roof["mfma_busy_frac"] = pk[key]["mfma_busy_frac"]
                        roof["mfma_busy_source"] = os.path.relpath(sq_path, ROOT)
                        break
        # Level 0 of the network (the three fused full-resolution kernels): algorithmic flops and bytes per launch, time inside the timed
        # region, and the fraction of their COMBINED roofline -- the larger of flops / MFMA peak and bytes / HBM peak is the time the
        # launch cannot beat (these kernels sit furthest below any roofline)
        level0 = None
        if args.unet in ("bf16", "f16") and all(k in times for k in ("unet.enc0b", "unet.dec0a", "unet.dec0b")) and "unet.up0" not in times and "unet.enc0a" not in times:
            HW = 512 * 512
            spec = {"unet.enc0b": ("enc0a + enc0b + 2x2 pool", ul["unet.enc0a"][0] + ul["unet.enc0b"][0], HW * (4 + 2 * 32 + 2 * 32 // 4)),
                    "unet.dec0a": ("up0 + dec0a", ul["unet.up0"][0] + ul["unet.dec0a"][0], HW * (2 * 32 + 2 * 64 // 4 + 2 * 32)),
                    "unet.dec0b": ("dec0b + head", ul["unet.dec0b"][0] + ul["unet.head"][0], HW * (2 * 32 + 4))}
            level0 = {"kernels": {}}
            l0_ms = l0_floor = 0.0
            for nm, (what, fl, by) in spec.items():
                ms = times[nm][0]
                t_mfma, t_hbm = fl * B / (PEAK_MFMA_BF16_TF * 1e12) * 1e3, by * B / (PEAK_HBM_GBS * 1e9) * 1e3
                level0["kernels"][sym_of.get(nm, nm)] = {"layers": what, "gflop_per_launch": round(fl * B / 1e9, 1), "gbytes_per_launch": round(by * B / 1e9, 3), "avg_ms": round(ms, 4),
                                                         "frac_of_mfma_peak": round(t_mfma / ms, 3), "frac_of_hbm_peak": round(t_hbm / ms, 3),
                                                         "frac_of_combined_roofline": round(max(t_mfma, t_hbm) / ms, 3)}
                l0_ms += ms
                l0_floor += max(t_mfma, t_hbm)
            level0.update(ms_per_pass=round(l0_ms, 3), frac_of_combined_roofline=round(l0_floor / l0_ms, 3),
                          share_of_unet_ms=round(l0_ms / sum(times[k][0] for k in times if k.startswith("unet.")), 3))
        unet_ms = sum(times[k][0] for k in times if k.startswith("unet."))
        unet_tf = sum(v[0] for v in ul.values()) * B / (unet_ms * 1e-3) / 1e12 if unet_ms else None
        top = sorted(((k, round(v["ms"] / args.steps, 3)) for k, v in sym.items()), key=lambda kv: -kv[1])[:10]
        # Per-kernel table of the geometry stage, per STEP: every launch of a kernel name in one step (the four plane sets go
        # through k_slice_emit / k_slice_link, the large-capacity instantiations have names of their own) -> time per step,
        # this implementation's algorithmic bytes for exactly those launches, SURVEY 8(d)'s bytes where it names them, and the
        # memory-side bytes rocprofv3 counted for the same kernel (profiles/, null without a matching profile).
        pmc_geom = {}
        for rnd in ROUNDS:
            pth = os.path.join(ROOT, "profiles", f"{rnd}_pmc_traffic_b{B}_{args.unet}.json")
            if os.path.exists(pth):
                pmc_geom = json.load(open(pth))["kernels"]
                break
        geom_tab = {}
        for name, (ms, n) in extra_times.items():
            if n and ms > 0 and name in gb and gb[name][0] > 0:
                per_step = ms * n
                gbs = gb[name][0] / (per_step * 1e-3) / 1e9
                row = {"ms_per_step": round(per_step, 4), "launches_per_step": n, "algorithmic_mb_per_step": round(gb[name][0] / 1e6, 2),
                       "gb_per_s": round(gbs, 1), "frac_of_hbm_peak": round(gbs / PEAK_HBM_GBS, 4)}
                if gb[name][1]:
                    row["survey_mb_per_step"] = round(gb[name][1] / 1e6, 2)
                    row["frac_of_hbm_peak_survey_bytes"] = round(gb[name][1] / (per_step * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
                pk = pmc_geom.get("sh::" + name)
                row["pmc_mb_per_step"] = round(pk["hbm_bytes_per_launch"] * pk.get("launches_per_step", n) / 1e6, 2) if pk and "hbm_bytes_per_launch" in pk else None
                geom_tab[name] = row
        geom_ms = sum(ms * n for name, (ms, n) in extra_times.items() if n and not name.startswith("unet.") and name not in ("k_synth_batch", "k_pack_w_bf16"))
        value = world * B * args.steps / el
        out = {"metric": "humerus meshes/s end-to-end (all 4 landmarks)", "value": round(value, 3), "unit": "meshes/s", "n_gpus": world,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 3), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f64 geometry + " + args.unet + " UNet",
               "data": "synthetic (similarity copies of humerus_left.stl, seed 1234; seeded teacher UNet weights)",
               "config": {"workload": f"BASELINE configs[{2 if world == 1 else 3}]: batch {B} synthetic humeri per GPU, all stages (SH_STAGE_ALL incl. the batch's apply_csys)",
                          "batch_per_gpu": B, "global_batch": B * world, "vertices": V, "triangles": F, "unet": f"base{unet_spec.BASE} depth{unet_spec.DEPTH} 512x512",
                          "parallelism": f"dp{world}", "input": "binary STL bytes every step (device parse + merge, PCIe inclusive)" if args.from_stl else "host buffers re-uploaded every step (PCIe inclusive)" if args.from_host else "resident in HBM",
                          "lanes": lanes, "hull": eng.hull_mode, "hull_threads_per_process": hull_threads(int(os.environ.get("LOCAL_WORLD_SIZE", "1"))), "host": host_info(),
                          "schedule": ((f"{lanes} engine contexts per GPU, step s on lane s % {lanes}, their streams overlap on the device (sh_submit / sh_collect); " if lanes > 1 else "two steps in flight (sh_submit / sh_collect); ") if pipelined else "") + ("host hulls of step k+1 overlap the device work of step k; pipeline filled and drained inside the timed region" if overlap else "serial hulls"),
                          "records": {"format": "packed (sh_set_record_rows)" if args.record_rows else "full sh_landmarks", "anp_rows": args.record_rows or 4096, "bytes_per_record": REC.itemsize,
                                      "bytes_per_step_to_rank0": world * B * REC.itemsize, "humeri_with_more_points_than_rows": int((lm["n_anp"] > (args.record_rows or 4096)).sum())},
                          "meshes_with_error_status": n_bad},
               "parity": PARITY_NOTE[args.unet],
               "roofline": roof, "cpu_baseline": cpu}
        out.update(extra)
        if level0 is not None:
            out["level0"] = level0
        if gather_check is not None:
            out["gather_check"] = gather_check
        out.update({"unet_tflops": None if unet_tf is None else round(unet_tf, 2), "device_ms_per_step_top": top,
                    "device_ms_per_step_total": round(total_dev / args.steps, 3), "host_ms_per_step": {k: round(v, 3) for k, v in host_ms.items()},
                    "geometry_ms_per_step_one_lane": round(geom_ms, 3),
                    "geometry_launches_per_step": int(sum(n for name, (ms, n) in extra_times.items() if n and not name.startswith("unet.") and name not in ("k_synth_batch", "k_pack_w_bf16"))),
                    "geometry_kernels": geom_tab})
        print(json.dumps(out))
    for e in engs:
        e.close()
    if use_dist:
        dist.destroy_process_group()


def single_humerus_leg(device, weights, verts, faces, reps=10):
    """BASELINE configs[1]: ONE humerus (tests/golden/bones/humerus_left.stl), full landmark set, f32 UNet -- latency, not
    throughput.  (a) `Engine.run(STAGE_ALL)` on the resident mesh: wall ms per humerus (median of `reps`, each run waited for),
    with the device time per stage from HIP events around every launch of one extra run; (b) the reference README's flow through
    the facade (bone.py:110-157): `Humerus(stl)` -> `apply_csys_canal_transepiconylar()` -> the four accessors, from the file on
    disk, wall ms (median)."""
    from shoulder_amd import _lib, bone, unet_spec
    from shoulder_amd.engine import Engine
    e = Engine(device)
    e.load_rfc()
    e.load_unet(weights, unet_spec.BASE, unet_spec.DEPTH)
    e.set_params(unet_dtype=_lib.UNET_F32)
    e.upload([(verts, faces)])
    for _ in range(2):
        e.run(_lib.STAGE_ALL)
    walls = []
    for _ in range(reps):
        t0 = time.perf_counter()
        lm = e.run(_lib.STAGE_ALL)
        walls.append(1e3 * (time.perf_counter() - t0))
    # the same humerus with SH_UNET_F32X (mask identical to the exact path's on every image compared so far; not the facade's default:
    # only the exact f32 chain is GUARANTEED to give the oracle's mask)
    e.set_params(unet_dtype=_lib.UNET_F32X)
    e.run(_lib.STAGE_ALL)
    walls_x = []
    for _ in range(reps):
        t0 = time.perf_counter()
        lmx = e.run(_lib.STAGE_ALL)
        walls_x.append(1e3 * (time.perf_counter() - t0))
    same_x = bool(lmx.tobytes() == lm.tobytes())
    e.set_params(unet_dtype=_lib.UNET_F32)
    e.enable_timing(1)
    e.reset_timers()
    e.run(_lib.STAGE_ALL)
    stages = {}
    for name in GEOM_KERNELS + list(unet_layers(unet_spec.BASE, unet_spec.DEPTH, 512, 512)) + ["unet.pool", "host.hull", "host.verts_d2h"]:
        ms, n = e.kernel_time_ms(name)
        if n:
            st = stage_of(name)
            stages[st] = stages.get(st, 0.0) + ms * n
    e.enable_timing(0)
    ok = int(lm["status"][0]) == 0
    e.close()
    # the facade on its process-wide default engine (created here with the same parameters)
    import warnings
    stl = os.path.join(ROOT, "tests", "golden", "bones", "humerus_left.stl")
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        de = bone.default_engine(device, unet_weights=weights)
    de.set_params(unet_dtype=_lib.UNET_F32)
    fw = []
    for i in range(1 + max(3, reps // 2)):
        t0 = time.perf_counter()
        h = bone.Humerus(stl)
        h.apply_csys_canal_transepiconylar()
        got = (h.canal.axis(), h.trans_epiconylar.axis(), h.anatomic_neck.points(), h.bicipital_groove.axis())
        if i:                                               # (the first pass allocates the engine's scratch)
            fw.append(1e3 * (time.perf_counter() - t0))
    med = lambda a: float(sorted(a)[len(a) // 2])
    return {"workload": "BASELINE configs[1]: single humerus (humerus_left.stl, 16 222 vertices), full landmark set, f32 UNet, 1 GPU",
            "engine_run_ms": round(med(walls), 3), "engine_run_ms_min": round(min(walls), 3), "reps": reps, "status_ok": ok,
            "engine_run_ms_f32x": round(med(walls_x), 3), "f32x_record_equal_to_f32": same_x,
            "device_ms_by_stage": {k: round(v, 3) for k, v in sorted(stages.items())}, "device_ms_total": round(sum(v for k, v in stages.items() if not k.endswith(".host")), 3),
            "facade_readme_flow_ms": round(med(fw), 3), "facade_readme_flow_ms_min": round(min(fw), 3),
            "facade_flow": "Humerus(stl) -> apply_csys_canal_transepiconylar() -> canal.axis(), trans_epiconylar.axis(), anatomic_neck.points(), bicipital_groove.axis(); file read + parse included",
            "landmarks_checked": int(len(got)), "parity": PARITY_NOTE["f32"]}


def make_stream(kind, verts, faces, B, nbatches, start=0):
    """`nbatches` DISTINCT batches of B humeri (similarity copies of the template, seed 1234, transforms start + k B ...), as the
    caller of a streaming pipeline would hold them in host memory: kind "host" -> the four packed arrays of sh_stage_meshes
    (pageable NumPy memory), kind "stl" -> B binary STL images per batch.  Built outside every timed region."""
    from shoulder_amd import synth
    from shoulder_amd.engine import Engine
    out = []
    for k in range(nbatches):
        T = synth.similarity_transforms(B, verts, seed=1234, start=start + k * B)
        meshes = [(synth.apply_similarity(T[i], verts), faces) for i in range(B)]
        out.append([stl_bytes(v, f) for v, f in meshes] if kind == "stl" else Engine.pack_meshes(meshes))
    return {"kind": kind, "batches": out}


def stl_bytes(verts, faces):
    """Binary STL image of a triangle mesh (float32, zero normals), as a scanner export would hand it over."""
    import struct
    tri = np.asarray(verts, dtype="<f4")[np.asarray(faces)]                      # (F, 3, 3)
    rec = np.zeros(len(tri), dtype=np.dtype([("n", "<f4", (3,)), ("v", "<f4", (3, 3)), ("a", "<u2")]))
    rec["v"] = tri
    return b"\0" * 80 + struct.pack("<I", len(tri)) + rec.tobytes()


def _oracle_one(i):
    """Worker of the pooled CPU leg: humerus i of the synthetic batch through the oracle; returns its wall time."""
    from threadpoolctl import threadpool_limits
    from oracle import rfc
    from oracle.humerus import OracleHumerus
    from shoulder_amd import synth, unet_spec
    from shoulder_amd.stl import load_stl
    verts, faces = load_stl(os.path.join(ROOT, "tests", "golden", "bones", "humerus_left.stl"))
    tab = rfc.load_tables(os.path.join(ROOT, "shoulder_amd", "models", "rfc_bg3.npz"))
    weights = unet_spec.make_teacher_weights()
    T = synth.similarity_transforms(i + 1, verts, seed=1234)      # prefix-stable: row i is humerus i of any batch
    t0 = time.perf_counter()
    with threadpool_limits(limits=1):
        OracleHumerus(synth.apply_similarity(T[i], verts), faces, tab, weights, unet_eval="f64").landmarks()
    return time.perf_counter() - t0


def cpu_baseline(verts, faces, T, weights, n_meshes, pool):
    """The oracle (port of the reference's CPU path) on the first n_meshes synthetic humeri with 1 thread, and -- SURVEY
    8(d)(ii) -- one humerus per worker process on the host cores this process may use."""
    from threadpoolctl import threadpool_limits
    from oracle import rfc
    from oracle.humerus import OracleHumerus
    from shoulder_amd import synth
    tab = rfc.load_tables(os.path.join(ROOT, "shoulder_amd", "models", "rfc_bg3.npz"))
    t0 = time.perf_counter()
    with threadpool_limits(limits=1):
        for i in range(n_meshes):
            h = OracleHumerus(synth.apply_similarity(T[i], verts), faces, tab, weights, unet_eval="f64")
            h.landmarks()
    el = time.perf_counter() - t0
    out = {"value": round(n_meshes / el, 4), "unit": "meshes/s", "cores": 1, "kind": "port",
           "sample": f"{n_meshes} of the batch's synthetic humeri, all stages, NumPy/SciPy oracle, BLAS limited to 1 thread, {el:.1f} s"}
    out.update(host_info())
    if pool < 0:
        pool = min(16, usable_cores())
    pool = min(pool, len(T))
    if pool > 1:
        import multiprocessing as mp
        from concurrent.futures import ProcessPoolExecutor
        t0 = time.perf_counter()
        with ProcessPoolExecutor(max_workers=pool, mp_context=mp.get_context("spawn")) as ex:
            per = list(ex.map(_oracle_one, range(pool)))
        elp = time.perf_counter() - t0
        # wall includes interpreter start-up and imports of every worker; `value_compute_only` counts the slowest worker's
        # oracle time alone (all workers run side by side), which is what a long-running pool would sustain
        out["pool"] = {"value": round(pool / elp, 4), "unit": "meshes/s", "cores": pool, "value_compute_only": round(pool / max(per), 4),
                       "per_worker_s": [round(x, 2) for x in per],
                       "host_cores": os.cpu_count(),
                       "sample": f"{pool} humeri, one per worker process (spawn) on {pool} of the host's {os.cpu_count()} hardware threads, wall {elp:.1f} s incl. start-up and imports; slowest worker's oracle time {max(per):.1f} s"}
        # (the same leg as flat keys: a reader that keeps only scalars still sees that the pool ran)
        out.update(pool_value=out["pool"]["value"], pool_cores=pool, pool_value_compute_only=out["pool"]["value_compute_only"], pool_wall_s=round(elp, 1))
    else:
        out["pool_skipped"] = f"pool of {pool} workers requested (--cpu-pool {pool})"
    return out


if __name__ == "__main__":
    main()
