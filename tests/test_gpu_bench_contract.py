"""bench.py's one-line JSON contract (a short run in a child process): the keys and value types the driver reads, the
`roofline` and `cpu_baseline` objects, `vs_baseline` null, weak scaling, no mesh with an error status."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def test_bench_line():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--batch", "8",
                          "--cpu-meshes", "1", "--cpu-pool", "0"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                    # ONE JSON line
    d = json.loads(lines[0])
    assert d["metric"].startswith("humerus meshes/s") and d["unit"] == "meshes/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 4 and d["warmup"] == 1 and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["value"] > 0 and d["ms_per_step"] > 0 and abs(d["value"] - 8 * 1e3 / d["ms_per_step"]) < 1e-2 * d["value"]
    assert "bf16" in d["dtype"] and d["data"].startswith("synthetic")
    assert "workload" in d["config"] and "model" not in d["config"] and d["config"]["meshes_with_error_status"] == 0
    r = d["roofline"]
    assert r["bound"] in ("hbm", "mfma") and r["unit"] in ("GB/s", "TFLOP/s") and r["peak"] > 0
    assert abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-3 and "traffic" in r and r["kernel"]
    c = d["cpu_baseline"]
    assert c["kind"] in ("port", "reference") and c["cores"] == 1 and c["value"] > 0 and c["unit"] == "meshes/s" and c["sample"]
    assert c["cpu_model"] and c["os_cpu_count"] >= 1
    # the tolerance-conformant configuration (f32 UNet) is timed in the same invocation, and the one-lane figure beside the two-lane value
    f = d["f32_unet"]
    assert f["value"] > 0 and "f32" in f["dtype"] and abs(f["value"] - 8 * 1e3 / f["ms_per_step"]) < 1e-2 * f["value"]
    assert f["roofline"]["peak"] == 157.3 and 0 < f["roofline"]["frac"] < 1
    assert d["one_lane"]["lanes"] == 1 and d["one_lane"]["value"] > 0
    # the step in the OTHER hull mode than the one `auto` picked on this host: same records either way
    assert d["config"]["hull"] in ("host", "device")
    if d["config"]["hull"] == "host":
        dh = d["device_hull"]      # hull on the device: no host work
        assert dh["value"] > 0 and dh["host_ms_per_step"] == {"host.verts_d2h": 0.0, "host.hull": 0.0} and dh["records_equal_to_host_hull_run"] is True
    else:
        hh = d["host_hull"]
        assert hh["value"] > 0 and hh["lanes"] == 2 and hh["host_ms_per_step"]["host.hull"] > 0 and hh["records_equal_to_device_hull_run"] is True
        assert d["host_ms_per_step"] == {"host.verts_d2h": 0.0, "host.hull": 0.0} and d["config"]["lanes"] == 3
    assert d["config"]["hull_threads_per_process"] >= 1 and d["config"]["host"]["os_cpu_count"] >= 1
    # geometry table: per STEP (every launch of a kernel name in one step), algorithmic and SURVEY bytes, PMC bytes or null
    gk = d["geometry_kernels"]
    row = gk["k_slice_link"]
    assert row["launches_per_step"] == 2 and row["ms_per_step"] > 0      # (full + distal, neck contour + proximal: two join grids per step since round 4)
    assert row["ms_per_step"] > 0 and row["algorithmic_mb_per_step"] > 0 and "pmc_mb_per_step" in row
    assert abs(row["frac_of_hbm_peak"] - row["algorithmic_mb_per_step"] / row["ms_per_step"] / 8000.0) < 2e-3
    assert "k_slice_link_large" not in gk or gk["k_slice_link_large"]["algorithmic_mb_per_step"] == 0      # (own timer name: not averaged into the row above)
    assert gk["k_resample_polar"]["launches_per_step"] == 1 and gk["k_resample_polar"]["survey_mb_per_step"] > 0
    assert d["geometry_ms_per_step_one_lane"] > 0
    assert "1e-4 mm" in d["parity"] and "bf16" in d["parity"]
    # BASELINE configs[1]: one humerus, f32 UNet: latency through the engine and through the facade's README flow
    sh = d["single_humerus_f32"]
    assert sh["status_ok"] is True and 0 < sh["engine_run_ms_min"] <= sh["engine_run_ms"] and sh["facade_readme_flow_ms"] > 0
    assert {"obb", "slices", "bicipital_groove", "anatomic_neck.unet", "trans_epicondylar", "csys"} <= set(sh["device_ms_by_stage"])
    assert sh["device_ms_total"] <= sh["engine_run_ms"] * 1.05
    # the headline configuration over a timed region ten times as long (full lanes: fill and drain weigh a tenth)
    ss = d["steady_state"]
    assert ss["steps"] == 10 * d["steps"] and ss["lanes"] == d["config"]["lanes"] and ss["value"] > 0.9 * d["value"]


def test_rccl_leg_single_rank():
    """The collective code path of bench.py on hardware (SH_BENCH_FORCE_DIST=1: process group "nccl" = RCCL with one rank): RCCL
    broadcast of the device parameter block wrapped through __cuda_array_interface__, sh_param_block_commit, every step's
    records written by sh_submit into a device send buffer and gathered asynchronously; rank 0's gathered records equal a
    run of its own engine."""
    env = dict(os.environ, SH_BENCH_FORCE_DIST="1", MASTER_ADDR="127.0.0.1", MASTER_PORT="29577")
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--batch", "8",
                          "--no-cpu-baseline", "--check-gather"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["config"]["meshes_with_error_status"] == 0
    assert d["gather_check"] == {"records": 8, "own_shard_equal_to_local_run": True}
    assert "f32_unet" not in d            # extra legs belong to the plain N=1 run


def test_rccl_leg_as_one_of_eight_ranks():
    """The same leg configured as a rank of an 8-GPU node sees itself (LOCAL_WORLD_SIZE=8: fewer than 48 hardware threads
    per rank even on a 256-thread host): hull mode `auto` resolves to the device hull, bench.py then runs three lanes, nothing
    goes through the host (`host_ms_per_step` = 0), stdout carries exactly one line, and the gathered records equal the
    rank's own run."""
    env = dict(os.environ, SH_BENCH_FORCE_DIST="1", LOCAL_WORLD_SIZE="8", MASTER_ADDR="127.0.0.1", MASTER_PORT="29578")
    env.pop("SHOULDER_HULL", None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "6", "--warmup", "1", "--batch", "8",
                          "--no-cpu-baseline", "--check-gather"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    assert len([ln for ln in out.stdout.splitlines() if ln.strip()]) == 1, out.stdout[:500]
    d = json.loads(out.stdout)
    import bench
    if bench.usable_cores() // 8 < 48:
        assert d["config"]["hull"] == "device" and d["config"]["lanes"] == 3
        assert d["host_ms_per_step"] == {"host.verts_d2h": 0.0, "host.hull": 0.0}
    assert d["value"] > 0 and d["config"]["meshes_with_error_status"] == 0
    assert d["gather_check"] == {"records": 8, "own_shard_equal_to_local_run": True}
