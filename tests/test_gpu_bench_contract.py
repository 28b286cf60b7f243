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
