"""N>1 path on CPU: world_size 2, gloo (sharding, parameter broadcast, landmark-record gather)."""
import os
import subprocess
import sys

from conftest import ROOT
from shoulder_amd import dist as shd


def test_shard_bounds():
    assert [shd.shard_bounds(512, 8, r) for r in range(8)] == [(64 * r, 64) for r in range(8)]
    b = [shd.shard_bounds(10, 4, r) for r in range(4)]
    assert b == [(0, 3), (3, 3), (6, 2), (8, 2)] and sum(c for _, c in b) == 10


def test_two_rank_gloo():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", "29611", os.path.join(ROOT, "tests", "_dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "DIST_OK" in r.stdout
