"""N>1 path on CPU: world_size 2 and 8, gloo (sharding, parameter broadcast, landmark-record gather), and what a rank of an 8-GPU
host chooses without a device (hull mode, lanes)."""
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


def test_eight_rank_gloo_packed_gather():
    """BASELINE configs[3] without the node: eight ranks, 64 packed records of 8 680 + 24 x 2 560 = 70 120 bytes each per rank into rank 0
    (36 MB per step) -- sharding of the 512-humerus transform sequence, the parameter broadcast and the gather in rank order, on gloo."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="1", SH_DIST_PER_RANK="64", SH_DIST_ROWS="2560")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=8", "--master-addr", "127.0.0.1",
           "--master-port", "29617", os.path.join(ROOT, "tests", "_dist_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "DIST_OK" in r.stdout


def test_a_rank_of_eight_takes_the_device_hull_and_three_lanes():
    """`auto` hull mode of a rank that shares its host with seven others: 48 usable hardware threads per rank are needed for the host
    hull (sh_auto_hull_mode: affinity mask, cgroup quota, LOCAL_WORLD_SIZE; no device needed), so on any host below 384 threads the
    rank takes the device hull -- and bench.py then runs three lanes.  A process of its own: the library reads LOCAL_WORLD_SIZE."""
    import bench
    code = ("import ctypes, sys; sys.path.insert(0, %r); from shoulder_amd import build as b; L = ctypes.CDLL(b.LIB); "
            "L.sh_auto_hull_mode.restype = ctypes.c_int; print('MODE', L.sh_auto_hull_mode())" % ROOT)
    ncpu = len(os.sched_getaffinity(0))
    for lws, want in (("8", 1 if ncpu < 384 else 0), ("1", 0 if ncpu >= 16 else 1)):
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LOCAL_WORLD_SIZE=lws), capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stdout + r.stderr
        mode = int(r.stdout.split("MODE")[1].split()[0])
        if lws == "8" or not os.path.exists("/sys/fs/cgroup/cpu.max"):      # (a CPU quota can only lower the count)
            assert mode == want, (lws, mode, ncpu)
        else:
            assert mode in (want, 1)
    assert bench.default_lanes("device") == 3 and bench.default_lanes("host") == 2 and bench.default_lanes("host", pipelined=False) == 1
