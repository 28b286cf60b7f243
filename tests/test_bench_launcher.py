"""bench.py --gpus N started without torch.distributed.run launches its own N ranks (bench.launch_ranks): environment
composition, exactly one line on stdout (rank 0's), failure propagation.  Runs on the CPU with a stub worker."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STUB = os.path.join(ROOT, "tests", "_launch_stub.py")
DRIVER = ("import sys; sys.path.insert(0, %r); import bench; "
          "sys.exit(bench.launch_ranks(int(sys.argv[1]), sys.argv[2:], worker=[sys.executable, %r], timeout=60))" % (ROOT, STUB))


def run(n, extra_env=None, args=("--steps", "3")):
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_PORT"):
        env.pop(k, None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", DRIVER, str(n), *args], env=env, capture_output=True, text=True, timeout=120)


def test_environment_and_single_stdout_line():
    r = run(4)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1                                  # rank 0's line only; ranks 1..3 printed to stderr
    d = json.loads(lines[0])
    assert d["env"]["RANK"] == "0" and d["env"]["LOCAL_RANK"] == "0" and d["env"]["WORLD_SIZE"] == "4" and d["env"]["LOCAL_WORLD_SIZE"] == "4"
    assert d["env"]["MASTER_ADDR"] == "127.0.0.1" and 1024 < int(d["env"]["MASTER_PORT"]) < 65536
    assert d["argv"] == ["--steps", "3"]
    others = [json.loads(ln) for ln in r.stderr.splitlines() if ln.startswith("{")]
    assert sorted(o["env"]["RANK"] for o in others) == ["1", "2", "3"]
    assert len({o["env"]["MASTER_PORT"] for o in others} | {d["env"]["MASTER_PORT"]}) == 1


def test_failure_of_one_rank_stops_the_others_and_is_reported():
    t0 = time.time()
    r = run(3, {"SH_STUB_FAIL_RANK": "1"})
    assert r.returncode == 7
    assert time.time() - t0 < 25                            # the healthy ranks (sleeping 30 s) were terminated
    assert r.stdout.strip() == "" and "rank 1 ended with exit code 7" in r.stderr


def test_bench_main_becomes_the_launcher_only_without_world_size():
    """`python bench.py --gpus 2` with WORLD_SIZE unset must start ranks (we see it try: the children fail here for want of a
    GPU, and that failure comes back as a non-zero exit code, not as a silent one-GPU measurement)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "stopping the other ranks" in r.stderr and r.stdout.strip() == ""
