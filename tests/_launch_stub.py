"""Stub rank for tests/test_bench_launcher.py: prints its launcher-given environment (rank 0: one JSON line on stdout;
other ranks print too -- the launcher must route that to stderr) and exits with SH_STUB_FAIL_RANK's code on that rank."""
import json
import os
import sys
import time

keys = ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")
rank = int(os.environ["RANK"])
fail = os.environ.get("SH_STUB_FAIL_RANK")
if fail is not None and int(fail) == rank:
    sys.exit(7)
if fail is not None:
    time.sleep(30)          # a healthy rank that would wait in a collective for the failed one: the launcher must stop it
print(json.dumps({"env": {k: os.environ.get(k) for k in keys}, "argv": sys.argv[1:]}))
