"""Worker for tests/test_dist_gloo.py (world_size 2, gloo, CPU): shard bounds, parameter broadcast and
record gather exactly as bench.py does them on RCCL."""
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from shoulder_amd._lib import LANDMARKS_DTYPE, record_dtype  # noqa: E402
from shoulder_amd import dist as shd  # noqa: E402
from shoulder_amd import synth  # noqa: E402

dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
per_rank = int(os.environ.get("SH_DIST_PER_RANK", "3"))          # (the 8-rank test: 64 records of 70 KB per rank, BASELINE configs[3])
ROWS = int(os.environ.get("SH_DIST_ROWS", "1536"))
start, count = shd.shard_bounds(per_rank * world, world, rank)
assert (start, count) == (rank * per_rank, per_rank)
# every rank derives its own shard of the seeded transform sequence; shards tile the global sequence
tmpl = np.random.default_rng(0).uniform(-50, 50, (100, 3))
T_all = synth.similarity_transforms(per_rank * world, tmpl, seed=1234)
T_mine = synth.similarity_transforms(count, tmpl, seed=1234, start=start)
np.testing.assert_array_equal(T_mine, T_all[start:start + count])
# parameter block: only rank 0 has the values
params = (np.arange(1000, dtype=np.float32) * 0.5) if rank == 0 else np.zeros(1000, dtype=np.float32)
shd.broadcast_params(params, src=0)
np.testing.assert_array_equal(params, np.arange(1000, dtype=np.float32) * 0.5)
# landmark records
rec = np.zeros(count, dtype=LANDMARKS_DTYPE)
for i in range(count):
    rec["neck_index"][i] = start + i
    rec["canal_axis"][i] = T_mine[i][:2, :3]
    rec["n_anp"][i] = 1000 + start + i
out = shd.gather_records(rec, LANDMARKS_DTYPE, dst=0) if per_rank <= 8 else None      # (full 104 KB records: the small case only)
# the packed wire format (sh_set_record_rows: 8 680 + 24 R bytes per record instead of 104 KB) gathers the same way
PK = record_dtype(ROWS)
recp = np.zeros(count, dtype=PK)
for i in range(count):
    recp["neck_index"][i] = start + i
    recp["n_anp"][i] = 1000 + start + i
    recp["anp_points"][i, :5] = start + i + np.arange(15).reshape(5, 3)
outp = shd.gather_records(recp, PK, dst=0)
if rank == 0:
    if out is not None:
        assert len(out) == per_rank * world
        np.testing.assert_array_equal(out["neck_index"], np.arange(per_rank * world))
        np.testing.assert_array_equal(out["canal_axis"], T_all[:, :2, :3])
        np.testing.assert_array_equal(out["n_anp"], 1000 + np.arange(per_rank * world))
    assert len(outp) == per_rank * world and outp.dtype.itemsize == 8680 + 24 * ROWS
    np.testing.assert_array_equal(outp["neck_index"], np.arange(per_rank * world))
    np.testing.assert_array_equal(outp["n_anp"], 1000 + np.arange(per_rank * world))
    np.testing.assert_array_equal(outp["anp_points"][:, 4, 2], np.arange(per_rank * world) + 14)
    print("DIST_OK")
else:
    assert out is None and outp is None
dist.destroy_process_group()
