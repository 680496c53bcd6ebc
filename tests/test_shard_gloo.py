"""Multi-rank path on CPU: world_size 2 and 3 over gloo (the GPU path uses the same
orchestration with backend nccl = RCCL and the HIP launcher as `compute`)."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

import synth

REPO = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_gather_over_gloo(engine, oracle, world):
    env = dict(os.environ, OMP_NUM_THREADS="2", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(29600 + world),
           str(REPO / "tests" / "_shard_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert f"SHARD_OK {world}" in r.stdout


def test_partition_rows_balances_nnz():
    import shard
    rows, cols, ro, ci = synth.random_pattern(500, 64, 9000, seed=1, empty_rows=40)
    for world in (1, 2, 4, 8):
        b = shard.partition_rows(ro, world)
        assert len(b) == world + 1 and b[0] == 0 and b[-1] == rows
        per = [int(ro[b[i + 1]]) - int(ro[b[i]]) for i in range(world)]
        assert sum(per) == ci.size and max(per) - min(per) <= 2 * int(np.diff(ro.astype(np.int64)).max())
    # degenerate: more ranks than rows with entries
    b = shard.partition_rows(np.array([0, 5, 5, 5], np.uint32), 4)
    assert b[0] == 0 and b[-1] == 3 and all(x <= y for x, y in zip(b, b[1:]))
