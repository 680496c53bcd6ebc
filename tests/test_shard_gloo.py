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


def test_partition_by_cost_balances_the_cost_model():
    """shard.partition_by_cost: contiguous ranges of equal cost = entries + a panel share per non-empty
    row, cut at row-panel boundaries; empty-row stretches do not attract a rank."""
    import shard
    rng = np.random.default_rng(2)
    deg = np.concatenate([rng.integers(200, 400, size=4000), np.zeros(6000, np.int64), rng.integers(1, 4, size=22000)])
    costs = shard.row_costs(deg)
    for world in (2, 4, 8):
        b = shard.partition_by_cost(costs, world)
        assert b[0] == 0 and b[-1] == deg.size and all(x <= y for x, y in zip(b, b[1:]))
        assert all(x % 16 == 0 for x in b[1:-1])
        per = [float(costs[b[i]:b[i + 1]].sum()) for i in range(world)]
        assert max(per) <= 1.02 * (sum(per) / world) + 16 * float(costs.max())
    # by nnz alone the 22 000 short rows (3 entries each, one panel share each) would all land on the last rank
    b_nnz = shard.partition_rows(np.concatenate([[0], np.cumsum(deg)]).astype(np.uint32), 8)
    b_cost = shard.partition_by_cost(costs, 8)
    assert (b_cost[-1] - b_cost[-2]) < (b_nnz[-1] - b_nnz[-2])
    # the graph generator gives the same rows whatever the cut
    a = synth.reddit_like_rows(500, 700, n=3000, avg_degree=10, communities=4)
    full = synth.reddit_like_rows(0, 3000, n=3000, avg_degree=10, communities=4)
    assert np.array_equal(a[3], full[3][full[2][500]:full[2][1200]])


def test_cpp_cost_partition_matches_the_python_one(engine):
    """partitionRowsByCost (the C++ operator sddmm_multi_gpu) and shard.partition_by_cost (bench.py) cut the same way."""
    import shard
    rows, cols, ro, ci = synth.reddit_like_rows(0, 3000, n=3000, avg_degree=30, communities=6)
    csr = engine.CSR.from_arrays(rows, cols, ro, ci)
    deg = np.diff(ro.astype(np.int64))
    for world in (1, 2, 3, 8):
        assert engine.partition_rows_by_cost(csr, world) == shard.partition_by_cost(shard.row_costs(deg), world)
