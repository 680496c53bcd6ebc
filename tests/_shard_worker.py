"""Worker for tests/test_shard_gloo.py: 2+ CPU ranks over gloo run the sharded
orchestration of bsmr-sddmm_amd/python/shard.py with the oracle standing in for the
device kernels, and the root checks the gathered P against a single-process oracle."""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
sys.path.insert(0, str(REPO / "tests"))
import bsmr_amd as eng  # noqa: E402
import shard  # noqa: E402
import synth  # noqa: E402
from conftest import Oracle  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    oracle = Oracle()
    K = 64
    rows, cols, ro, ci = synth.random_pattern(97, 120, 2500, seed=5, empty_rows=6)
    A = eng.make_data(rows * K, 21)
    B = eng.make_data(cols * K, 22)

    # ---- strong: one pattern cut into contiguous row ranges of equal nnz ----
    bounds = shard.partition_rows(ro, world)
    assert bounds[0] == 0 and bounds[-1] == rows and all(a <= b for a, b in zip(bounds, bounds[1:]))
    parts = [shard.local_slice(rows, cols, ro, ci, bounds[r], bounds[r + 1]) for r in range(world)]
    counts = [p[5] for p in parts]
    offsets = [p[4] for p in parts]
    assert sum(counts) == ci.size and offsets == list(np.cumsum([0] + counts[:-1]))
    lrows, _, lro, lci, e0, lnnz = parts[rank]
    # every rank runs the whole host pipeline on its slice (device=-1: no GPU here)
    csr = eng.CSR.from_arrays(lrows, cols, lro, lci)
    pipe = eng.Pipeline(csr, alpha=0.3, delta=0.2, device=-1)
    assert pipe.check()
    root_out = torch.zeros(ci.size if rank == 0 else 1, dtype=torch.float32)
    local_out = root_out[:lnnz] if rank == 0 else torch.zeros(lnnz, dtype=torch.float32)
    A_local = A.reshape(rows, K)[bounds[rank]:bounds[rank + 1]].ravel().copy()

    def compute():  # the oracle plays the device: local slice, local rows of A, all of B
        local_out.copy_(torch.from_numpy(oracle.sddmm_cpu(lrows, cols, K, lro, lci, A_local, B)))

    shard.sharded_sddmm(dist, rank, world, ro, (offsets, counts), compute, local_out, root_out)
    if rank == 0:
        want = oracle.sddmm_cpu(rows, cols, K, ro, ci, A, B)
        assert np.array_equal(root_out.numpy().view(np.uint32), want.view(np.uint32)), "strong: gathered P differs"

    # ---- weak: rank r owns row-stacked copy r (its own pattern and A rows) ----
    prow, pcol, pro, pci = synth.random_pattern(40, cols, 600 + 50 * rank, seed=100 + rank)
    n = torch.tensor([pci.size], dtype=torch.int64)
    allc = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allc, n)
    counts = [int(c.item()) for c in allc]
    offsets = [int(x) for x in np.cumsum([0] + counts[:-1])]
    Ar = eng.make_data(prow * K, 300 + rank)
    root_out = torch.zeros(sum(counts) if rank == 0 else 1, dtype=torch.float32)
    local_out = root_out[:counts[0]] if rank == 0 else torch.zeros(counts[rank], dtype=torch.float32)

    def compute_weak():
        local_out.copy_(torch.from_numpy(oracle.sddmm_cpu(prow, pcol, K, pro, pci, Ar, B)))

    shard.sharded_sddmm(dist, rank, world, None, (offsets, counts), compute_weak, local_out, root_out)
    if rank == 0:
        got = root_out.numpy()
        for r in range(world):
            rr, rc, rro, rci = synth.random_pattern(40, cols, 600 + 50 * r, seed=100 + r)
            want = oracle.sddmm_cpu(rr, rc, K, rro, rci, eng.make_data(rr * K, 300 + r), B)
            seg = got[offsets[r]:offsets[r] + counts[r]]
            assert np.array_equal(seg.view(np.uint32), want.view(np.uint32)), f"weak: shard {r} differs"
    # ---- pipelined steps: the gather of step i is waited for after step i+1 has computed; each step's
    #      gathered P (step-dependent operands) must be intact when its buffer comes up again ----
    root_outs = [torch.zeros(sum(counts) if rank == 0 else 1, dtype=torch.float32) for _ in range(2)]
    local_outs = [root_outs[b][:counts[0]] if rank == 0 else torch.zeros(counts[rank], dtype=torch.float32)
                  for b in range(2)]
    runner = shard.PipelinedSteps(dist, rank, world, offsets, counts, local_outs, root_outs)
    base = oracle.sddmm_cpu(prow, pcol, K, pro, pci, Ar, B)
    seen = {}
    for stepno in range(5):
        scale = np.float32(stepno + 1)

        def compute_step(buf, scale=scale):
            buf.copy_(torch.from_numpy(base * scale))

        b = runner.step(compute_step)
        # the buffer of the previous step is complete now (its gather was waited for inside step())
        if stepno >= 1 and rank == 0:
            seen[stepno - 1] = root_outs[1 - b].numpy().copy()
    runner.drain()
    if rank == 0:
        seen[4] = root_outs[0].numpy().copy()          # step 4 used buffer 0
        for stepno, got in seen.items():
            for r in range(world):
                rr, rc, rro, rci = synth.random_pattern(40, cols, 600 + 50 * r, seed=100 + r)
                want = oracle.sddmm_cpu(rr, rc, K, rro, rci, eng.make_data(rr * K, 300 + r), B) * np.float32(stepno + 1)
                seg = got[offsets[r]:offsets[r] + counts[r]]
                assert np.array_equal(seg.view(np.uint32), want.view(np.uint32)), f"pipelined step {stepno}, shard {r}"
    # ---- strong scaling by cost (bench.py --gpus N default): every rank builds ONLY its own rows of one graph,
    #      the row ranges come from the cost partition of the degree sequence, which every rank can compute ----
    n, Kc = 2600, 32
    deg = synth.reddit_like_degrees(n=n, avg_degree=24)
    bounds = shard.partition_by_cost(shard.row_costs(deg), world)
    assert bounds[0] == 0 and bounds[-1] == n and all(a <= b for a, b in zip(bounds, bounds[1:]))
    assert all(b % 16 == 0 for b in bounds[1:-1])
    lrows, lcols, lro, lci = synth.reddit_like_rows(bounds[rank], bounds[rank + 1] - bounds[rank], n=n, avg_degree=24,
                                                    communities=5, degrees=deg)
    csr = eng.CSR.from_arrays(lrows, lcols, lro, lci)
    pipe = eng.Pipeline(csr, alpha=0.3, delta=0.3, device=-1)       # the whole host pipeline on the rank's own rows
    assert pipe.check()
    mine = torch.tensor([lci.size], dtype=torch.int64)
    allc = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(allc, mine)
    counts = [int(c.item()) for c in allc]
    offsets = [int(x) for x in np.cumsum([0] + counts[:-1])]
    Bc = eng.make_data(n * Kc, 41)
    Ac = eng.make_data(n * Kc, 40).reshape(n, Kc)[bounds[rank]:bounds[rank + 1]].ravel().copy()
    root_out = torch.zeros(sum(counts) if rank == 0 else 1, dtype=torch.float32)
    local_out = root_out[:counts[0]] if rank == 0 else torch.zeros(counts[rank], dtype=torch.float32)

    def compute_cost():
        local_out.copy_(torch.from_numpy(oracle.sddmm_cpu(lrows, lcols, Kc, lro, lci, Ac, Bc)))

    shard.sharded_sddmm(dist, rank, world, None, (offsets, counts), compute_cost, local_out, root_out)
    if rank == 0:
        fr, fc, fro, fci = synth.reddit_like_rows(0, n, n=n, avg_degree=24, communities=5, degrees=deg)
        assert fci.size == sum(counts)
        want = oracle.sddmm_cpu(fr, fc, Kc, fro, fci, eng.make_data(n * Kc, 40), Bc)
        assert np.array_equal(root_out.numpy().view(np.uint32), want.view(np.uint32)), "strong by cost: gathered P differs"
        per = [int(deg[bounds[r]:bounds[r + 1]].sum()) for r in range(world)]
        assert max(per) - min(per) <= 0.1 * max(per), per
        print("SHARD_OK", world)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
