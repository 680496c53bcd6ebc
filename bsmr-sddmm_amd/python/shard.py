"""Row-range sharding of one SDDMM over the GPUs of a node.

The reference is single-GPU; this layer is new.  Row panels are independent (a
panel owns a disjoint set of output entries), so:

  * the rows of S are cut into `world` contiguous ranges of (nearly) equal COST
    (entries plus a fixed charge per non-empty row, cuts at multiples of 16 rows:
    partition_rows_by_cost / bsmr_partition_rows_by_cost);
    rank r owns range r: its slice of S (a CSR with local row ids), the matching
    rows of A, and a full copy of B.  Each rank runs the whole BSMR pipeline
    (cluster, reorder, split, plan) on its own slice - clustering never crosses a
    shard boundary - and computes its entries of P in the slice's CSR order.
  * Because the ranges are contiguous in the original row order, the global P
    (S's CSR order) is simply the concatenation of the shards' outputs.  The only
    data-path collective is therefore ONE gather-v to rank 0: every peer sends its
    compact fp32 vector straight into its slot of the root's P (grouped
    send/recv = ncclGroupStart/End on RCCL; 7 peers arrive over 7 separate xGMI
    links).  No permutation pass is needed on the root.

`compute` is injected so that the orchestration can be exercised on CPU (gloo)
with the oracle standing in for the device; the product path passes the HIP
launcher.
"""
from __future__ import annotations

import numpy as np


def partition_rows(row_offsets: np.ndarray, world: int):
    """Contiguous row ranges with nearly equal nnz: returns world+1 row boundaries."""
    ro = row_offsets.astype(np.int64)
    rows = ro.size - 1
    nnz = int(ro[-1])
    bounds = [0]
    for r in range(1, world):
        target = nnz * r // world
        cut = int(np.searchsorted(ro, target, side="left"))
        cut = min(max(cut, bounds[-1]), rows)
        bounds.append(cut)
    bounds.append(rows)
    return bounds


def row_costs(degrees: np.ndarray, c_entry: float = 1.0, c_panel: float = 24.0) -> np.ndarray:
    """Cost model of SURVEY.md 8(e), per row and before a plan exists: a row's entries (each one a share of a dense
    block's B gather and MFMA work, or one residue entry) plus its sixteenth of a 16-row panel's fixed work (A
    fragments, block records) - so that ranges of many short rows do not look free."""
    d = np.asarray(degrees, dtype=np.float64)
    return c_entry * d + (c_panel / 16.0) * (d > 0)


def partition_by_cost(costs: np.ndarray, world: int):
    """Contiguous row ranges of nearly equal total cost: world+1 row boundaries, cut at multiples of 16 rows (a row
    panel never straddles two ranks' plans for identity row order) except at the end."""
    c = np.concatenate([[0.0], np.cumsum(np.asarray(costs, dtype=np.float64))])
    rows = c.size - 1
    bounds = [0]
    for r in range(1, world):
        cut = int(np.searchsorted(c, c[-1] * r / world, side="left"))
        cut = min(max((cut + 8) // 16 * 16, bounds[-1]), rows)
        bounds.append(cut)
    bounds.append(rows)
    return bounds


def local_slice(rows, cols, ro, ci, r0, r1):
    """CSR of rows [r0, r1) with local row ids; also the slice's offset in the global P."""
    ro64 = ro.astype(np.int64)
    e0, e1 = int(ro64[r0]), int(ro64[r1])
    lro = (ro64[r0:r1 + 1] - e0).astype(np.uint32)
    return r1 - r0, cols, lro, ci[e0:e1].copy(), e0, e1 - e0


def start_gather(dist, rank, world, local_out, root_out, offsets, counts):
    """Issues one gather-v and returns its requests: peers send their compact outputs into
    root_out[offsets[r]:+counts[r]].  On the root the local result already lives in its own slot."""
    if world == 1:
        return []
    ops = []
    if rank == 0:
        for r in range(1, world):
            if counts[r]:
                ops.append(dist.P2POp(dist.irecv, root_out[offsets[r]:offsets[r] + counts[r]], r))
    elif counts[rank]:
        ops.append(dist.P2POp(dist.isend, local_out, 0))
    return list(dist.batch_isend_irecv(ops)) if ops else []


def gather_to_root(dist, rank, world, local_out, root_out, offsets, counts):
    """One gather-v, completed before returning (stream-ordered on GPU backends)."""
    for req in start_gather(dist, rank, world, local_out, root_out, offsets, counts):
        req.wait()


class PipelinedSteps:
    """SDDMM + gather, step after step, with the gather of step i in flight while step i+1 computes.

    Two output buffers alternate (on the root: two full P vectors, the local result being a view into the
    current one), so a step's gathered P stays intact until the step after next starts.  Every step still
    ends with its own gather; only the wait for it is issued one step later.  `compute(buf)` fills the local
    output `buf`."""

    def __init__(self, dist, rank, world, offsets, counts, local_outs, root_outs):
        self.dist, self.rank, self.world = dist, rank, world
        self.offsets, self.counts = offsets, counts
        self.local_outs, self.root_outs = local_outs, root_outs
        self.pending = []
        self.index = 0

    def step(self, compute):
        b = self.index & 1
        self.index += 1
        compute(self.local_outs[b])
        reqs = start_gather(self.dist, self.rank, self.world, self.local_outs[b], self.root_outs[b], self.offsets,
                            self.counts)
        for req in self.pending:      # the previous step's gather, overlapped with this step's compute
            req.wait()
        self.pending = reqs
        return b

    def drain(self):
        for req in self.pending:
            req.wait()
        self.pending = []


def sharded_sddmm(dist, rank, world, ro, counts_offsets, compute, local_out, root_out):
    """compute() fills local_out (the root's local_out is a view into root_out); then gather."""
    offsets, counts = counts_offsets
    compute()
    gather_to_root(dist, rank, world, local_out, root_out, offsets, counts)


def run_sharded(eng, torch, dist, dev, rank, world, make_pattern, K, alpha, delta, mode, steps, warmup,
                scaling="weak", strong_rows=None):
    """bench.py's N > 1 path.  weak scaling: the job is `world` row-stacked patterns
    (rank r generates and owns copy r); strong: one pattern cut into `world` ranges - with `strong_rows` =
    (row costs of the whole pattern, make_rows(first_row, rows)) every rank builds ONLY its own rows, the
    boundaries come from the cost partition."""
    import time

    if scaling == "weak":
        rows, cols, lro, lci = make_pattern(rank)
        counts = [0] * world
        t = torch.tensor([lci.size], dtype=torch.int64, device=dev)
        allc = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(allc, t)
        counts = [int(c.item()) for c in allc]
        lrows, lnnz = rows, int(lci.size)
        total_rows = rows * world
    elif strong_rows is not None:
        costs, make_rows = strong_rows
        b = partition_by_cost(costs, world)
        lrows, cols, lro, lci = make_rows(b[rank], b[rank + 1] - b[rank])
        lnnz = int(lci.size)
        t = torch.tensor([lnnz], dtype=torch.int64, device=dev)
        allc = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
        dist.all_gather(allc, t)
        counts = [int(c.item()) for c in allc]
        total_rows = int(len(costs))
    else:
        rows, cols, ro, ci = make_pattern(0)
        b = partition_rows(ro, world)
        parts = [local_slice(rows, cols, ro, ci, b[r], b[r + 1]) for r in range(world)]
        lrows, _, lro, lci, _, lnnz = parts[rank]
        counts = [p[5] for p in parts]
        total_rows = rows
    offsets = [0]
    for c in counts[:-1]:
        offsets.append(offsets[-1] + c)
    total_nnz = sum(counts)

    t0 = time.perf_counter()
    csr = eng.CSR.from_arrays(lrows, cols, lro, lci)
    pipe = eng.Pipeline(csr, alpha=alpha, delta=delta, device=dev.index)
    plan_s = time.perf_counter() - t0
    A = torch.from_numpy(eng.make_data(lrows * K, 5489 + 17 * rank)).to(dev)
    B = torch.from_numpy(eng.make_data(cols * K, 5490)).to(dev)   # replicated
    root_outs = [torch.zeros(total_nnz if rank == 0 else 1, dtype=torch.float32, device=dev) for _ in range(2)]
    local_outs = [root_outs[b][:lnnz] if rank == 0
                  else torch.zeros(max(lnnz, 1), dtype=torch.float32, device=dev)[:lnnz] for b in range(2)]
    local_out = local_outs[0]
    sh = torch.cuda.current_stream(dev).cuda_stream
    eng.hip().bsmr_plan_reserve(pipe.plan, K)

    def compute(buf):
        eng.sddmm(pipe.plan, K, A.data_ptr(), B.data_ptr(), buf.data_ptr(), mode, sh)

    # the gather of step i overlaps the compute of step i+1 (BSMR_SHARD_PIPELINE=0: strictly one after the other)
    import os
    pipelined = os.environ.get("BSMR_SHARD_PIPELINE", "1") != "0"
    runner = PipelinedSteps(dist, rank, world, offsets, counts, local_outs, root_outs)

    def step():
        runner.step(compute)
        if not pipelined:
            runner.drain()

    for _ in range(warmup):
        step()
    runner.drain()
    torch.cuda.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    runner.drain()
    torch.cuda.synchronize()
    dist.barrier()
    dt = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
    dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    ms = float(dt.item()) / steps * 1e3

    # compute-only time (no gather), max over ranks
    kt = eng.sddmm_timed(pipe.plan, K, A.data_ptr(), B.data_ptr(), local_out.data_ptr(), mode, sh, 3, max(steps // 4, 5))
    ct = torch.tensor([kt["total_ms"]], dtype=torch.float64, device=dev)
    per_rank = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
    dist.all_gather(per_rank, ct.clone())
    per_rank_ms = [float(x.item()) for x in per_rank]
    dist.all_reduce(ct, op=dist.ReduceOp.MAX)
    # the gather alone: a few gather-v rounds with nothing to compute
    for _ in range(2):
        gather_to_root(dist, rank, world, local_outs[0], root_outs[0], offsets, counts)
    torch.cuda.synchronize()
    dist.barrier()
    g0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        gather_to_root(dist, rank, world, local_outs[0], root_outs[0], offsets, counts)
    torch.cuda.synchronize()
    dist.barrier()
    gt = torch.tensor([(time.perf_counter() - g0) / reps * 1e3], dtype=torch.float64, device=dev)
    dist.all_reduce(gt, op=dist.ReduceOp.MAX)
    return {
        "metric": "SDDMM GFLOP/s", "value": round(2.0 * total_nnz * K / (ms * 1e6), 2), "unit": "GFLOP/s",
        "ms_per_step": round(ms, 5), "higher_is_better": True, "scaling": scaling, "vs_baseline": None,
        "data": "synthetic",
        "config": {"workload": f"{world} row shards of {lrows} rows x {cols} cols, total {total_rows} rows, "
                               f"nnz {total_nnz}, K={K}, alpha={alpha}, delta={delta}",
                   "parallelism": f"row-range shards x{world}, B replicated, one RCCL gather-v to rank 0 per step"
                                  + (" (the gather of step i overlaps the compute of step i+1)" if pipelined else "")},
        "compute_only_ms_max": round(float(ct.item()), 5),
        "compute_only_ms_per_rank": [round(x, 5) for x in per_rank_ms],
        "compute_imbalance": round(max(per_rank_ms) / (sum(per_rank_ms) / len(per_rank_ms)), 4) if sum(per_rank_ms) > 0 else None,
        "gather_only_ms": round(float(gt.item()), 5),
        "nnz_per_rank": counts,
        "plan_build_s": round(plan_s, 3),
        # this rank's shard, for the caller's roofline (used on rank 0)
        "rank0": {"kernels_ms": kt, "pattern": (lrows, cols, lro, lci), "dense_tiles": pipe.dense_choice(K)["tiles"],
                  "union_columns": pipe.dense_choice(K)["union_columns"], "sparse_nnz": pipe.plan_stats()["num_sparse_entries"],
                  "sparse_lowp": bool(pipe.sparse_choice(K, mode)["low_precision"])},
    }
