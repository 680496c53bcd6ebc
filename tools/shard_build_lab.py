#!/usr/bin/env python3
"""GPU box: what ONE rank of an N-rank strong-scaling run of the reddit-like graph does before its first step -
generate its rows, build its plan - timed, for N in argv (default 2 4).  Rank 0's range (the cost partition's first)."""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
import bsmr_amd as eng   # noqa: E402
import shard             # noqa: E402
import synth             # noqa: E402

n = 232965
t0 = time.perf_counter()
deg = synth.reddit_like_degrees(n=n)
costs = shard.row_costs(deg)
print(f"degrees + costs {time.perf_counter() - t0:.1f} s", flush=True)
for world in [int(a) for a in sys.argv[1:]] or [2, 4]:
    b = shard.partition_by_cost(costs, world)
    t0 = time.perf_counter()
    rows, cols, ro, ci = synth.reddit_like_rows(int(b[0]), int(b[1] - b[0]), n=n, degrees=deg)
    t1 = time.perf_counter()
    csr = eng.CSR.from_arrays(rows, cols, ro, ci)
    pipe = eng.Pipeline(csr, alpha=0.3, delta=0.3, device=0)
    t2 = time.perf_counter()
    print(f"N={world}: rank 0 owns {rows} rows, {ci.size} entries: rows generated in {t1 - t0:.1f} s, pipeline + plan {t2 - t1:.1f} s "
          f"(row clustering {pipe.row_reordering_ms / 1e3:.1f} s, {pipe.num_clusters} clusters; plan {pipe.plan_build_ms()})", flush=True)
    del pipe
