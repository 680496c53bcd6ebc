#!/usr/bin/env python3
"""GPU box: device vs host clustering on a larger synthetic graph (equality + time)."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
import hostinfo; hostinfo.limit_openmp_threads()
import numpy as np
import bsmr_amd as eng, synth
n, deg = int(sys.argv[1]), int(sys.argv[2])
rows, cols, ro, ci = synth.community_graph(n=n, avg_degree=deg, communities=32, seed=12)
print("pattern", rows, cols, ci.size, flush=True)
for alpha in (0.3, 0.7):
    t0 = time.perf_counter(); st, perm, clusters, stats = eng.cluster_rows_device(rows, cols, ro, ci, 16, alpha); td = time.perf_counter() - t0
    print(f"alpha {alpha}: device status {st} clusters {clusters} {td * 1e3:.0f} ms {stats}", flush=True)
    t0 = time.perf_counter(); pipe = eng.Pipeline(eng.CSR.from_arrays(rows, cols, ro, ci), alpha=alpha, delta=0.3, block_size=16, device=-1); th = time.perf_counter() - t0
    print(f"           host pipeline {th * 1e3:.0f} ms clusters {pipe.num_clusters} identical={np.array_equal(pipe.array('reorderedRows'), perm)}", flush=True)
