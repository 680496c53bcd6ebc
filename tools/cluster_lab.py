#!/usr/bin/env python3
"""GPU box: device clustering vs host clustering, timed, one line per step (flushed)."""
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
sys.path.insert(0, str(REPO / "tests"))
sys.path.insert(0, str(REPO / "oracle"))
import bsmr_amd as eng  # noqa: E402
import synth  # noqa: E402
from test_clustering_exact import CASES, clustered_pattern  # noqa: E402


def say(*a):
    print(*a, flush=True)


def one(name, rows, cols, ro, ci, bw, alpha, host=True):
    say(f"-- {name} rows={rows} cols={cols} nnz={ci.size} bw={bw} alpha={alpha}")
    t0 = time.perf_counter()
    st, perm, clusters, stats = eng.cluster_rows_device(rows, cols, ro, ci, bw, alpha)
    say(f"   device: status {st}, {clusters} clusters, wall {1e3 * (time.perf_counter() - t0):.1f} ms, {stats}")
    if host:
        t0 = time.perf_counter()
        pipe = eng.Pipeline(eng.CSR.from_arrays(rows, cols, ro, ci), alpha=alpha, delta=0.3, block_size=bw, device=-1)
        same = np.array_equal(pipe.array("reorderedRows"), perm) and pipe.num_clusters == clusters
        say(f"   host:   {pipe.num_clusters} clusters, wall {1e3 * (time.perf_counter() - t0):.1f} ms, identical={same}")


which = sys.argv[1] if len(sys.argv) > 1 else "small"
if which == "small":
    for case in CASES:
        rows, cols, bw, groups, per_row, seed = case
        r, c, ro, ci = clustered_pattern(rows, cols, groups, per_row, seed)
        for alpha in (0.1, 0.3, 0.6, 0.9):
            one(f"bins{-(-cols // bw)}", r, c, ro, ci, bw, alpha)
else:
    pats = {"tref": lambda: synth.trefethen_pattern(20000), "myc14": lambda: synth.mycielskian_pattern(14),
            "myc15": lambda: synth.mycielskian_pattern(15), "wathen100": lambda: synth.wathen_pattern(100, 100),
            "cop20k": lambda: synth.banded_mesh_like(), "nips": lambda: synth.nips_like()}
    r, c, ro, ci = pats[which]()
    for alpha in [float(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["0.3"])]:
        one(which, r, c, ro, ci, 16 if which != "cop20k" else 20, alpha, host=len(sys.argv) <= 3)
