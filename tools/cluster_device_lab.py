#!/usr/bin/env python3
"""GPU box: bsmr_cluster_rows alone on the bench workloads' patterns - time, passes, judged pairs, dropped tentative
clusters - and, with --check, the row order against the host implementation.
usage: cluster_device_lab.py [--check] [--only NAME] [alpha ...]"""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
import numpy as np       # noqa: E402

import bsmr_amd as eng   # noqa: E402
import synth             # noqa: E402

PATTERNS = [("nips-like", lambda: synth.nips_like()), ("mycielskian14", lambda: synth.mycielskian_pattern(k=14)),
            ("mycielskian15", lambda: synth.mycielskian_pattern(k=15)), ("trefethen20000", lambda: synth.trefethen_pattern(n=20000)),
            ("wathen100", lambda: synth.wathen_pattern(nx=100, ny=100)), ("cop20k-like", lambda: synth.banded_mesh_like()),
            ("bernoulli4096", lambda: synth.bernoulli()), ("reddit-like shard", lambda: synth.reddit_shard_like())]


def main():
    args = sys.argv[1:]
    check = "--check" in args
    only = args[args.index("--only") + 1] if "--only" in args else None
    alphas = [float(a) for a in args if a[0].isdigit()] or [0.3]
    for name, make in PATTERNS:
        if only and only not in name:
            continue
        rows, cols, ro, ci = make()
        csr = eng.CSR.from_arrays(rows, cols, ro, ci)
        bw = csr.calculate_block_size(200 << 30)
        for alpha in alphas:
            best, stats = 1e30, None
            for _ in range(3):
                t0 = time.perf_counter()
                st, perm, clusters, s = eng.cluster_rows_device(rows, cols, ro, ci, bw, alpha)
                wall = (time.perf_counter() - t0) * 1e3
                assert st == 0, st
                if s["elapsed_ms"] < best:
                    best, stats = s["elapsed_ms"], s
            line = (f"{name:18s} rows {rows:7d} bins {-(-cols // bw):6d} alpha {alpha}: device {best:9.1f} ms (wall {wall:8.1f}) {clusters:6d} clusters "
                    f"{stats['passes']:6d} passes {stats['similarities'] / 1e6:8.2f} M judged {stats['exact_similarities']:8d} exact {stats.get('dropped_seeds', -1):6d} dropped {stats.get('passes_ahead', -1):6d} ahead")
            if check:
                pipe = eng.Pipeline(csr, alpha=alpha, delta=0.3, block_size=bw, device=-1)
                same = np.array_equal(pipe.array("reorderedRows"), perm) and pipe.num_clusters == clusters
                line += f"  host-identical={same}"
            print(line, flush=True)


if __name__ == "__main__":
    main()
