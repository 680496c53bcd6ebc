#!/usr/bin/env python3
"""GPU box: row clustering on the host against the device for the bench workloads' patterns (the rule of
BSMR::rowReordering that picks one of the two).  usage: cluster_rule_lab.py [alpha ...]"""
import os
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
import bsmr_amd as eng   # noqa: E402
import synth             # noqa: E402

PATTERNS = [("nips-like", lambda: synth.nips_like()), ("mycielskian14", lambda: synth.mycielskian_pattern(k=14)),
            ("mycielskian15", lambda: synth.mycielskian_pattern(k=15)), ("trefethen20000", lambda: synth.trefethen_pattern(n=20000)),
            ("wathen100", lambda: synth.wathen_pattern(nx=100, ny=100)), ("cop20k-like", lambda: synth.banded_mesh_like()),
            ("bernoulli4096", lambda: synth.bernoulli()), ("reddit-like shard", lambda: synth.reddit_shard_like())]


def main():
    alphas = [float(a) for a in sys.argv[1:]] or [0.3, 0.9]
    for name, make in PATTERNS:
        rows, cols, ro, ci = make()
        csr = eng.CSR.from_arrays(rows, cols, ro, ci)
        for alpha in alphas:
            line = f"{name:18s} rows {rows:7d} nnz/row {ci.size / max(rows, 1):7.1f} alpha {alpha}:"
            for where in ("host", "device"):
                if where == "host" and name == "reddit-like shard" and alpha > 0.5:
                    continue
                os.environ["BSMR_CLUSTER"] = where
                best = 1e30
                for _ in range(2):
                    pipe = eng.Pipeline(csr, alpha=alpha, delta=0.3, device=0)
                    best = min(best, pipe.row_reordering_ms)
                    clusters = pipe.num_clusters
                    del pipe
                line += f"  {where} {best:9.1f} ms ({clusters} clusters)"
            print(line, flush=True)


if __name__ == "__main__":
    main()
