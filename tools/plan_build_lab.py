#!/usr/bin/env python3
"""Where plan creation spends its time on the GPU box: host pipeline phases and bsmr_plan_build_times, for the first
plan of the process (HIP initialisation, code-object load) and for a second one.
usage: plan_build_lab.py [workload ...]"""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
import bsmr_amd as eng   # noqa: E402
import synth             # noqa: E402
sys.path.insert(0, str(REPO))
import bench             # noqa: E402


def main():
    names = sys.argv[1:] or ["nips_k128_dense", "mycielskian15_k128", "cop20k_k128_hybrid", "reddit_shard_k256"]
    for name in names:
        gen, kwargs, K, alpha, delta = bench.WORKLOADS[name]
        rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
        csr = eng.CSR.from_arrays(rows, cols, ro, ci)
        for rep in range(2):
            t0 = time.perf_counter()
            pipe = eng.Pipeline(csr, alpha=alpha, delta=delta, device=0)
            total = (time.perf_counter() - t0) * 1e3
            b = pipe.plan_build_ms()
            print(f"{name} #{rep}: pipeline+plan {total:8.1f} ms | rows {pipe.row_reordering_ms:8.1f} cols {pipe.col_reordering_ms:6.1f} "
                  f"rphm {pipe.rphm_ms:6.1f} | plan rules {b['rules_ms']:6.1f} pack {b['pack_ms']:6.1f} upload {b['upload_ms']:6.1f} "
                  f"second format {b['second_format_ms']:6.1f} total {b['total_ms']:6.1f}", flush=True)
            del pipe


if __name__ == "__main__":
    main()
