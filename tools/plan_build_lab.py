#!/usr/bin/env python3
"""GPU box: plan build phases for one workload's matrix at a given delta."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python")); sys.path.insert(0, str(REPO))
import hostinfo; hostinfo.limit_openmp_threads()
import bsmr_amd as eng, synth
from bench import WORKLOADS
name, delta = sys.argv[1], float(sys.argv[2])
gen, kwargs, K, alpha, _ = WORKLOADS[name]
rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
csr = eng.CSR.from_arrays(rows, cols, ro, ci)
t0 = time.perf_counter(); host = eng.Pipeline(csr, alpha=alpha, delta=delta, device=-1); t_host = time.perf_counter() - t0
arrays = host.arrays()
for rep in range(2):
    t0 = time.perf_counter(); st, plan = eng.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0); t_plan = time.perf_counter() - t0
    assert st == 0
    eng.plan_destroy(plan)
print(f"{name} delta={delta}: host pipeline {t_host * 1e3:.0f} ms (row {host.row_reordering_ms:.0f}, col {host.col_reordering_ms:.0f}, rphm {host.rphm_ms:.0f}); "
      f"bsmr_plan_create {t_plan * 1e3:.0f} ms")
