#!/usr/bin/env python3
"""GPU box: plan statistics (both dense formats) of bench workloads."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
sys.path.insert(0, str(REPO))
import hostinfo
hostinfo.limit_openmp_threads()
import bsmr_amd as eng, synth
from bench import WORKLOADS
for name in sys.argv[1:]:
    gen, kwargs, K, alpha, delta = WORKLOADS[name]
    rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
    pipe = eng.Pipeline(eng.CSR.from_arrays(rows, cols, ro, ci), alpha=alpha, delta=delta, device=0)
    st = pipe.plan_stats()
    print(name, {k: st[k] for k in ("num_dense_blocks", "num_dense_tiles", "union_columns", "grouped_group_size",
                                     "grouped_dense_tiles", "grouped_union_columns", "num_dense_entries")}, flush=True)
