#!/usr/bin/env python3
"""GPU box: device clustering of one matrix, for rocprofv3 --kernel-trace --stats."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
import hostinfo; hostinfo.limit_openmp_threads()
import bsmr_amd as eng, synth
rows, cols, ro, ci = synth.mycielskian_pattern(15)
for _ in range(2):
    st, perm, clusters, stats = eng.cluster_rows_device(rows, cols, ro, ci, 16, 0.3)
    print(st, clusters, stats, flush=True)
