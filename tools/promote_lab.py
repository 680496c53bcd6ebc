#!/usr/bin/env python3
"""GPU box: panels whose residue is promoted to dense blocks (BSMR_PROMOTE_AVERAGE = entries per 16-column block a
panel's residue must average; 0 = off),
microseconds per SDDMM.  Usage: python tools/promote_lab.py [--from 0,12,16,20,28,40] [workload ...]"""
import json
import os
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
sys.path.insert(0, str(REPO))
import torch  # noqa: E402

import bsmr_amd as eng  # noqa: E402
import synth  # noqa: E402
from bench import WORKLOADS  # noqa: E402

args = sys.argv[1:]
levels = ["0", "16", "20", "24"]
knob = "BSMR_PROMOTE_AVERAGE"
if args and args[0] in ("--from", "--head"):   # --head: leading blocks of panels that do not qualify (BSMR_PROMOTE_HEAD)
    knob = "BSMR_PROMOTE_AVERAGE" if args[0] == "--from" else "BSMR_PROMOTE_HEAD"
    levels = args[1].split(",")
    args = args[2:]
names = args or ["mycielskian15_k128", "mycielskian15_k32", "mycielskian14_k128", "nips_k32_hybrid", "dlmc_k512_d01",
                 "cop20k_k128_hybrid", "wathen100_k128", "trefethen20000_k128"]
dev = torch.device("cuda:0")
s = torch.cuda.current_stream(dev).cuda_stream
for name in names:
    gen, kwargs, K, alpha, delta = WORKLOADS[name]
    rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
    csr = eng.CSR.from_arrays(rows, cols, ro, ci)
    arrays = eng.Pipeline(csr, alpha=alpha, delta=delta, device=-1).arrays()
    A = torch.from_numpy(eng.make_data(rows * K, 5489)).to(dev)
    B = torch.from_numpy(eng.make_data(cols * K, 5490)).to(dev)
    P = torch.zeros(csr.nnz, dtype=torch.float32, device=dev)
    for level in levels:
        os.environ[knob] = level
        st, plan = eng.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0)
        assert st == 0, st
        raw = eng.PlanStats()
        assert eng.hip().bsmr_plan_get_stats(plan, eng.C.byref(raw)) == 0
        stats = {k: getattr(raw, k) for k, _ in eng.PlanStats._fields_}
        best = min((eng.sddmm_timed(plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, s, warmup=5, iters=100)
                    for _ in range(3)), key=lambda t: t["total_ms"])
        eng.plan_destroy(plan)
        print(json.dumps({"workload": name, knob: int(level), "promoted": stats["promoted_sparse_entries"],
                          "dense_blocks": stats["num_dense_blocks"], "dense": stats["num_dense_entries"],
                          "sparse": stats["num_sparse_entries"], **{k: round(v * 1e3, 2) for k, v in best.items()}}),
              flush=True)
