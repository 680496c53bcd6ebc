#!/usr/bin/env python3
"""GPU box: residue entries per workgroup item (BSMR_SPARSE_ENTRIES_PER_WG) with the tuned lane shapes,
microseconds per SDDMM.  Usage: python tools/residue_items_lab.py [workload ...]"""
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

names = sys.argv[1:] or ["trefethen20000_k32", "trefethen20000_k128", "trefethen20000_k512", "wathen100_k32", "wathen100_k128",
                         "wathen100_k512", "cop20k_k128_hybrid"]
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
    line = {"workload": name}
    for per in ("32", "64", "128", "256", "512"):
        os.environ["BSMR_SPARSE_ENTRIES_PER_WG"] = per
        st, plan = eng.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0)
        assert st == 0, st
        best = min((eng.sddmm_timed(plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, s, warmup=5, iters=100)
                    for _ in range(3)), key=lambda t: t["total_ms"])
        eng.plan_destroy(plan)
        line[per] = round(best["total_ms"] * 1e3, 2)
    print(json.dumps(line), flush=True)
