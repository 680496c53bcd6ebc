#!/usr/bin/env python3
"""GPU box: total / per-kernel microseconds of a few workloads with the current library (A/B runs of a change)."""
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
sys.path.insert(0, str(REPO))
import torch  # noqa: E402

import bsmr_amd as eng  # noqa: E402
import synth  # noqa: E402
from bench import WORKLOADS  # noqa: E402

names = sys.argv[1:] or ["nips_k128_dense", "nips_k32_hybrid", "mycielskian15_k128", "mycielskian14_k128", "dlmc_k512_dense",
                         "trefethen20000_k512", "cop20k_k128_hybrid"]
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
    st, plan = eng.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0)
    assert st == 0, st
    best = min((eng.sddmm_timed(plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, s, warmup=5, iters=200)
                for _ in range(5)), key=lambda t: t["total_ms"])
    eng.plan_destroy(plan)
    print(json.dumps({"workload": name, **{k: round(v * 1e3, 2) for k, v in best.items()}}), flush=True)
