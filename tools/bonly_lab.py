#!/usr/bin/env python3
"""GPU box: all-sparse plans with the conversion of B alone (forced: BSMR_B_ONLY_WORK_M=0) against the fp32
residue (BSMR_B_ONLY=0) and the default rule, microseconds per SDDMM.  Usage: python tools/bonly_lab.py [workload ...]"""
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

names = sys.argv[1:] or ["cop20k_k128_hybrid", "trefethen20000_k32", "trefethen20000_k128", "trefethen20000_k256",
                         "trefethen20000_k512", "wathen100_k32", "wathen100_k128", "wathen100_k256", "wathen100_k512"]
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
    line = {"workload": name, "nnz_per_operand_row": round(csr.nnz / (rows + cols), 2)}
    for label, env in (("fp32", {"BSMR_B_ONLY": "0"}), ("b_only", {"BSMR_B_ONLY_WORK_M": "0"}), ("default", {})):
        for k in ("BSMR_B_ONLY", "BSMR_B_ONLY_WORK_M"):
            os.environ.pop(k, None)
        os.environ.update(env)
        st, plan = eng.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0)
        assert st == 0, st
        best = min((eng.sddmm_timed(plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, s, warmup=5, iters=100)
                    for _ in range(3)), key=lambda t: t["total_ms"])
        eng.plan_destroy(plan)
        line[label] = {k: round(v * 1e3, 2) for k, v in best.items()}
    print(json.dumps(line), flush=True)
