#!/usr/bin/env python3
"""GPU box: the plan-time rules as bsmr_plan_tune's variants (plans created with k_hint) against the knob sweep of
tools/promote_lab.py: per pattern and K, whole-call microseconds of every BSMR_PROMOTE_AVERAGE level (plans tuned as
usual), and what a hinted plan chose and measured.
usage: khint_lab.py [workload ...]        (the K of the workload is replaced by 32, 128, 512)"""
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

names = sys.argv[1:] or ["mycielskian15_k128", "mycielskian14_k128", "nips_k32_hybrid", "dlmc_k512_d01",
                         "cop20k_k128_hybrid", "wathen100_k128", "trefethen20000_k128", "nips_k128_dense"]
dev = torch.device("cuda:0")
s = torch.cuda.current_stream(dev).cuda_stream


def whole_call_us(plan, K, A, B, P, iters=50):
    best = 1e30
    for _ in range(3):
        for _ in range(3):
            eng.sddmm(plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, s)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            eng.sddmm(plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, s)
        e1.record()
        e1.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
    return best


worst = 0.0
for name in names:
    gen, kwargs, _, alpha, delta = WORKLOADS[name]
    rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
    csr = eng.CSR.from_arrays(rows, cols, ro, ci)
    arrays = eng.Pipeline(csr, alpha=alpha, delta=delta, device=-1).arrays()
    for K in (32, 128, 512):
        A = torch.from_numpy(eng.make_data(rows * K, 5489)).to(dev)
        B = torch.from_numpy(eng.make_data(cols * K, 5490)).to(dev)
        P = torch.zeros(csr.nnz, dtype=torch.float32, device=dev)
        knobs = {}
        for level in (0, 16, 20, 24):
            st, plan = eng.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0,
                                            options=eng.plan_options(dense_engine=eng.ENGINE_TUNED, promote_average=level))
            assert st == 0, st
            eng.plan_tune(plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, s)
            knobs[level] = round(whole_call_us(plan, K, A, B, P), 2)
            eng.plan_destroy(plan)
        st, plan = eng.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0,
                                        options=eng.plan_options(dense_engine=eng.ENGINE_TUNED, k_hint=K))
        assert st == 0, st
        report = eng.plan_tune(plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, s)
        hinted = round(whole_call_us(plan, K, A, B, P), 2)
        eng.plan_destroy(plan)
        over = hinted / min(knobs.values()) - 1.0
        worst = max(worst, over)
        print(json.dumps({"workload": name, "K": K, "promote_average_us": knobs, "hinted_us": hinted, "variant": report["variant"],
                          "variant_us": report["variant_us"], "over_best_knob_pct": round(100 * over, 1)}), flush=True)
        del A, B, P
print(f"worst: hinted plan {100 * worst:.1f} % over the best knob")
