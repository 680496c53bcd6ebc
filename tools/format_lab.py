#!/usr/bin/env python3
"""GPU box: the two dense formats of a plan (ungrouped streaming kernel vs 4 panels per group) per K, with the
statistics the choice is made from.  Usage: python tools/format_lab.py [workload ...]   (K of the workload is ignored)"""
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

names = sys.argv[1:] or ["mycielskian15_k128", "mycielskian14_k128", "nips_k128_dense", "dlmc_k512_dense", "dlmc_k512_d01"]
dev = torch.device("cuda:0")
s = torch.cuda.current_stream(dev).cuda_stream


def stats_of(plan):
    raw = eng.PlanStats()
    assert eng.hip().bsmr_plan_get_stats(plan, eng.C.byref(raw)) == 0
    return {k: getattr(raw, k) for k, _ in eng.PlanStats._fields_}


for name in names:
    gen, kwargs, _, alpha, delta = WORKLOADS[name]
    rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
    csr = eng.CSR.from_arrays(rows, cols, ro, ci)
    arrays = eng.Pipeline(csr, alpha=alpha, delta=delta, device=-1).arrays()
    P = torch.zeros(csr.nnz, dtype=torch.float32, device=dev)
    plans = {}
    for label, env in (("auto", {}), ("ungrouped", {"BSMR_DENSE_GROUP": "1"}), ("grouped4", {"BSMR_DENSE_GROUP": "4", "BSMR_DENSE_BLOCKS_PER_WG": "16"})):
        for k in ("BSMR_DENSE_GROUP", "BSMR_DENSE_BLOCKS_PER_WG"):
            os.environ.pop(k, None)
        os.environ.update(env)
        st, plan = eng.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0)
        assert st == 0, st
        plans[label] = plan
    st = stats_of(plans["auto"])
    print(json.dumps({"workload": name, **{k: st[k] for k in ("num_dense_blocks", "num_dense_tiles", "union_columns",
                                                               "grouped_group_size", "grouped_dense_tiles", "grouped_union_columns")}}), flush=True)
    for K in (32, 64, 128, 256, 512):
        A = torch.from_numpy(eng.make_data(rows * K, 5489)).to(dev)
        B = torch.from_numpy(eng.make_data(cols * K, 5490)).to(dev)
        line = {"K": K}
        for label, plan in plans.items():
            best = min((eng.sddmm_timed(plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, s, warmup=5, iters=50)
                        for _ in range(3)), key=lambda t: t["dense_ms"])
            line[label] = round(best["dense_ms"] * 1e3, 2)
        g = eng.C.c_uint32(0)
        eng.hip().bsmr_plan_dense_choice(plans["auto"], K, eng.C.byref(g), None, None)
        line["auto_group"] = g.value
        print(json.dumps(line), flush=True)
    for plan in plans.values():
        eng.plan_destroy(plan)
