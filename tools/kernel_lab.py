#!/usr/bin/env python3
"""Kernel tuning lab (GPU box): times the dense / sparse / convert kernels of one
workload under several plan-time knobs (environment variables read by
bsmr_plan_create).  Usage: python tools/kernel_lab.py [workload] [mode]"""
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

name = sys.argv[1] if len(sys.argv) > 1 else "nips_k128_dense"
mode_name = sys.argv[2] if len(sys.argv) > 2 else "f16"
mode = {"f16": 0, "bf16": 1, "f32": 2}[mode_name]
gen, kwargs, K, alpha, delta = WORKLOADS[name]
rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
dev = torch.device("cuda:0")
csr = eng.CSR.from_arrays(rows, cols, ro, ci)
base = eng.Pipeline(csr, alpha=alpha, delta=delta, device=-1)   # host arrays once
arrays = base.arrays()
A = torch.from_numpy(eng.make_data(rows * K, 5489)).to(dev)
B = torch.from_numpy(eng.make_data(cols * K, 5490)).to(dev)
P = torch.zeros(csr.nnz, dtype=torch.float32, device=dev)
s = torch.cuda.current_stream(dev).cuda_stream

knobs = [{}]
if "--colorder" in sys.argv:
    for st in (0, 1, 2):
        for bpw in (16, 32):
            knobs.append({"BSMR_OUTPUT_MODE": str(st), "BSMR_DENSE_BLOCKS_PER_WG": str(bpw)})
else:
    for h in (1, 2, 4):
        for bpw in (8, 16, 32):
            for batch in (4, 8):
                knobs.append({"BSMR_DENSE_GROUP": str(h), "BSMR_DENSE_BLOCKS_PER_WG": str(bpw), "BSMR_DENSE_BATCH": str(batch)})
if "--sparse" in sys.argv:
    knobs = [{}]
    for lpe in (4, 8, 16):
        for e in (64, 128, 256, 512, 1024):
            knobs.append({"BSMR_SPARSE_LPE": str(lpe), "BSMR_SPARSE_ENTRIES_PER_WG": str(e)})
ALL = ("BSMR_DENSE_GROUP", "BSMR_DENSE_BLOCKS_PER_WG", "BSMR_DENSE_BATCH", "BSMR_SPARSE_LPE",
       "BSMR_SPARSE_ENTRIES_PER_WG", "BSMR_FORCE_TILE32", "BSMR_COLUMN_ORDER", "BSMR_OUTPUT_MODE")
for kn in knobs:
    for k in ALL:
        os.environ.pop(k, None)
    os.environ.update(kn)
    st, plan = eng.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0)
    assert st == 0, st
    best = None
    for rep in range(3):
        t = eng.sddmm_timed(plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), mode, s, warmup=5, iters=100)
        if best is None or t["total_ms"] < best["total_ms"]:
            best = t
    eng.plan_destroy(plan)
    print(json.dumps({"knobs": kn, **{k: round(v * 1e3, 2) for k, v in best.items()}}), flush=True)
print("units: microseconds per launch (best of 3 x 100)")
