#!/usr/bin/env python3
"""GPU box: SDDMM time of one workload's matrix over the reference's delta grid (same row order)."""
import sys
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
sys.path.insert(0, str(REPO))
import hostinfo
hostinfo.limit_openmp_threads()
import torch
import bsmr_amd as eng, synth
from bench import WORKLOADS
name = sys.argv[1]
gen, kwargs, K, alpha, _ = WORKLOADS[name]
rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
dev = torch.device("cuda:0")
csr = eng.CSR.from_arrays(rows, cols, ro, ci)
A = torch.from_numpy(eng.make_data(rows * K, 5489)).to(dev)
B = torch.from_numpy(eng.make_data(cols * K, 5490)).to(dev)
P = torch.zeros(csr.nnz, dtype=torch.float32, device=dev)
s = torch.cuda.current_stream(dev).cuda_stream
for alpha in [float(a) for a in (sys.argv[2].split(",") if len(sys.argv) > 2 else [str(alpha)])]:
    pipe = eng.Pipeline(csr, alpha=alpha, delta=0.0, device=0)
    for delta in (0.0, 0.1, 0.3, 0.5, 0.7, 0.9, 1.1):
        pipe.resplit(delta, device=0)
        t = eng.sddmm_timed(pipe.plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, s, warmup=5, iters=100)
        st = pipe.plan_stats()
        print(f"{name} alpha={alpha} delta={delta}: {t['total_ms'] * 1e3:.1f} us  {2.0 * csr.nnz * K / (t['total_ms'] * 1e6):.0f} GFLOP/s  "
              f"(convert {t['convert_ms'] * 1e3:.1f} dense {t['dense_ms'] * 1e3:.1f} sparse {t['sparse_ms'] * 1e3:.1f}; "
              f"dense blocks {st['num_dense_blocks']}, dense nnz {st['num_dense_entries']}, sparse nnz {st['num_sparse_entries']})", flush=True)
