#!/usr/bin/env python3
"""GPU box: bsmr_sddmm called directly vs replayed from a captured graph (default workload)."""
import sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python")); sys.path.insert(0, str(REPO))
import hostinfo; hostinfo.limit_openmp_threads()
import torch
import bsmr_amd as eng, synth
from bench import WORKLOADS
name = sys.argv[1] if len(sys.argv) > 1 else "nips_k128_dense"
gen, kwargs, K, alpha, delta = WORKLOADS[name]
rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
dev = torch.device("cuda:0")
csr = eng.CSR.from_arrays(rows, cols, ro, ci)
pipe = eng.Pipeline(csr, alpha=alpha, delta=delta, device=0)
A = torch.from_numpy(eng.make_data(rows * K, 5489)).to(dev); B = torch.from_numpy(eng.make_data(cols * K, 5490)).to(dev)
P = torch.zeros(csr.nnz, dtype=torch.float32, device=dev)
eng.hip().bsmr_plan_reserve(pipe.plan, K)
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    s = side.cuda_stream
    for _ in range(20): eng.sddmm(pipe.plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, s)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): eng.sddmm(pipe.plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, s)
    torch.cuda.synchronize(); direct = (time.perf_counter() - t0) / 500 * 1e6
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        eng.sddmm(pipe.plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, side.cuda_stream)
    for _ in range(20): g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(500): g.replay()
    torch.cuda.synchronize(); replay = (time.perf_counter() - t0) / 500 * 1e6
    g8 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g8, stream=side):
        for _ in range(8): eng.sddmm(pipe.plan, K, A.data_ptr(), B.data_ptr(), P.data_ptr(), 0, side.cuda_stream)
    for _ in range(5): g8.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(100): g8.replay()
    torch.cuda.synchronize(); replay8 = (time.perf_counter() - t0) / 800 * 1e6
print(f"{name}: direct {direct:.2f} us per SDDMM, graph replay {replay:.2f} us, graph of 8 calls {replay8:.2f} us per SDDMM")
