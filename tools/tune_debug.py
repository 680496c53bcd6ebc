import sys, time, os
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
import torch, numpy as np
import bsmr_amd as eng, synth
rows, cols, ro, ci = synth.bernoulli()
csr = eng.CSR.from_arrays(rows, cols, ro, ci)
arrays = eng.Pipeline(csr, alpha=0.3, delta=0.0, device=-1).arrays()
K = 512; mode = eng.COMPUTE_BF16
dev = torch.device("cuda:0")
A, B = eng.make_data(rows * K, 5489), eng.make_data(cols * K, 5490)
tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
tP = torch.zeros(csr.nnz, dtype=torch.float32, device=dev)
def loop(plan, label):
    for _ in range(20): eng.sddmm(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): eng.sddmm(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, 0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    kt = eng.sddmm_timed(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, 0, warmup=5, iters=200)
    print(f"{label}: enqueue {1e6*(t1-t0)/200:.1f} us/call, wall {1e6*(t2-t0)/200:.1f} us/step, timed {kt}", flush=True)
for label, opts in (("stream", eng.plan_options()), ("shared forced H8 b16", eng.plan_options(dense_engine=eng.ENGINE_SHARED, tile_group=8, tile_blocks_per_item=16)),
                    ("tuned", eng.plan_options(dense_engine=eng.ENGINE_TUNED))):
    st, plan = eng.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0, options=opts)
    assert st == eng.OK
    if label == "tuned":
        print(eng.plan_tune(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, 0))
    loop(plan, label)
    eng.plan_destroy(plan)
