#!/usr/bin/env python3
"""GPU box, LAB build (make -B LAB=1 lib/libbsmr_hip.so): in-kernel phase stamps of the tiles dense kernel."""
import os, sys, time
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python")); sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import numpy as np, torch
import bsmr_amd as eng, synth, bench
cases = sys.argv[1:] or ["nips_k128_dense:1", "nips_k128_dense:4", "mycielskian15_k128:2", "dlmc_k512_dense:2"]
dev = torch.device("cuda:0")
for case in cases:
    wl, _, rest = case.partition(":")
    g, _, b = rest.partition(":")
    mode_name = "bf16" if wl.startswith("dlmc") else "f16"
    gen, kwargs, K, alpha, delta = bench.WORKLOADS[wl]
    rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
    os.environ["BSMR_TILE_GROUP"] = g or "1"
    os.environ["BSMR_DENSE_ENGINE"] = os.environ.get("LAB_ENGINE", "tiles")
    if b: os.environ["BSMR_TILE_BLOCKS"] = b
    else: os.environ.pop("BSMR_TILE_BLOCKS", None)
    csr = eng.CSR.from_arrays(rows, cols, ro, ci)
    pipe = eng.Pipeline(csr, alpha=alpha, delta=delta, device=0)
    A = eng.make_data(rows * K, 5489); B = eng.make_data(cols * K, 5490)
    tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    tP = torch.zeros(ci.size, dtype=torch.float32, device=dev)
    sh = torch.cuda.current_stream(dev).cuda_stream
    for _ in range(20):
        eng.sddmm(pipe.plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), (eng.COMPUTE_BF16 if mode_name == 'bf16' else eng.COMPUTE_F16), sh)
    torch.cuda.synchronize()
    kt = eng.sddmm_timed(pipe.plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), (eng.COMPUTE_BF16 if mode_name == 'bf16' else eng.COMPUTE_F16), sh, warmup=5, iters=50)
    print(f"## {case}: dense {kt['dense_ms']*1e3:.2f} us (stamp build, no stamps taken)", flush=True)
    os.environ["BSMR_TILE_STAMPS"] = "1"
    sys.stderr.flush()
    eng.sddmm(pipe.plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), (eng.COMPUTE_BF16 if mode_name == 'bf16' else eng.COMPUTE_F16), sh)
    torch.cuda.synchronize()
    os.environ.pop("BSMR_TILE_STAMPS")
    del pipe
