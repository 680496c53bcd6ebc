#!/usr/bin/env python3
"""GPU box: the PCIe-inclusive rate of one SDDMM - operands in host memory, result back in host memory
(bsmr_sddmm_host: upload A and B, SDDMM, download P; the reference's sddmm_gpu(Matrix ...) shape, src/sddmmKernel.cu:2518-2538).
Never the bench's `value`: wall time of the whole call, best of five."""
import sys
import time
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
sys.path.insert(0, str(REPO))
import bsmr_amd as eng   # noqa: E402
import synth             # noqa: E402
import bench             # noqa: E402

for name in sys.argv[1:] or ["nips_k128_dense", "cop20k_k128_hybrid", "mycielskian15_k128"]:
    gen, kwargs, K, alpha, delta = bench.WORKLOADS[name]
    rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
    csr = eng.CSR.from_arrays(rows, cols, ro, ci)
    pipe = eng.Pipeline(csr, alpha=alpha, delta=delta, device=0)
    A, B = eng.make_data(rows * K, 5489), eng.make_data(cols * K, 5490)
    best, inner = 1e30, 0.0
    for _ in range(5):
        t0 = time.perf_counter()
        P, ms = eng.sddmm_host(pipe.plan, K, A, B, csr.nnz)
        best = min(best, (time.perf_counter() - t0) * 1e3)
        inner = ms
    mb = (A.nbytes + B.nbytes + P.nbytes) / 1e6
    print(f"{name}: host buffers -> host result {best:.3f} ms ({2 * csr.nnz * K / best / 1e6:.0f} GFLOP/s), {mb:.1f} MB over PCIe "
          f"({mb / best:.1f} GB/s); device part of the call {inner * 1e3:.1f} us", flush=True)
