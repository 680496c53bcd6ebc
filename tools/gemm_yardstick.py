#!/usr/bin/env python3
"""Lab yardstick (never the product path): the library GEMM of the box - torch.mm, i.e. hipBLASLt /
rocBLAS - on the shapes of the GEMM engine's bench lines, as a DENSE product C = A B^T with bf16 / fp16 inputs.  Printed
beside it: what the GEMM engine's dense kernel takes for the masked product on the same shape (profiles/r04_*_bench.json).
The library writes all M x N values of C (as 16-bit, and as fp32 where torch offers out_dtype); the engine writes the
stored entries only, after an epilogue the library does not have - a ceiling for the K loop, not a like-for-like race.
  python tools/gemm_yardstick.py"""
import json
import time
from pathlib import Path

import torch

REPO = Path(__file__).resolve().parent.parent
dev = torch.device("cuda:0")


def timed(fn, iters=200):
    for _ in range(20):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    best = 1e30
    for _ in range(3):
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


cases = [("configs[4] 4096^2 K=512 bf16", 4096, 4096, 512, torch.bfloat16, "r04_dlmc_rules_bench.json"),
         ("nips-like 1500 x 12419 K=512 fp16", 1500, 12419, 512, torch.float16, "r04_nips_k512_bench.json"),
         ("nips-like 1500 x 12419 K=128 fp16", 1500, 12419, 128, torch.float16, "r04_nips_k128_bench.json")]
for name, M, N, K, dt, line in cases:
    a = torch.randn(M, K, device=dev, dtype=dt)
    b = torch.randn(N, K, device=dev, dtype=dt)          # B as the SDDMM has it: a column = K contiguous values
    out16 = torch.empty(M, N, device=dev, dtype=dt)
    us16 = timed(lambda: torch.mm(a, b.t(), out=out16))
    us32 = None
    try:
        us32 = timed(lambda: torch.mm(a, b.t(), out_dtype=torch.float32))
    except (TypeError, RuntimeError) as e:
        note = str(e).splitlines()[0][:80]
    flop = 2.0 * M * N * K
    engine = json.loads((REPO / "profiles" / line).read_text())
    print(f"{name}: library GEMM {us16:.1f} us to 16-bit C ({flop / us16 / 1e6:.0f} TFLOP/s)"
          + (f", {us32:.1f} us to fp32 C ({flop / us32 / 1e6:.0f} TFLOP/s)" if us32 else f" (fp32 C: not offered - {note})")
          + f"; GEMM engine's dense kernel on the masked product {engine['kernels_ms']['dense_ms'] * 1e3:.1f} us "
          f"({flop / (engine['kernels_ms']['dense_ms'] * 1e3) / 1e6:.0f} TFLOP/s executed over full tiles), whole step {engine['ms_per_step'] * 1e3:.1f} us", flush=True)
