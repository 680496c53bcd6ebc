#!/usr/bin/env python3
"""Lab: the sweep engine (BSMR_ENGINE_SWEEP) against the streaming engine on one workload.
   python tools/sweep_lab.py [workload ...] [--mode f16|bf16] [--shapes W:PW:SB:PERCU,...] [--iters N]
Per shape: dense-kernel and whole-call microseconds (bsmr_sddmm_timed) and whether P equals the streaming
engine's P bit for bit (same casts, same MFMA, same k order)."""
import argparse
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))
sys.path.insert(0, str(REPO))
import numpy as np
import torch

import bsmr_amd as eng
import synth
import bench


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["nips_k128_dense"])
    ap.add_argument("--mode", default=None)
    ap.add_argument("--shapes", default="")
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(dev).cuda_stream
    for name in args.workloads:
        gen, kw, K, alpha, delta = bench.WORKLOADS[name]
        rows, cols, ro, ci = getattr(synth, gen)(**kw)
        mode = {"f16": 0, "bf16": 1}[args.mode] if args.mode else (1 if name.startswith("dlmc") else 0)
        csr = eng.CSR.from_arrays(rows, cols, ro, ci)
        pipe = eng.Pipeline(csr, alpha=alpha, delta=delta, device=0)
        arrays = pipe.arrays()
        A = eng.make_data(rows * K, 5489)
        B = eng.make_data(cols * K, 5490)
        tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)

        def run(options, label):
            st, plan = eng.plan_from_arrays(rows, cols, csr.nnz, arrays, device=0, options=options)
            if st != eng.OK:
                print(f"  {label}: plan status {st}")
                return None
            tP = torch.full((csr.nnz,), float("nan"), dtype=torch.float32, device=dev)
            eng.sddmm(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, stream)
            torch.cuda.synchronize()
            t = eng.sddmm_timed(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, stream, 5, args.iters)
            got = tP.cpu().numpy()
            eng.plan_destroy(plan)
            print(f"  {label:34s} total {t['total_ms']*1e3:8.2f} us  convert {t['convert_ms']*1e3:6.2f}  dense {t['dense_ms']*1e3:8.2f}  "
                  f"sparse {t['sparse_ms']*1e3:7.2f}", end="")
            return got

        print(f"{name}: {rows} x {cols}, nnz {csr.nnz}, K {K}, mode {mode}")
        ref = run(eng.plan_options(), "stream (rules)")
        print()
        shapes = [tuple(int(x) for x in s.split(":")) for s in args.shapes.split(",") if s] or [(0, 0, 0, 0)]
        for w, pw, sb, percu in shapes:
            for fp32 in ((1, 0) if K <= 128 else (0,)):
                got = run(eng.plan_options(dense_engine=eng.ENGINE_SWEEP, sweep_panels=pw, sweep_strip_blocks=sb, sweep_fp32=fp32,
                                           sweep_waves=w, sweep_per_cu=percu),
                          f"sweep W={w} PW={pw} SB={sb} x{percu} {'fp32' if fp32 else '16bit'}")
                if got is None:
                    continue
                same = np.array_equal(got.view(np.uint32), ref.view(np.uint32))
                nan = int(np.isnan(got).sum())
                print(f"   bitwise equal: {same}  nan: {nan}  max|diff| {np.nanmax(np.abs(got - ref)):.3g}")


if __name__ == "__main__":
    main()
