#!/usr/bin/env python3
"""bench.py -- SDDMM GFLOP/s of the MI355X BSMR-SDDMM engine on BASELINE.json's
metric configuration.

  python bench.py [--gpus N] [--steps K] [--warmup W] [--workload NAME] [--mode f16|bf16|f32]

A "step" is one full SDDMM through the C ABI (bsmr_sddmm): fp32 A and B resident
in HBM -> P (fp32, S's CSR order) resident in HBM, i.e. operand conversion +
dense-block MFMA kernel + residual sparse kernel.  That is the region the
reference times (src/sddmmKernel.cu:2563-2652).  Plan construction (row
clustering, column reordering, RPHM) happens before the timed region, like the
reference's bsmr_reordering time, and is reported separately.

Default workload = BASELINE.json configs[1]: nips-like 1500 x 12419, nnz 746316,
K=128, delta=0.0 (dense-block MFMA path only), one MI355X.  The real nips.mtx is
not available offline; the pattern is the seeded stand-in
synth.nips_like().  Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python"))

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)
MFMA_PEAK_TFLOPS = {"f16": 2500.0, "bf16": 2500.0, "f32": 157.3}

WORKLOADS = {
    # name: (generator, generator kwargs, K, alpha, delta)
    "nips_k128_dense": ("nips_like", {}, 128, 0.3, 0.0),      # BASELINE configs[1]
    "nips_k32_hybrid": ("nips_like", {}, 32, 0.3, 0.3),       # configs[0] on the GPU
    "nips_k512_dense": ("nips_like", {}, 512, 0.3, 0.0),
    "cop20k_k128_hybrid": ("banded_mesh_like", {}, 128, 0.3, 0.3),  # configs[2]
    "dlmc_k512_dense": ("bernoulli", {}, 512, 0.3, 0.0),      # configs[4], delta = 0 point of the sweep
    "dlmc_k512_d01": ("bernoulli", {}, 512, 0.3, 0.1),
    "dlmc_k512_sparse": ("bernoulli", {}, 512, 0.3, 1.1),     # configs[4], delta = 1.1 point
}
BASELINE_CONFIG = {
    "nips_k128_dense": "BASELINE configs[1] stand-in; real nips.mtx unavailable offline",
    "nips_k32_hybrid": "BASELINE configs[0] shape run on the GPU",
    "nips_k512_dense": "K=512 point of the metric",
    "cop20k_k128_hybrid": "BASELINE configs[2] stand-in; real cop20k_A.mtx unavailable offline",
    "dlmc_k512_dense": "BASELINE configs[4] stand-in (4096^2, 90 % sparse), delta=0",
    "dlmc_k512_d01": "BASELINE configs[4] stand-in, delta=0.1",
    "dlmc_k512_sparse": "BASELINE configs[4] stand-in, delta=1.1 (sparse path only)",
}


def algorithmic_bytes(rows, cols, ro, ci, K, elem_bytes):
    """e_in*K*(M_nz + N_nz) + 4*nnz (P) + 4*nnz (col index) + 4*(M+1)."""
    m_nz = int(np.count_nonzero(np.diff(ro.astype(np.int64))))
    n_nz = int(np.unique(ci).size)
    nnz = int(ci.size)
    return elem_bytes * K * (m_nz + n_nz) + 8 * nnz + 4 * (rows + 1)


def cpu_baseline(rows, cols, ro, ci, K, A, B, budget_s=12.0):
    """Oracle sddmm_cpu (restatement of the reference's OpenMP host path) timed on
    this box's host cores over whole passes of the same workload."""
    lib = C.CDLL(str(REPO / "oracle" / "liboracle.so"))
    lib.oracle_sddmm_cpu.argtypes = [C.c_uint32] * 3 + [C.c_void_p] * 5
    lib.oracle_num_threads.restype = C.c_int
    P = np.empty(ci.size, dtype=np.float32)
    p = lambda a: a.ctypes.data_as(C.c_void_p)
    call = lambda: lib.oracle_sddmm_cpu(rows, cols, K, p(ro), p(ci), p(A), p(B), p(P))
    call()  # warm-up (page faults, thread pool)
    t0 = time.perf_counter()
    passes = 0
    while True:
        call()
        passes += 1
        dt = time.perf_counter() - t0
        if dt >= budget_s or passes >= 20000:
            break
    gflops = 2.0 * ci.size * K * passes / dt / 1e9
    return {"value": round(gflops, 3), "unit": "GFLOP/s", "cores": int(lib.oracle_num_threads()),
            "kind": "port", "sample": f"{passes} full passes of the same workload in {dt:.1f} s "
            f"({dt / passes * 1e3:.2f} ms per SDDMM)"}, P


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="nips_k128_dense", choices=sorted(WORKLOADS))
    ap.add_argument("--mode", default="f16", choices=["f16", "bf16", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU code path even with one rank (rehearsal on a 1-GPU box)")
    args = ap.parse_args()

    import torch

    import bsmr_amd as eng
    import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    sharded = world > 1 or args.force_sharded
    if sharded:
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29512")
        dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)

    gen, kwargs, K, alpha, delta = WORKLOADS[args.workload]
    mode = {"f16": eng.COMPUTE_F16, "bf16": eng.COMPUTE_BF16, "f32": eng.COMPUTE_F32}[args.mode]
    if sharded:
        import shard
        # weak scaling: rank r owns row-stacked copy r of the workload pattern (own seed)
        def make_pattern(r):
            kw = dict(kwargs)
            kw["seed"] = {"nips_like": 1, "banded_mesh_like": 2, "bernoulli": 4}[gen] + 100 * r
            return getattr(synth, gen)(**kw)
        result = shard.run_sharded(eng, torch, dist, dev, rank, world, make_pattern, K, alpha, delta, mode,
                                   args.steps, args.warmup, scaling="weak")
        if rank == 0:
            result.update({"n_gpus": world, "steps": args.steps, "warmup": args.warmup, "dtype": args.mode})
            result["config"]["workload"] = f"{args.workload} x{world} (weak): " + result["config"]["workload"]
            print(json.dumps(result))
        dist.destroy_process_group()
        return

    rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
    nnz = int(ci.size)
    t0 = time.perf_counter()
    csr = eng.CSR.from_arrays(rows, cols, ro, ci)
    pipe = eng.Pipeline(csr, alpha=alpha, delta=delta, device=local_rank)
    plan_build_s = time.perf_counter() - t0
    stats = pipe.plan_stats()

    A = eng.make_data(rows * K, 5489)
    B = eng.make_data(cols * K, 5490)
    tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    tP = torch.zeros(nnz, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream(dev)
    sh = stream.cuda_stream
    plan = pipe.plan
    eng.hip().bsmr_plan_reserve(plan, K)

    step = lambda: eng.sddmm(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, sh)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    ms_per_step = wall / args.steps * 1e3
    gflops = 2.0 * nnz * K / (ms_per_step * 1e6)

    # per-kernel durations: HIP events on the launch stream, inside the library
    # (bsmr_sddmm_timed brackets `steps` back-to-back launches of each kernel)
    kt = eng.sddmm_timed(plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, sh, warmup=5,
                         iters=args.steps)
    dominant = max(("dense", "sparse", "convert"), key=lambda k: kt[f"{k}_ms"])
    elem = 4 if args.mode == "f32" else 2
    if dominant == "dense":
        alg_bytes = algorithmic_bytes(rows, cols, ro, ci, K, elem)
    elif dominant == "sparse":
        alg_bytes = algorithmic_bytes(rows, cols, ro, ci, K, 4)
    else:
        alg_bytes = (rows + cols) * K * (4 + 2)
    dom_ms = kt[f"{dominant}_ms"]
    achieved = alg_bytes / (dom_ms * 1e-3) / 1e9
    choice = pipe.dense_choice(K)
    exec_flops = choice["tiles"] * 2 * 256 * K
    traffic = None
    tfile = REPO / "profiles" / "traffic.json"
    if tfile.exists():
        traffic = json.loads(tfile.read_text()).get(f"{args.workload}:{args.mode}:{dominant}")

    out = {
        "metric": "SDDMM GFLOP/s", "value": round(gflops, 2), "unit": "GFLOP/s", "n_gpus": 1,
        "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.mode,
        "data": "synthetic",
        "config": {"workload": f"{args.workload}: {gen} {rows}x{cols} nnz={nnz} K={K} alpha={alpha} "
                               f"delta={delta} ({BASELINE_CONFIG[args.workload]})",
                   "boundary": "fp32 A,B in HBM -> fp32 P in HBM via bsmr_sddmm (conversion included)",
                   "dense_blocks": stats["num_dense_blocks"], "dense_tiles": choice["tiles"], "group_size": choice["group_size"], "union_columns": choice["union_columns"], "dense_nnz": stats["num_dense_entries"],
                   "sparse_nnz": stats["num_sparse_entries"]},
        "roofline": {"bound": "hbm", "kernel": dominant, "achieved": round(achieved, 2),
                     "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
                     "traffic": traffic, "algorithmic_bytes": alg_bytes,
                     "kernel_ms": round(dom_ms, 5),
                     "mfma_executed_tflops": round(exec_flops / (kt["dense_ms"] * 1e-3) / 1e12, 2)
                     if kt["dense_ms"] > 0 else None,
                     "mfma_peak_tflops": MFMA_PEAK_TFLOPS[args.mode]},
        "kernels_ms": {k: round(v, 5) for k, v in kt.items()},
        "plan_build_s": round(plan_build_s, 3),
        "host_pipeline_ms": {"row_reordering": round(pipe.row_reordering_ms, 2),
                             "col_reordering": round(pipe.col_reordering_ms, 2),
                             "rphm": round(pipe.rphm_ms, 2)},
    }
    if not args.no_cpu_baseline:
        base, want = cpu_baseline(rows, cols, ro, ci, K, A, B)
        out["cpu_baseline"] = base
        got = tP.cpu().numpy()
        lib = C.CDLL(str(REPO / "oracle" / "liboracle.so"))
        lib.oracle_check_data.argtypes = [C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.oracle_check_data.restype = C.c_uint64
        bad = lib.oracle_check_data(nnz, want.ctypes.data_as(C.c_void_p), got.ctypes.data_as(C.c_void_p), None)
        out["parity_mismatches_vs_cpu"] = int(bad)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
