#!/usr/bin/env python3
"""GPU box: parity + timing of the tiles dense engine over group sizes / blocks per item.
usage: tiles_lab.py [workload ...]   (names of bench.WORKLOADS; default: a fixed set)"""
import os, sys, time, json
from pathlib import Path
REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO / "bsmr-sddmm_amd" / "python")); sys.path.insert(0, str(REPO)); sys.path.insert(0, str(REPO / "tests"))
import numpy as np, torch
import bsmr_amd as eng, synth, bench
from conftest import Oracle

names = sys.argv[1:] or ["nips_k128_dense", "nips_k512_dense", "dlmc_k512_dense:bf16", "mycielskian15_k128", "reddit_shard_k256", "nips_k32_hybrid"]
groups = [int(x) for x in os.environ.get("LAB_GROUPS", "1,2,4,8").split(",")]
blocks = [int(x) for x in os.environ.get("LAB_BLOCKS", "0").split(",")]
depths = [int(x) for x in os.environ.get("LAB_DEPTHS", "0").split(",")]
batches = [int(x) for x in os.environ.get("LAB_BATCHES", "0").split(",")]
dev = torch.device("cuda:0")
orc = Oracle()
for name in names:
    wl, _, mode_name = name.partition(":")
    mode = {"": eng.COMPUTE_F16, "f16": eng.COMPUTE_F16, "bf16": eng.COMPUTE_BF16}[mode_name]
    gen, kwargs, K, alpha, delta = bench.WORKLOADS[wl]
    rows, cols, ro, ci = getattr(synth, gen)(**kwargs)
    t0 = time.perf_counter()
    csr = eng.CSR.from_arrays(rows, cols, ro, ci)
    pipe = eng.Pipeline(csr, alpha=alpha, delta=delta, device=0)
    print(f"## {name}: {rows}x{cols} nnz={ci.size} K={K} plan {time.perf_counter()-t0:.2f}s", flush=True)
    A = eng.make_data(rows * K, 5489); B = eng.make_data(cols * K, 5490)
    tA, tB = torch.from_numpy(A).to(dev), torch.from_numpy(B).to(dev)
    want = orc.sddmm_cpu(rows, cols, K, ro, ci, A, B) if ci.size * K < 4e9 else None
    sh = torch.cuda.current_stream(dev).cuda_stream
    def run(label):
        tP = torch.full((ci.size,), float("nan"), dtype=torch.float32, device=dev)
        eng.sddmm(pipe.plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, sh)
        torch.cuda.synchronize()
        got = tP.cpu().numpy()
        bad = -1
        if want is not None:
            if mode == eng.COMPUTE_BF16 and K < 512:
                rel = np.abs(got - want) / np.maximum(np.abs(want), 1e-3); bad = int((~(rel < 2.0 ** -7)).sum())
            else:
                bad, _ = orc.check_data(want, got)
        kt = eng.sddmm_timed(pipe.plan, K, tA.data_ptr(), tB.data_ptr(), tP.data_ptr(), mode, sh, warmup=10, iters=100)
        ch = pipe.dense_choice(K); st = pipe.plan_stats()
        print(f"{label:28s} total {kt['total_ms']*1e3:8.2f} us conv {kt['convert_ms']*1e3:6.2f} dense {kt['dense_ms']*1e3:8.2f} sparse {kt['sparse_ms']*1e3:7.2f}"
              f" | H={ch['group_size']} tiles={ch['tiles']} ucols={ch['union_columns']} idxMB={st['device_index_bytes']/1e6:.1f} nan={int(np.isnan(got).sum())} bad={bad}", flush=True)
    os.environ["BSMR_DENSE_ENGINE"] = "legacy"
    # engine is a plan-creation knob: the legacy numbers come from a second pipeline only when asked
    if os.environ.get("LAB_LEGACY", "0") == "1":
        pipe_l = eng.Pipeline(csr, alpha=alpha, delta=delta, device=0)
        keep = pipe; pipe = pipe_l; run("legacy"); pipe = keep
    os.environ.pop("BSMR_DENSE_ENGINE")
    os.environ.pop("BSMR_TILE_GROUP", None); os.environ.pop("BSMR_TILE_BLOCKS", None)
    run("tiles auto")
    for g in groups:
        for b in blocks:
            for dep in [(d, nb) for d in depths for nb in batches]:
                dep, nb = dep
                if nb: os.environ["BSMR_TILE_BATCH"] = str(nb)
                else: os.environ.pop("BSMR_TILE_BATCH", None)
                os.environ["BSMR_TILE_GROUP"] = str(g)
                if b: os.environ["BSMR_TILE_BLOCKS"] = str(b)
                else: os.environ.pop("BSMR_TILE_BLOCKS", None)
                if dep: os.environ["BSMR_TILE_DEPTH"] = str(dep)
                else: os.environ.pop("BSMR_TILE_DEPTH", None)
                try:
                    run(f"tiles H={g} blk={b or 'auto'} D={dep or 'dflt'} NB={nb or 'dflt'}")
                except Exception as e:
                    print(f"tiles H={g} blocks={b}: {e}", flush=True)
    os.environ.pop("BSMR_TILE_DEPTH", None); os.environ.pop("BSMR_TILE_BATCH", None)
    os.environ.pop("BSMR_TILE_GROUP", None); os.environ.pop("BSMR_TILE_BLOCKS", None)
    del pipe
