#!/bin/bash
# GPU box: BASELINE configs[4] - 4096^2 Bernoulli(0.1), K = 512, bf16, delta in {0, .1, .3, .5, 1.1}: time, MFMA rate
# executed and algorithmic GB/s per delta, once with the plan's own rules and once pinned to the RPHM's split.
OUT=gpurun_out/dlmc_sweep
mkdir -p $OUT
for pin in 0 1; do
  for wl in dlmc_k512_dense dlmc_k512_d01 dlmc_k512_d03 dlmc_k512_d05 dlmc_k512_sparse; do
    if [ $pin = 1 ]; then export BSMR_PROMOTE_AVERAGE=0 BSMR_FOLD_DENSE_BELOW=0; else unset BSMR_PROMOTE_AVERAGE BSMR_FOLD_DENSE_BELOW; fi
    timeout -k 10 300 python3 bench.py --workload $wl --mode bf16 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${wl}_pin$pin.json 2> $OUT/${wl}_pin$pin.err || echo "$wl failed"
  done
done
python3 - > gpurun_out/dlmc_sweep.md <<'PY'
import json
print("# BASELINE configs[4]: 4096^2 Bernoulli(0.1) (nnz 1 678 821), K = 512, bf16 operands, one MI355X\n")
print("| delta | plan | us / SDDMM | GFLOP/s | convert / dense / residue us | dense blocks | MFMA executed TFLOP/s (of 2500) | dominant kernel alg. GB/s (of 8000) |")
print("|---|---|---|---|---|---|---|---|")
for wl, delta in (("dlmc_k512_dense", 0.0), ("dlmc_k512_d01", 0.1), ("dlmc_k512_d03", 0.3), ("dlmc_k512_d05", 0.5), ("dlmc_k512_sparse", 1.1)):
    for pin, label in ((1, "RPHM split as is"), (0, "plan rules (promotion)")):
        try:
            d = json.loads(open(f"gpurun_out/dlmc_sweep/{wl}_pin{pin}.json").read().strip().splitlines()[-1])
        except Exception as e:
            print(f"| {delta} | {label} | failed {e} |")
            continue
        k, r = d["kernels_ms"], d["roofline"]
        print(f"| {delta} | {label} | {d['ms_per_step'] * 1e3:.1f} | {d['value']:.0f} | {k['convert_ms'] * 1e3:.1f} / {k['dense_ms'] * 1e3:.1f} / {k['sparse_ms'] * 1e3:.1f} | "
              f"{d['config']['dense_blocks']} | {r.get('mfma_executed_tflops', '-')} | {r['kernel']}: {r['achieved']:.0f} ({r['frac']:.3f}) |")
PY
cat gpurun_out/dlmc_sweep.md
