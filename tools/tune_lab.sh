#!/bin/bash
# GPU box: bsmr_plan_tune - its parity test, then a bench line per workload with the tuned dense engine.
timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -s -k "tuned_plans" > gpurun_out/tuned.log 2>&1; echo "test rc=$?"
grep -v amdgpu gpurun_out/tuned.log | tail -8
for w in ${WLS:-nips_k128_dense:f16 dlmc_k512_dense:bf16 nips_k512_dense:f16 mycielskian15_k128:f16 cop20k_blocks_k128_hybrid:f16}; do
  set -- ${w%%:*} ${w##*:}
  timeout -k 10 300 python3 bench.py --workload $1 --mode $2 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(d['config']['workload'][:26], round(d['ms_per_step']*1e3,2), d['kernels_ms'], d['dense_engine'], d['plan_build_s'], d.get('parity_mismatches_vs_cpu'))"
done
