#!/bin/bash
# Quick A/B on the GPU box: parity subset, then kernel times of the main workloads.
# EXTRA_ENVS="A=1,B=2 C=3": one run per space-separated group (commas separate variables).
set -e
mkdir -p gpurun_out
WLS=${WLS:-"nips_k128_dense nips_k32_hybrid nips_k512_dense dlmc_k512_dense cop20k_k128_hybrid"}
for env in "" $EXTRA_ENVS; do
  envs=$(echo "$env" | tr ',' ' ')
  echo "#### [$envs]"
  env $envs timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_random or other_k or nips or edge" > gpurun_out/quick_tests.log 2>&1 || { tail -30 gpurun_out/quick_tests.log; exit 1; }
  tail -1 gpurun_out/quick_tests.log
  for wl in $WLS; do
    echo -n "$wl: "
    env $envs timeout -k 10 200 python3 bench.py --workload $wl --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,2),'us', d['value'], d['unit'], d['kernels_ms'])"
  done
done
