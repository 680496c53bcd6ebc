#!/bin/bash
# Quick A/B on the GPU box: parity subset, then kernel times of the main workloads.
set -e
mkdir -p gpurun_out
timeout -k 10 400 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "small_random or other_k or nips or edge" > gpurun_out/quick_tests.log 2>&1 || { tail -30 gpurun_out/quick_tests.log; exit 1; }
tail -3 gpurun_out/quick_tests.log
for wl in nips_k128_dense nips_k32_hybrid nips_k512_dense dlmc_k512_dense cop20k_k128_hybrid; do
  for env in "" $EXTRA_ENVS; do
    echo "== $wl [$env]"
    env $env timeout -k 10 200 python3 bench.py --workload $wl --no-cpu-baseline --steps 300 --warmup 30 | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step']*1000,'us', d['value'], d['unit'], d['roofline'])"
  done
done
