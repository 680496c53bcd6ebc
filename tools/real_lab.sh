#!/bin/bash
# GPU box: parity on the real SuiteSparse patterns, then their bench lines.
set -e
mkdir -p gpurun_out
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "suitesparse" > gpurun_out/real_tests.log 2>&1 || { tail -40 gpurun_out/real_tests.log; exit 1; }
tail -3 gpurun_out/real_tests.log
for wl in mycielskian15_k128 mycielskian15_k32 mycielskian15_k512 mycielskian14_k128 trefethen20000_k128 wathen100_k128; do
  echo "== $wl"
  timeout -k 10 300 python3 bench.py --workload $wl --steps 200 --warmup 20 > gpurun_out/bench_$wl.json 2> gpurun_out/bench_$wl.err || { tail -5 gpurun_out/bench_$wl.err; exit 1; }
  python3 -c "
import json,sys
d=json.loads(open('gpurun_out/bench_$wl.json').read().strip().splitlines()[-1])
print(d['ms_per_step']*1000,'us', d['value'], d['unit'], 'vs_baseline', d['vs_baseline'], d['kernels_ms'], d['host_pipeline_ms'], 'cpu', d['cpu_baseline']['value'], 'mismatch', d['parity_mismatches_vs_cpu'])
print('  ', d['roofline'])"
done
