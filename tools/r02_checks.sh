#!/bin/bash
# GPU box: round-2 checks - sharded operator test, strong-scaling rehearsal on one rank, hybrid overlap timings.
mkdir -p gpurun_out/r02c
timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sharded_operator or two_plans or options_struct" > gpurun_out/r02c/tests.log 2>&1; tail -3 gpurun_out/r02c/tests.log
echo "== strong-scaling rehearsal, 1 rank, 1/8 of the reddit-like graph"
timeout -k 10 400 python3 bench.py --gpus 1 --force-sharded --graph-scale 0.125 --steps 50 --warmup 5 2> gpurun_out/r02c/sharded.err | tail -1 | tee gpurun_out/r02c/sharded.json | cut -c1-900
tail -3 gpurun_out/r02c/sharded.err
for wl in cop20k_blocks_k128_hybrid cop20k_k128_hybrid; do
  for ov in -1 0; do
    echo -n "$wl overlap=$ov: "
    BSMR_OVERLAP_STREAMS=$ov timeout -k 10 300 python3 bench.py --workload $wl --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,2),'us', d['kernels_ms'], 'dense_nnz', d['config']['dense_nnz'], 'sparse_nnz', d['config']['sparse_nnz'])"
  done
done
for ov in -1 0; do
  echo -n "dlmc_k512_d01 bf16 RPHM as is (no promotion) overlap=$ov: "
  BSMR_PROMOTE_AVERAGE=0 BSMR_OVERLAP_STREAMS=$ov timeout -k 10 300 python3 bench.py --workload dlmc_k512_d01 --mode bf16 --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,2),'us', d['kernels_ms'], 'dense_nnz', d['config']['dense_nnz'], 'sparse_nnz', d['config']['sparse_nnz'])"
done
echo -n "nips_k128 host clustering: "; BSMR_CLUSTER=host timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,2),'us', d['plan_build_s'], d['host_pipeline_ms'])"
echo -n "nips_k128 device clustering: "; timeout -k 10 200 python3 bench.py --no-cpu-baseline --steps 100 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,2),'us', d['plan_build_s'], d['host_pipeline_ms'])"
