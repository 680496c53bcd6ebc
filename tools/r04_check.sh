#!/bin/bash
# Runs on the GPU box (via gpurun): round-4 checks - new GPU tests, then the bench lines the verdict asks about.
# usage: tools/r04_check.sh <tag> [tests|bench|all]
TAG=${1:-r4a}; WHAT=${2:-all}
OUT=gpurun_out/$TAG
mkdir -p $OUT
if [ "$WHAT" = tests ] || [ "$WHAT" = all ]; then
  python -m pytest tests/test_gpu_gemm.py tests/test_gpu_cluster.py -x -q -m gpu -s > $OUT/pytest_gemm.log 2>&1; echo "rc=$?" >> $OUT/pytest_gemm.log
  tail -25 $OUT/pytest_gemm.log
  python -m pytest tests/test_gpu_parity.py -x -q -m gpu -s -k "tuned_plans or full_size_parity" > $OUT/pytest_tuned.log 2>&1; echo "rc=$?" >> $OUT/pytest_tuned.log
  grep -v "^$" $OUT/pytest_tuned.log | tail -12
fi
if [ "$WHAT" = bench ] || [ "$WHAT" = all ]; then
  python bench.py --steps 20 --warmup 5 > $OUT/bench20.json 2> $OUT/bench.err
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline > $OUT/bench200.json 2>> $OUT/bench.err
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload dlmc_k512_dense --mode bf16 > $OUT/dlmc.json 2>> $OUT/bench.err
  python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload nips_k512_dense > $OUT/nips512.json 2>> $OUT/bench.err
  tail -3 $OUT/bench.err
  python - <<PY
import json
for f in ("bench20","bench200","dlmc","nips512"):
    try:
        d=json.loads(open("$OUT/%s.json" % f).read().strip().splitlines()[-1])
        e=d["dense_engine"]
        print(f, d["value"], d["ms_per_step"], d["kernels_ms"], d["step_breakdown_us"], e.get("chosen"), e.get("group"), e.get("blocks_per_item"), {k:v for k,v in e.items() if k.endswith("_us") and isinstance(v,(int,float)) and v>0}, d.get("parity_mismatches_vs_cpu"))
    except Exception as ex: print(f, "ERR", ex)
PY
fi
