#!/bin/bash
# the round's closing check on the GPU box: the whole -m gpu suite, smoke(), the driver's bench command, two more lines and
# a rehearsal of the N > 1 bench path on the one device
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/final
python -m pytest tests -x -q -m gpu 2>&1 | tee gpurun_out/final/pytest_all.log | grep -E "passed|failed|Error|rror" | tail -4 &&
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 &&
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench20.json 2> gpurun_out/final/bench.err &&
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload nips_k512_dense > gpurun_out/final/nips512.json 2>> gpurun_out/final/bench.err &&
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload dlmc_k512_dense --mode bf16 > gpurun_out/final/dlmc.json 2>> gpurun_out/final/bench.err &&
python bench.py --force-sharded --steps 5 --warmup 2 --graph-scale 0.125 > gpurun_out/final/sharded.json 2>> gpurun_out/final/bench.err &&
python - <<'P'
import json
for t in ("bench20","nips512","dlmc","sharded"):
    d=json.loads(open(f"gpurun_out/final/{t}.json").read().strip().splitlines()[-1])
    print(t, d["value"], d["ms_per_step"], d.get("kernels_ms"), d.get("step_breakdown_us"), (d.get("dense_engine") or {}).get("chosen"), (d.get("device_plan_ms") or {}).get("total_ms"), d.get("parity_mismatches_vs_cpu"), d.get("roofline", {}).get("frac"))
P
