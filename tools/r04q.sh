export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4q
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tee gpurun_out/r4q/pytest.log | grep -E "passed|failed|Error|rror" | tail -4 &&
bash tools/probes/lab_gemm_set.sh r4q | awk '/^k/{print $1, $2, $11, $12, $13, $14, "bad", $(NF-4)}' &&
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4q/bench20.json 2> gpurun_out/r4q/bench.err &&
python bench.py --steps 200 --warmup 20 --no-cpu-baseline --workload dlmc_k512_dense --mode bf16 > gpurun_out/r4q/dlmc.json 2>> gpurun_out/r4q/bench.err &&
python - <<'P'
import json
for t in ("bench20","dlmc"):
    d=json.loads(open(f"gpurun_out/r4q/{t}.json").read().strip().splitlines()[-1])
    print(t, d["value"], d["ms_per_step"], d["kernels_ms"], d["step_breakdown_us"], d["dense_engine"]["chosen"], d["device_plan_ms"]["total_ms"])
P
