export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4k
python -m pytest tests/test_gpu_pack.py -x -q -m gpu -s 2>&1 | tee gpurun_out/r4k/pytest_pack.log | grep -E "listed entries|passed|failed|Error|error|assert" | tail -20 &&
BSMR_PLAN_TIMING=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4k/bench20.json 2> gpurun_out/r4k/bench.err &&
grep "\[plan\]" gpurun_out/r4k/bench.err | head -40 &&
python - <<'P'
import json
d=json.loads(open("gpurun_out/r4k/bench20.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["device_plan_ms"], d["plan_build_s"], d["host_pipeline_ms"], d["parity_mismatches_vs_cpu"] if "parity_mismatches_vs_cpu" in d else None)
P
