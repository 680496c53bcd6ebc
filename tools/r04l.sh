export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4l
BSMR_PLAN_TIMING=1 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4l/bench20.json 2> gpurun_out/r4l/bench.err &&
grep "\[plan\]" gpurun_out/r4l/bench.err | head -40 &&
python - <<'P'
import json
d=json.loads(open("gpurun_out/r4l/bench20.json").read().strip().splitlines()[-1])
print(d["ms_per_step"], d["device_plan_ms"], d["plan_build_s"], d["host_pipeline_ms"])
P
