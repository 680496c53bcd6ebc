export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4i
python -m pytest tests/test_gpu_gemm.py tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tee gpurun_out/r4i/pytest.log | tail -4 &&
python tools/step_latency_lab.py 2>&1 | tee gpurun_out/r4i/latency.txt &&
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r4i/bench20.json 2> gpurun_out/r4i/bench.err &&
python bench.py --steps 200 --warmup 20 --no-cpu-baseline > gpurun_out/r4i/bench200.json 2>> gpurun_out/r4i/bench.err &&
python - <<'P'
import json
for t in ("bench20","bench200"):
    d=json.loads(open(f"gpurun_out/r4i/{t}.json").read().strip().splitlines()[-1])
    print(t, d["value"], d["ms_per_step"], d["kernels_ms"], d["step_breakdown_us"], d["dense_engine"]["chosen"], d["dense_engine"]["gemm_fp32_call_us"], d["dense_engine"]["gemm_us"])
P
