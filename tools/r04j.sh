export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4j
python tools/step_latency_lab.py 2>&1 | tee gpurun_out/r4j/latency.txt
