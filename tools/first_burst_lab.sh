export PYTHONUNBUFFERED=1 LAB_FIRST_ONLY=1
mkdir -p gpurun_out/firstburst
(
for rep in 1 2 3; do
echo "as is:"; python tools/step_latency_lab.py 2>&1 | grep "first bursts"
echo "20 more steps first:"; LAB_PRE=20 python tools/step_latency_lab.py 2>&1 | grep "first bursts"
echo "200 more steps first:"; LAB_PRE=200 python tools/step_latency_lab.py 2>&1 | grep "first bursts"
echo "50 ms pause:"; LAB_SLEEP_MS=50 python tools/step_latency_lab.py 2>&1 | grep "first bursts"
done
) | tee gpurun_out/firstburst/out.txt
