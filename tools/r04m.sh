export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4m
python -m pytest tests -x -q -m gpu 2>&1 | tee gpurun_out/r4m/pytest_all.log | grep -E "passed|failed|Error|error" | tail -5 &&
bash tools/r04_plan_build.sh > /dev/null && cat gpurun_out/plan_build/r04_plan_build.txt | cut -c1-230
