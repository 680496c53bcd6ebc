export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4o
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "shard" 2>&1 | tee gpurun_out/r4o/pytest_shard.log | grep -E "configs\[3\]|passed|failed|Error|rror" | cut -c1-600 | tail -8
