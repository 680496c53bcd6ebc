export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/r4p
python -m pytest tests/test_gpu_pack.py tests/test_gpu_parity.py tests/test_gpu_gemm.py tests/test_gpu_sweep.py -x -q -m gpu 2>&1 | tee gpurun_out/r4p/pytest.log | grep -E "passed|failed|Error|rror|^K=" | tail -8 &&
BSMR_DENSE_ENGINE=tuned python tools/plan_build_lab.py dlmc_k512_dense nips_k128_dense 2>&1 | grep -v amdgpu
