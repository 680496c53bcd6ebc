#!/bin/bash
# GPU box: the whole -m gpu suite, output kept under gpurun_out/.
mkdir -p gpurun_out
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1
rc=$?
tail -25 gpurun_out/gpu_tests.log
exit $rc
