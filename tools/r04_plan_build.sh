#!/bin/bash
# plan build times of tunable and untuned plans -> gpurun_out/plan_build/r04_plan_build.txt (copied to profiles/r04_plan_build.md)
export PYTHONUNBUFFERED=1
mkdir -p gpurun_out/plan_build
(
echo "## BSMR_DENSE_ENGINE=tuned (the plans bench.py builds)"
BSMR_DENSE_ENGINE=tuned python tools/plan_build_lab.py nips_k128_dense mycielskian15_k128 cop20k_k128_hybrid dlmc_k512_dense reddit_shard_k256 || exit 1
echo "## rules (untuned plans)"
python tools/plan_build_lab.py nips_k128_dense mycielskian15_k128 cop20k_k128_hybrid dlmc_k512_dense reddit_shard_k256
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/plan_build/r04_plan_build.txt
