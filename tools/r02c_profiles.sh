#!/bin/bash
# GPU box: kernel statistics + counters for the workloads whose tuned dense engine is not the rules' engine.
set -o pipefail
rm -rf gpurun_out/prof_dlmc_tuned gpurun_out/prof_myc15_k512_tuned
bash tools/profile_bench.sh dlmc_tuned --workload dlmc_k512_dense --mode bf16 > gpurun_out/prof_dlmc_tuned.log 2>&1 || echo "dlmc failed"
echo "dlmc done"
bash tools/profile_bench.sh myc15_k512_tuned --workload mycielskian15_k512 > gpurun_out/prof_myc15_k512_tuned.log 2>&1 || echo "myc15 failed"
echo "myc15 done"
