#!/bin/bash
# GPU box: rocprofv3 kernel stats + PMC passes at the round-2 HEAD for the workloads whose traffic entries were still
# round-1 passes (mycielskian15, reddit-like shard) and the default workload.
set -o pipefail
bash tools/profile_bench.sh nips_k128 --workload nips_k128_dense > gpurun_out/prof_nips_k128.log 2>&1 || echo "nips_k128 failed"
echo "nips_k128 done"
bash tools/profile_bench.sh myc15_k128 --workload mycielskian15_k128 > gpurun_out/prof_myc15_k128.log 2>&1 || echo "myc15 failed"
echo "myc15 done"
bash tools/profile_bench.sh reddit_shard --workload reddit_shard_k256 > gpurun_out/prof_reddit_shard.log 2>&1 || echo "reddit failed"
echo "reddit done"
du -sh gpurun_out/prof_*
