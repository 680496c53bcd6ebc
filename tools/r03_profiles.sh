#!/bin/bash
# GPU box: the round-3 evidence set, one tag per argument (a gpurun call fits three or four of them):
#   nips_k128            the headline workload, tuned (BENCH line, kernel statistics, counters)
#   dlmc_{dense,d01,d03,d05,sparse}   BASELINE configs[4] per delta in bf16 with the RPHM's split taken as is
#   dlmc_rules           ... with the plan's own rules (every delta runs this all-dense plan), tuned
#   reddit_shard         BASELINE configs[3], the share of one of eight GPUs
#   nips_k512, myc15_k128   further workloads of profiles/r03_results.md
set -o pipefail
for tag in "$@"; do
  unset BSMR_PROMOTE_AVERAGE BSMR_FOLD_DENSE_BELOW
  case $tag in
    nips_k128) args="--workload nips_k128_dense" ;;
    nips_k512) args="--workload nips_k512_dense" ;;
    myc15_k128) args="--workload mycielskian15_k128" ;;
    reddit_shard) args="--workload reddit_shard_k256" ;;
    dlmc_rules) args="--workload dlmc_k512_dense --mode bf16" ;;
    dlmc_dense|dlmc_d01|dlmc_d03|dlmc_d05|dlmc_sparse)
      export BSMR_PROMOTE_AVERAGE=0 BSMR_FOLD_DENSE_BELOW=0
      args="--workload dlmc_k512_${tag#dlmc_} --mode bf16" ;;
    *) echo "unknown tag $tag"; exit 2 ;;
  esac
  rm -rf gpurun_out/prof_$tag
  bash tools/profile_bench.sh $tag $args > gpurun_out/prof_$tag.log 2>&1 || { echo "$tag failed"; tail -5 gpurun_out/prof_$tag.log; }
  echo "$tag done: $(tail -1 gpurun_out/prof_$tag.log | cut -c1-200)"
done
