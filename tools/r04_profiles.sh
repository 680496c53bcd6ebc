#!/bin/bash
# GPU box: the round-4 evidence set, one tag per argument (a gpurun call fits three or four of them).  Every profiler pass
# replays the tuned choice of the un-profiled line (tools/profile_bench.sh).
#   nips_k128      the headline workload (BASELINE configs[1]), tuned
#   dlmc_rules     BASELINE configs[4] (4096^2, K = 512, bf16) with the plan's own rules (every delta runs this all-dense plan), tuned
#   dlmc_d01 ...   ... per delta with the RPHM's split taken as is
#   nips_k512, myc15_k128, reddit_shard   further workloads
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
