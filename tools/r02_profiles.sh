#!/bin/bash
# GPU box: rocprofv3 kernel stats + PMC passes (tools/profile_bench.sh) for the round-2 evidence set.
set -o pipefail
bash tools/profile_bench.sh nips_k128 --workload nips_k128_dense > gpurun_out/prof_nips_k128.log 2>&1 || echo "nips_k128 failed"
echo "nips_k128 done"
bash tools/profile_bench.sh dlmc_k512_bf16 --workload dlmc_k512_dense --mode bf16 > gpurun_out/prof_dlmc_k512_bf16.log 2>&1 || echo "dlmc failed"
echo "dlmc done"
bash tools/profile_bench.sh nips_k512 --workload nips_k512_dense > gpurun_out/prof_nips_k512.log 2>&1 || echo "nips_k512 failed"
echo "nips_k512 done"
bash tools/profile_bench.sh cop20k --workload cop20k_k128_hybrid > gpurun_out/prof_cop20k.log 2>&1 || echo "cop20k failed"
echo "cop20k done"
bash tools/profile_bench.sh cop20k_blocks --workload cop20k_blocks_k128_hybrid > gpurun_out/prof_cop20k_blocks.log 2>&1 || echo "cop20k_blocks failed"
echo "cop20k_blocks done"
du -sh gpurun_out/prof_*
