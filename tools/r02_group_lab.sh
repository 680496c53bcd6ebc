#!/bin/bash
# GPU box: the streaming engine with 2 panels per group and fewer blocks per wave.
for wl in nips_k128_dense:f16 mycielskian15_k128:f16 mycielskian15_k32:f16 reddit_shard_k256:f16 mycielskian14_k128:f16; do
  w=${wl%%:*}; m=${wl##*:}
  for env in "" "BSMR_DENSE_GROUP=2,BSMR_DENSE_BLOCKS_PER_WG=2" "BSMR_DENSE_GROUP=2,BSMR_DENSE_BLOCKS_PER_WG=3" "BSMR_DENSE_GROUP=2,BSMR_DENSE_BLOCKS_PER_WG=4" "BSMR_DENSE_GROUP=2,BSMR_DENSE_BLOCKS_PER_WG=6" "BSMR_DENSE_GROUP=1,BSMR_DENSE_BLOCKS_PER_WG=4" "BSMR_DENSE_GROUP=1,BSMR_DENSE_BLOCKS_PER_WG=6"; do
    echo -n "$w [$env]: "
    env $(echo $env | tr ',' ' ') timeout -k 10 300 python3 bench.py --workload $w --mode $m --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,2),'us', d['kernels_ms']['dense_ms'], 'H', d['config']['group_size'], 'ucols', d['config']['union_columns'])"
  done
done
