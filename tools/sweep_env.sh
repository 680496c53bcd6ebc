#!/bin/bash
# GPU box: kernel times of workloads $WLS under each environment group in $EXTRA_ENVS (no parity run).
mkdir -p gpurun_out
for env in "" $EXTRA_ENVS; do
  envs=$(echo "$env" | tr ',' ' ')
  echo "#### [$envs]"
  for wl in $WLS; do
    echo -n "$wl: "
    env $envs timeout -k 10 200 python3 bench.py --workload $wl --no-cpu-baseline --steps 300 --warmup 30 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step']*1000,2),'us', d['kernels_ms'])"
  done
done
