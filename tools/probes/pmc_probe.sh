#!/bin/bash
# Runs on the GPU box: L2 / fabric counters of the sweep probe.  usage: tools/probes/pmc_probe.sh <tag> <probe args...>
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; TAG=$1; shift
OUT=$ROOT/gpurun_out/pmc_probe_$TAG; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
i=0
for set in "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES"; do
  i=$((i+1))
  timeout -k 5 120 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- $ROOT/tools/probes/sweep_probe "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections, os
root=os.environ['OUT'] if 'OUT' in os.environ else None
PY
python3 - "$OUT" <<'PY'
import csv, glob, collections, sys
root=sys.argv[1]
for f in sorted(glob.glob(root+'/p*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'denseSweep' in r['Kernel_Name']:
            agg[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k, 'mean', round(sum(v)/len(v),1), 'n', len(v))
PY
