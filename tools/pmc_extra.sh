#!/bin/bash
# extra PMC passes for latency diagnosis (GPU box)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_extra; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum" "SQ_INST_LEVEL_VMEM SQ_LEVEL_WAVES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_FLAT_READ_LDS_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed: $set"
done
python3 - <<'PY'
import csv, glob, collections, os
root=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_extra'
for f in sorted(glob.glob(root+'/p*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'denseGroups' in r['Kernel_Name'] or 'sparseEntries' in r['Kernel_Name']:
            agg[(r['Kernel_Name'][:28], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k[0], k[1], round(sum(v)/len(v),1))
PY
