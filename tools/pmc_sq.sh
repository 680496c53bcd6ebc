#!/bin/bash
# SQ-only PMC passes (safe sets), each under its own timeout.  usage: tools/pmc_sq.sh [bench args]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/pmc_sq; rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_MFMA SQ_INSTS_BRANCH" "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH"; do
  i=$((i+1))
  timeout -k 5 150 rocprofv3 --pmc $set --output-format csv -d $OUT/p$i -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > $OUT/p$i.log 2>&1 || echo "pass $i failed"
done
python3 - <<'PY'
import csv, glob, collections, os
root=os.environ.get('GRAFT_REPO_ROOT','.')+'/gpurun_out/pmc_sq'
for f in sorted(glob.glob(root+'/p*/**/*counter_collection.csv', recursive=True)):
    agg=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'bsmr::dense' in r['Kernel_Name'] or 'sparseEntries' in r['Kernel_Name']:
            agg[(r['Kernel_Name'].split('<')[0][-14:], r['Counter_Name'])].append(float(r['Counter_Value']))
    for k,v in agg.items(): print(k[0], k[1], round(sum(v)/len(v),1))
PY
