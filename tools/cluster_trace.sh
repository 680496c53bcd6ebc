#!/bin/bash
# GPU box: kernel statistics of the device row clustering (rocprofv3 --kernel-trace --stats).
# usage: tools/cluster_trace.sh [reddit|myc15]
set -o pipefail
WHAT=${1:-reddit}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/cluster_trace_$WHAT
rm -rf $OUT && mkdir -p $OUT
cat > $OUT/run.py <<PY
import sys
sys.path.insert(0, "bsmr-sddmm_amd/python")
import bsmr_amd as eng, synth
rows, cols, ro, ci = synth.reddit_shard_like() if "$WHAT" == "reddit" else synth.mycielskian_pattern(15)
csr = eng.CSR.from_arrays(rows, cols, ro, ci)
bw = eng.host().bsmr_calculate_block_size(csr.handle, 200 << 30)
st, perm, clusters, stats = eng.cluster_rows_device(rows, cols, ro, ci, bw, 0.3)
print(st, clusters, stats)
PY
cd /tmp && export TMPDIR=/tmp && cd $ROOT
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 $OUT/run.py > $OUT/run.log 2>&1 || exit 1
F=$(find $OUT/stats -name '*kernel_stats.csv' | head -1)
cut -d, -f1-8 $F | cut -c1-200 | head -12
K=$(find $OUT/stats -name '*kernel_trace.csv' | head -1)
python3 tools/trace_steps.py $K | sed -n 1,30p
find $OUT -name '*kernel_trace.csv' -delete
tail -1 $OUT/run.log
