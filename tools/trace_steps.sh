#!/bin/bash
# GPU box: rocprofv3 kernel trace of bench.py for one workload, reduced to a timeline summary (tools/trace_steps.py).
# usage: tools/trace_steps.sh <tag> [bench args...]     output gpurun_out/trace_<tag>.txt
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/trace_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
export BSMR_CLUSTER=host
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/kt -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline "$@" > $OUT/bench.log 2>&1 || exit 1
F=$(find $OUT/kt -name '*kernel_trace.csv' | head -1)
python3 tools/trace_steps.py $F --dump 40 > $ROOT/gpurun_out/trace_$TAG.txt
rm -rf $OUT/kt
cat $ROOT/gpurun_out/trace_$TAG.txt
