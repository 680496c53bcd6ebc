#!/bin/bash
# Runs on the GPU box (via gpurun): rocprofv3 kernel stats + PMC passes of bench.py for one workload.
# usage: tools/profile_bench.sh <tag> [bench args...]      outputs under gpurun_out/prof_<tag>/
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
# the row clustering is not what is profiled here: keep its thousands of pass launches out of the traces
export BSMR_CLUSTER=host
# the un-profiled line first: it tunes (times every candidate engine) and writes what it chose; every profiler pass below
# REPLAYS that choice (bsmr_plan_set_tuned: nothing is timed), so all passes launch exactly the kernels the line names -
# under rocprofv3 the tuner's comparisons come out differently from pass to pass
python3 bench.py --steps 200 --warmup 20 --save-tune $OUT/tune.json "$@" > $OUT/bench.json 2> $OUT/bench.err || exit 1
REPLAY=""
[ -s $OUT/tune.json ] && REPLAY="--replay-tune $OUT/tune.json"
timeout -k 5 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -- python3 bench.py --steps 200 --warmup 20 --no-cpu-baseline $REPLAY "$@" > $OUT/stats.log 2>&1 || exit 1
timeout -k 5 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $REPLAY "$@" > $OUT/pmc_fetch.log 2>&1 || exit 1
timeout -k 5 200 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $REPLAY "$@" > $OUT/pmc_write.log 2>&1 || exit 1
timeout -k 5 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $OUT/pmc_tcc -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $REPLAY "$@" > $OUT/pmc_tcc.log 2>&1 || exit 1
timeout -k 5 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $OUT/pmc_sq -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $REPLAY "$@" > $OUT/pmc_sq.log 2>&1 || echo "sq pass failed"
timeout -k 5 120 rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ta -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline $REPLAY "$@" > $OUT/pmc_ta.log 2>&1 || echo "ta pass failed"
# keep the summaries, drop the per-launch traces (gpurun copies back at most 64 MiB)
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
du -sh $OUT
cat $OUT/bench.json
