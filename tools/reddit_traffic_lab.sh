#!/bin/bash
# GPU box: reddit-like shard (configs[3] per-GPU share), dense kernel time + HBM-side traffic + L2 hit rate per knob set.
# usage: tools/reddit_traffic_lab.sh TAG "set1" "set2" ...     (a set = comma-separated NAME=VALUE plan knobs, "" = the rules)
set -o pipefail
TAG=$1; shift
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/traffic_$TAG
rm -rf $OUT && mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd $ROOT
WL=${WL:-reddit_shard_k256}
MODE=${MODE:-f16}
export BSMR_PACK_ON_DEVICE=${BSMR_PACK_ON_DEVICE:-0}
python3 tools/traffic_lab.py $WL $MODE "$@" > $OUT/times.jsonl 2> $OUT/times.err || { tail -5 $OUT/times.err; exit 1; }
echo "times done"
timeout -k 5 400 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 tools/traffic_lab.py --counted $WL $MODE "$@" > $OUT/pmc_fetch.log 2>&1 || { tail -5 $OUT/pmc_fetch.log; exit 1; }
echo "fetch done"
timeout -k 5 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $OUT/pmc_tcc -- python3 tools/traffic_lab.py --counted $WL $MODE "$@" > $OUT/pmc_tcc.log 2>&1 || { tail -5 $OUT/pmc_tcc.log; exit 1; }
python3 tools/traffic_lab.py --summarize $OUT/pmc_fetch $OUT/pmc_tcc -- "$@" > $OUT/counters.jsonl
find $OUT -name '*agent_info.csv' -delete
cat $OUT/times.jsonl $OUT/counters.jsonl
