#!/bin/bash
# Collects the rocprofv3 evidence for one bench workload on the GPU box:
#   kernel trace + stats, then one counter pass each for FETCH_SIZE and WRITE_SIZE.
# usage: tools/profile_round.sh WORKLOAD TAG [extra bench args]
set -e
W=$1; TAG=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $REPO/bench.py --workload $W --steps 2 --warmup 1 --no-cpu "$@" > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $REPO/bench.py --workload $W --steps 1 --warmup 0 --no-cpu --no-profile "$@" > /dev/null 2> $OUT/fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $REPO/bench.py --workload $W --steps 1 --warmup 0 --no-cpu --no-profile "$@" > /dev/null 2> $OUT/write.err
cd $REPO
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -n 1)
Wc=$(find $OUT/write -name '*counter_collection.csv' | head -n 1)
python3 tools/pmc_traffic.py $F $Wc $W $OUT/pmc_traffic_$W.json > /dev/null
S=$(find $OUT/stats -name '*kernel_stats.csv' | head -n 1)
cp $S $OUT/kernel_stats.csv
# the raw counter CSVs are large; keep only the summaries
rm -rf $OUT/fetch $OUT/write $OUT/stats
ls -la $OUT
