#!/bin/bash
# Collects the rocprofv3 evidence for one bench workload on the GPU box, every pass a run of its own:
#   kernel trace + stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; two SQ counter passes.
# Leaves under gpurun_out/prof_TAG: kernel_stats.csv, bench_under_rocprof.json,
# pmc_traffic_WORKLOAD.json, sq_counters_WORKLOAD.json/.txt
# usage: tools/profile_round.sh WORKLOAD TAG [extra bench args]
set -e
W=$1; TAG=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$REPO/bench.py --workload $W --no-cpu --no-legs"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o s -- python3 $B --steps 2 --warmup 1 "$@" > $OUT/bench_under_rocprof.json 2> $OUT/stats.err
timeout -s KILL 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o f -- python3 $B --steps 1 --warmup 0 --no-profile "$@" > /dev/null 2> $OUT/fetch.err
timeout -s KILL 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o w -- python3 $B --steps 1 --warmup 0 --no-profile "$@" > /dev/null 2> $OUT/write.err
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM --output-format csv -d $OUT/sq1 -o q -- python3 $B --steps 1 --warmup 0 --no-profile "$@" > /dev/null 2> $OUT/sq1.err
timeout -s KILL 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_INSTS_SMEM --output-format csv -d $OUT/sq2 -o q -- python3 $B --steps 1 --warmup 0 --no-profile "$@" > /dev/null 2> $OUT/sq2.err
cd $REPO
F=$(find $OUT/fetch -name '*counter_collection.csv' | head -n 1)
Wc=$(find $OUT/write -name '*counter_collection.csv' | head -n 1)
python3 tools/pmc_traffic.py $F $Wc $W $OUT/pmc_traffic_$W.json > /dev/null
Q1=$(find $OUT/sq1 -name '*counter_collection.csv' | head -n 1)
Q2=$(find $OUT/sq2 -name '*counter_collection.csv' | head -n 1)
python3 tools/pmc_sq.py $W $OUT/sq_counters_$W.json $OUT/sq_counters_$W.txt $Q1 $Q2 > /dev/null
S=$(find $OUT/stats -name '*kernel_stats.csv' | head -n 1)
cp $S $OUT/kernel_stats.csv
# what the counters were collected from: the hash bench.py compares with the tree (profiles_stale)
python3 -c "import bench; print(bench.csrc_sha256())" > $OUT/csrc_sha256.txt
# the raw counter CSVs are large; keep only the summaries
rm -rf $OUT/fetch $OUT/write $OUT/stats $OUT/sq1 $OUT/sq2
ls -la $OUT
