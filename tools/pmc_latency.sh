#!/bin/bash
# rocprofv3's derived occupancy and latency metrics for one bench workload, one pass per pair:
# MeanOccupancyPerCU (waves), VmemLatency, SmemLatency, InstrFetchLatency, LdsLatency (cycles), per kernel.
# usage (GPU box): bash tools/pmc_latency.sh WORKLOAD TAG [bench args]
W=$1; TAG=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/lat_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$REPO/bench.py --workload $W --no-cpu --no-legs --steps 1 --warmup 0 --no-profile"
i=0
for P in "MeanOccupancyPerCU VmemLatency" "SmemLatency InstrFetchLatency" "LdsLatency VALUBusy"; do
  i=$((i+1))
  timeout -s KILL 300 rocprofv3 --pmc $P --output-format csv -d $OUT/p$i -o q -- python3 $B "$@" > /dev/null 2> $OUT/p$i.err || { tail -5 $OUT/p$i.err; exit 1; }
done
cd $REPO
python3 - $OUT <<'PY'
import collections, csv, glob, sys
sums = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.defaultdict(lambda: collections.defaultdict(int))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        if k.startswith("k_"):
            sums[k][r["Counter_Name"]] += float(r["Counter_Value"]); n[k][r["Counter_Name"]] += 1
for k in sorted(sums):
    print(f"{k:22s} launches {max(n[k].values()):4d} " + " ".join(f"{c} {sums[k][c] / n[k][c]:9.1f}" for c in sorted(sums[k])))
PY
rm -rf $OUT/p1 $OUT/p2 $OUT/p3
