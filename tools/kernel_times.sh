#!/bin/bash
# rocprofv3 --kernel-trace --stats over one bench command; prints calls, average and total per kernel.
# usage (GPU box): bash tools/kernel_times.sh TAG <bench.py arguments>
TAG=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/kt_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o s -- python3 $REPO/bench.py --no-cpu --no-legs --no-profile "$@" > /dev/null 2> $OUT/err
cd $REPO
python3 - $OUT <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True)[0]
for r in csv.DictReader(open(f)):
    name = r["Name"].split("(")[0].replace("void ", "").replace("rtx::", "")
    print(f"{name:28s} calls {int(r['Calls']):5d} avg {float(r['AverageNs'])/1e3:10.1f} us total {float(r['TotalDurationNs'])/1e6:9.2f} ms")
PY
rm -rf $OUT
