#!/bin/bash
# VALU/SALU/LDS instruction counts per wave and busy cycles per kernel for one bench workload, for the library in
# RTX_LIB (or the in-tree one).  usage (GPU box): bash tools/pmc_insts.sh WORKLOAD TAG [bench args]
W=$1; TAG=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/insts_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
B="$REPO/bench.py --workload $W --no-cpu --no-legs --steps 1 --warmup 0 --no-profile"
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES --output-format csv -d $OUT/p1 -o q -- python3 $B "$@" > /dev/null 2> $OUT/p1.err || { tail -5 $OUT/p1.err; exit 1; }
cd $REPO
python3 - $OUT <<'PY'
import collections, csv, glob, sys
sums = collections.defaultdict(lambda: collections.defaultdict(float))
for f in glob.glob(sys.argv[1] + "/p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0].split("::")[-1].split("<")[0]
        if k.startswith("k_"): sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in sorted(sums):
    s = sums[k]; w = max(s["SQ_WAVES"], 1)
    print(f"{k:22s} waves {w:12.0f} valu/wave {s['SQ_INSTS_VALU']/w:8.1f} salu/wave {s['SQ_INSTS_SALU']/w:8.1f} lds/wave {s['SQ_INSTS_LDS']/w:7.1f} "
          f"lanes {s['SQ_THREAD_CYCLES_VALU']/max(s['SQ_ACTIVE_INST_VALU'],1):5.1f} wave_cycles/wave {s['SQ_WAVE_CYCLES']/w:9.0f}")
PY
rm -rf $OUT/p1
