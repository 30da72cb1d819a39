#!/bin/bash
# Lane utilisation of the VALU work per kernel: thread-cycles / (instruction-cycles x 64).
# usage: tools/pmc_lanes.sh WORKLOAD [extra bench args]
set -e
W=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/lanes_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d $OUT/raw -o q -- python3 $REPO/bench.py --workload $W --steps 1 --warmup 0 --no-cpu --no-profile "$@" > /dev/null 2> $OUT/err
cd $REPO
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + '/raw/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].split('::')[-1]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
with open(out + '/summary.txt', 'w') as o:
    for k, a in acc.items():
        if not k.startswith('k_'): continue
        w = max(a['SQ_WAVES'], 1)
        line = (f"{k:24s} VALU/wave {a['SQ_INSTS_VALU']/w:8.1f} LDS/wave {a['SQ_INSTS_LDS']/w:7.1f} SALU/wave {a['SQ_INSTS_SALU']/w:7.1f} "
                f"thread_cycles/active_inst {a['SQ_THREAD_CYCLES_VALU']/max(a['SQ_ACTIVE_INST_VALU'],1):7.2f} "
                f"thread_cycles/insts {a['SQ_THREAD_CYCLES_VALU']/max(a['SQ_INSTS_VALU'],1):7.2f}")
        print(line); o.write(line + '\n')
PY
rm -rf $OUT/raw
