#!/bin/bash
# One SQ counter pass over a bench workload; prints per-kernel means.
# usage: tools/pmc_sq.sh WORKLOAD [extra bench args]
set -e
W=$1; shift
REPO=$(pwd)
OUT=$REPO/gpurun_out/sq_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_INSTS_VMEM --output-format csv -d $OUT/raw -o q -- python3 $REPO/bench.py --workload $W --steps 1 --warmup 0 --no-cpu --no-profile "$@" > /dev/null 2> $OUT/err
cd $REPO
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + '/raw/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].split('::')[-1]
    acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Counter_Name'] == 'SQ_WAVES': n[k] += 1
with open(out + '/summary.txt', 'w') as o:
    for k in acc:
        if not k.startswith('k_'): continue
        a = acc[k]; w = max(a['SQ_WAVES'], 1)
        line = (f"{k:24s} launches {n[k]:4d} waves/launch {w/max(n[k],1):9.0f} VALU/wave {a['SQ_INSTS_VALU']/w:8.1f} SALU/wave {a['SQ_INSTS_SALU']/w:7.1f} "
                f"VMEM/wave {a['SQ_INSTS_VMEM']/w:6.1f} wave_cycles/wave {a['SQ_WAVE_CYCLES']/w:9.0f} valu_active/wave_cycles {a['SQ_ACTIVE_INST_VALU']/max(a['SQ_WAVE_CYCLES'],1):.3f} "
                f"wait_any/wave_cycles {a['SQ_WAIT_INST_ANY']/max(a['SQ_WAVE_CYCLES'],1):.3f} busy_cycles/launch {a['SQ_BUSY_CYCLES']/max(n[k],1):10.0f}")
        print(line); o.write(line + '\n')
PY
rm -rf $OUT/raw
