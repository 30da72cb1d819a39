#!/bin/bash
# One counter pass over a bench workload; prints per-kernel sums per wave.
# usage: tools/pmc_generic.sh "COUNTER COUNTER ..." WORKLOAD [extra bench args]
set -e
CTRS=$1; W=$2; shift 2
REPO=$(pwd)
OUT=$REPO/gpurun_out/pmc_$W
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout -s KILL 300 rocprofv3 --pmc $CTRS --output-format csv -d $OUT/raw -o q -- python3 $REPO/bench.py --workload $W --steps 1 --warmup 0 --no-cpu --no-legs --no-profile "$@" > /dev/null 2> $OUT/err
cd $REPO
python3 - $OUT <<'PY'
import csv, sys, glob, collections
out = sys.argv[1]
f = glob.glob(out + '/raw/**/*counter_collection.csv', recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
names = []
for r in csv.DictReader(open(f)):
    k = r['Kernel_Name'].split('(')[0].split('::')[-1]
    c = r['Counter_Name']
    if c not in names: names.append(c)
    acc[k][c] += float(r['Counter_Value'])
    if c == names[0]: n[k] += 1
for k, a in acc.items():
    if not k.startswith('k_'): continue
    print(f"{k:22s} launches {n[k]:4d} " + " ".join(f"{c}={a[c]/max(n[k],1):.4g}" for c in names))
PY
rm -rf $OUT/raw
