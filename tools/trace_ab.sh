#!/bin/bash
# Kernel-trace timelines (tools/timeline.py) of the frame-by-frame bench for the in-tree library and for tools/ab_old.so: which stage of a frame a change moved.  usage (GPU box): bash tools/trace_ab.sh
REPO=$(pwd)
cd /tmp && export TMPDIR=/tmp
for L in new old; do
  if [ $L = old ]; then export RTX_LIB=$REPO/tools/ab_old.so; fi
  rm -rf $REPO/gpurun_out/tr_$L
  timeout -k 10 200 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/tr_$L -o t -- python3 $REPO/bench.py --workload sample2_glass --no-cpu --no-legs --no-profile --sync --steps 6 --warmup 2 > $REPO/gpurun_out/tr_$L.json 2> $REPO/gpurun_out/tr_$L.err || exit 1
  F=$(find $REPO/gpurun_out/tr_$L -name '*kernel_trace.csv' | head -n 1)
  python3 $REPO/tools/timeline.py $F 3 > $REPO/gpurun_out/tr_$L.txt
  cp $F $REPO/gpurun_out/tr_$L.csv
  rm -rf $REPO/gpurun_out/tr_$L
done
