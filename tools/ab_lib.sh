#!/bin/bash
# A/B of two builds of the library on the GPU box: runs the bench workloads given once per library,
# alternating, and prints one line per run.
# usage: bash tools/ab_lib.sh OLD.so "WORKLOAD [bench args]" ["WORKLOAD2 ..."]
OLD=$1; shift
for W in "$@"; do
  for rep in 1 2 3; do
    for L in new old; do
      ( if [ $L = old ]; then export RTX_LIB=$OLD; fi
        timeout -k 10 200 python bench.py --workload $W --no-cpu --no-legs > gpurun_out/ab_tmp.json 2>gpurun_out/ab_err.txt ) || { echo "$L failed"; tail -3 gpurun_out/ab_err.txt; exit 1; }
      python - "$L $W" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(f"{sys.argv[1]:44s} {d['ms_per_step']:9.3f} ms  {d['stage_ms']}")
PY
    done
  done
done
