#!/bin/bash
# A/B of environment settings on the GPU box: runs one bench workload once per setting given
# ("-" = nothing set; otherwise VAR=value[,VAR=value...]) and prints one line per run.
# usage: bash tools/ab_env.sh "WORKLOAD [bench args]" - RTX_X=1 RTX_X=1,RTX_Y=2 ...
W=$1; shift
for E in "$@"; do
  for rep in 1 2; do
  ( if [ "$E" != "-" ]; then IFS=, ; for kv in $E; do export "$kv"; done; unset IFS; fi
    timeout -k 10 200 python bench.py --workload $W --no-cpu --no-legs > gpurun_out/ab_tmp.json 2>gpurun_out/ab_err.txt ) || { echo "$E failed"; tail -3 gpurun_out/ab_err.txt; exit 1; }
  python - "$E" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(f"{sys.argv[1]:36s} {d['ms_per_step']:9.3f} ms  {d['stage_ms']}")
PY
  done
done
