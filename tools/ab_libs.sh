#!/bin/bash
# Runs one bench workload once per library given ("-" = the tree's own build) and prints one line per run.
# usage: bash tools/ab_libs.sh "WORKLOAD [bench args]" - raytracer_amd/librtx_hip_x.so ...
W=$1; shift
for rep in 1 2; do
for L in "$@"; do
  ( if [ "$L" != "-" ]; then export RTX_LIB=$L; fi
    timeout -k 10 300 python bench.py --workload $W --no-cpu --no-legs > gpurun_out/ab_tmp.json 2>gpurun_out/ab_err.txt ) || { echo "$L failed"; tail -3 gpurun_out/ab_err.txt; exit 1; }
  python - "$L" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(f"{sys.argv[1]:44s} {d['ms_per_step']:9.3f} ms  {d['stage_ms']}")
PY
done
done
