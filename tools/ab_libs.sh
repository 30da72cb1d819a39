#!/bin/bash
# A/B of library builds on the GPU box: runs one bench workload with each library given
# (RTX_LIB selects the build; "default" = raytracer_amd/librtx_hip.so) and prints one line per run.
# usage: bash tools/ab_libs.sh "WORKLOAD [bench args]" default ab_B.so ab_C.so ...
set -e
W=$1; shift
for L in "$@"; do
  if [ "$L" = default ]; then unset RTX_LIB; else export RTX_LIB=$PWD/$L; fi
  for rep in 1 2; do
  timeout -k 10 200 python bench.py --workload $W --no-cpu --no-mis > gpurun_out/ab_tmp.json 2>/dev/null
  python - "$L" <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
print(f"{sys.argv[1]:12s} {d['ms_per_step']:9.3f} ms  {d['stage_ms']}")
PY
  done
done
