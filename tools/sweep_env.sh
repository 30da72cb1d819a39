#!/bin/bash
# One environment variable swept over values and workloads on the GPU box; prints frame and stage times.
# usage: bash tools/sweep_env.sh VAR "v1 v2 ..." "W1 args" "W2 args" ...
VAR=$1; VALS=$2; shift 2
for W in "$@"; do
  for V in $VALS; do
    export $VAR=$V
    timeout -k 10 200 python bench.py --workload $W --no-cpu --no-legs > gpurun_out/ab_tmp.json 2>gpurun_out/ab_err.txt || { tail -3 gpurun_out/ab_err.txt; exit 1; }
    python - "$W" $VAR=$V <<'PY'
import json, sys
d = json.load(open("gpurun_out/ab_tmp.json"))
st = d['stage_ms']
print(f"{sys.argv[1]:22s} {sys.argv[2]:28s} {d['ms_per_step']:9.3f} ms  intersect {st.get('intersect')} shadow {st.get('shadow')} shade {st.get('shade')}")
PY
  done
done
