#!/bin/bash
# The same small scenes walked from LDS (two-child tree staged by every block) and from global
# memory (four-child quantised tree): frame times side by side.  LDS residency is forced up to
# 48 KB of scene (what fits 64 KB of dynamic LDS with the stacks of a 128-thread block).
# usage (GPU box): bash tools/lds_vs_global.sh
set -e
for W in sample3_mis_cornell soup100 soup150 soup220 soup300 soup380; do
  for KB in 48 0; do
    RTX_LDS_SCENE_MAX_KB=$KB timeout -k 10 200 python bench.py --workload $W --no-cpu --no-legs > gpurun_out/lg_tmp.json 2>/dev/null
    python - "$W" "$KB" <<'PY'
import json, sys
d = json.load(open("gpurun_out/lg_tmp.json"))
print(f"{sys.argv[1]:22s} lds_max_kb {sys.argv[2]:>3s} bvh_width {d['config']['bvh_width']} {d['ms_per_step']:9.3f} ms  {d['stage_ms']}")
PY
  done
done
