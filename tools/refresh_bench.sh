#!/bin/bash
# Re-runs the bench lines kept under profiles/ (no profiler attached): the default line (which
# carries the MIS Cornell leg), the other profiled workloads and the other configurations.
# usage (on the GPU box): bash tools/refresh_bench.sh
set -e
O=gpurun_out/refresh; mkdir -p $O
timeout -k 10 300 python bench.py > $O/sample2_glass.json
timeout -k 10 200 python bench.py --workload sample3_mis_cornell > $O/sample3_mis_cornell.json
timeout -k 10 300 python bench.py --workload mesh_1m --steps 2 > $O/mesh_1m.json
timeout -k 10 200 python bench.py --workload sample1_diffuse > $O/sample1.json
timeout -k 10 300 python bench.py --workload sample3_mis_cornell --spp 1024 --steps 2 > $O/cornell_1024.json
timeout -k 10 200 python bench.py --workload sample3_mis_cornell_4lights > $O/cornell_4lights.json
timeout -k 10 300 python bench.py --workload mis_4k --spp 128 --steps 2 > $O/mis_4k.json
ls $O
