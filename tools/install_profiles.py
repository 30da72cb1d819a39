"""Copies what tools/profile_round.sh left under gpurun_out/prof_<workload>/ into profiles/ (the
tracked summaries): once under the fixed names bench.py reads, once under names carrying the tag.
usage: python tools/install_profiles.py TAG [workload ...]"""
import os
import shutil
import sys

tag = sys.argv[1]
workloads = sys.argv[2:] or ["sample2_glass", "sample3_mis_cornell", "mesh_1m"]
for w in workloads:
    d = f"gpurun_out/prof_{w}"
    if not os.path.exists(d + "/kernel_stats.csv"):
        print("missing", d)
        continue
    for src, fixed, tagged in [
        ("kernel_stats.csv", f"kernel_stats_{w}.csv", f"{tag}_{w}_kernel_stats.csv"),
        ("bench_under_rocprof.json", None, f"{tag}_{w}_bench_under_rocprof.json"),
        (f"pmc_traffic_{w}.json", f"pmc_traffic_{w}.json", f"{tag}_pmc_traffic_{w}.json"),
        (f"sq_counters_{w}.json", f"sq_counters_{w}.json", f"{tag}_sq_counters_{w}.json"),
        (f"sq_counters_{w}.txt", None, f"{tag}_{w}_sq_counters.txt"),
    ]:
        if not os.path.exists(f"{d}/{src}"):
            print("missing", f"{d}/{src}")
            continue
        if fixed:
            shutil.copy(f"{d}/{src}", f"profiles/{fixed}")
        shutil.copy(f"{d}/{src}", f"profiles/{tag}_{src}" if tagged is None else f"profiles/{tagged}")
    print("installed", w)

# bench lines left by tools/refresh_bench.sh under gpurun_out/refresh/
import json
ref = "gpurun_out/refresh"
if os.path.isdir(ref):
    for w in workloads:
        if os.path.exists(f"{ref}/{w}.json"):
            shutil.copy(f"{ref}/{w}.json", f"profiles/{tag}_{w}_bench.json")
    others = {}
    for key, name in [("configs[0] sample1 512x512 64 spp", "sample1"),
                      ("configs[2] sample3 MIS 1024x1024 1024 spp on one GPU", "cornell_1024"),
                      ("sample3 MIS, four quad lights, 256 spp", "cornell_4lights"),
                      ("configs[4] 4K MIS, 128 of 4096 spp on one GPU", "mis_4k")]:
        if os.path.exists(f"{ref}/{name}.json"):
            d = json.load(open(f"{ref}/{name}.json"))
            others[key] = {k: d[k] for k in ("value", "unit", "ms_per_step", "mpaths_per_s", "mshadow_rays_per_s",
                                             "bounces_per_path", "config", "stage_ms", "roofline", "cpu_baseline")
                           if k in d}
    if others:
        json.dump(others, open(f"profiles/{tag}_other_configs.json", "w"), indent=1)
        print("installed bench lines")
