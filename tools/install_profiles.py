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
