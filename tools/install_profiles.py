"""Copies what tools/profile_round.sh left under gpurun_out/prof_<workload>/ into profiles/ (the
tracked summaries) under the fixed names bench.py reads; the files that have no fixed name (the bench line
printed under rocprofv3, the counters as text) get the tag in their name.  profiles/MANIFEST.json records,
per workload, the hash of the kernel sources the counters were collected from (bench.py's profiles_stale)
and the tag.
usage: python tools/install_profiles.py TAG [workload ...]"""
import json
import os
import shutil
import sys

tag = sys.argv[1]
workloads = sys.argv[2:] or ["sample2_glass", "sample3_mis_cornell", "mesh_1m", "sample1_diffuse",
                             "sample3_mis_cornell_4lights", "mis_4k"]
manifest_path = "profiles/MANIFEST.json"
manifest = json.load(open(manifest_path)) if os.path.exists(manifest_path) else {}
for w in workloads:
    d = f"gpurun_out/prof_{w}"
    if not os.path.exists(d + "/kernel_stats.csv"):
        print("missing", d)
        continue
    for src, dst in [
        ("kernel_stats.csv", f"kernel_stats_{w}.csv"),
        ("bench_under_rocprof.json", f"{tag}_{w}_bench_under_rocprof.json"),
        (f"pmc_traffic_{w}.json", f"pmc_traffic_{w}.json"),
        (f"sq_counters_{w}.json", f"sq_counters_{w}.json"),
        (f"sq_counters_{w}.txt", f"sq_counters_{w}.txt"),
    ]:
        if not os.path.exists(f"{d}/{src}"):
            print("missing", f"{d}/{src}")
            continue
        shutil.copy(f"{d}/{src}", f"profiles/{dst}")
    sha = open(f"{d}/csrc_sha256.txt").read().strip() if os.path.exists(f"{d}/csrc_sha256.txt") else None
    manifest[w] = {"csrc_sha256": sha, "collected_in": tag,
                   "files": [f"kernel_stats_{w}.csv", f"pmc_traffic_{w}.json", f"sq_counters_{w}.json"]}
    print("installed", w)
json.dump(manifest, open(manifest_path, "w"), indent=1, sort_keys=True)
