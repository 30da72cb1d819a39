"""Copies what tools/profile_round.sh and tools/refresh_bench.sh left under gpurun_out/ into profiles/
(the tracked summaries) and prints one line per bench.  usage: python tools/install_profiles.py [round tag]"""
import csv, json, os, shutil, sys
tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
for w in ["sample2_glass", "sample3_mis_cornell", "mesh_1m"]:
    d = f"gpurun_out/prof_{w}"
    if os.path.exists(d + "/kernel_stats.csv"):
        shutil.copy(d + "/kernel_stats.csv", f"profiles/{tag}_{w}_kernel_stats.csv")
        shutil.copy(d + "/bench_under_rocprof.json", f"profiles/{tag}_{w}_bench_under_rocprof.json")
        shutil.copy(d + f"/pmc_traffic_{w}.json", f"profiles/pmc_traffic_{w}.json")
        shutil.copy(d + f"/pmc_traffic_{w}.json", f"profiles/{tag}_pmc_traffic_{w}.json")
R = "gpurun_out/refresh/"
for w in ["sample2_glass", "sample3_mis_cornell", "mesh_1m"]:
    shutil.copy(R + w + ".json", f"profiles/{tag}_{w}_bench.json")
other = f"profiles/{tag}_other_configs.json"
old = json.load(open(other))
names = {"configs[0] sample1 512x512 64 spp": "sample1", "configs[2] sample3 MIS 1024x1024 1024 spp on one GPU": "cornell_1024",
         "sample3 MIS, four quad lights, 256 spp": "cornell_4lights", "configs[4] 4K MIS, 128 of 4096 spp on one GPU": "mis_4k"}
new = {}
for k, f in names.items():
    d = json.load(open(R + f + ".json"))
    new[k] = {x: d[x] for x in old[k].keys() if x in d}
json.dump(new, open(other, "w"), indent=1)
for f in ["sample2_glass", "sample3_mis_cornell", "mesh_1m", "sample1", "cornell_1024", "cornell_4lights", "mis_4k"]:
    d = json.load(open(R + f + ".json"))
    r = d["roofline"]
    print(f, d["ms_per_step"], d["value"], d["mpaths_per_s"], "roof", r["kernel"], r["frac"], r["traffic"], "cpu",
          d["cpu_baseline"]["value"], d.get("gpu_over_cpu"))
