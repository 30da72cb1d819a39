#!/usr/bin/env python3
"""Reduces rocprofv3 counter-collection CSVs of SQ counter passes (each pass a separate run of
the same bench.py command) to per-kernel means per launch, as JSON and as a text table.

usage: pmc_sq.py WORKLOAD OUT.json OUT.txt PASS1.csv [PASS2.csv ...]

Per kernel the JSON holds, for every counter collected, the sum over the kernel's dispatches
divided by their number ("per launch"), plus derived figures:
  valu_per_wave          SQ_INSTS_VALU / SQ_WAVES
  lanes_active           SQ_THREAD_CYCLES_VALU / SQ_ACTIVE_INST_VALU  (of 64)
  valu_active_frac       SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES         (share of a wave's life spent issuing VALU)
  lds_conflict_frac      SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
"""
import collections
import csv
import json
import sys


def kernel_name(raw):
    return raw.split("(")[0].split("::")[-1].split("<")[0]


def main():
    workload, out_json, out_txt = sys.argv[1:4]
    sums = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.defaultdict(lambda: collections.defaultdict(int))
    for path in sys.argv[4:]:
        for r in csv.DictReader(open(path)):
            k = kernel_name(r["Kernel_Name"])
            if not k.startswith("k_"):
                continue
            sums[k][r["Counter_Name"]] += float(r["Counter_Value"])
            launches[k][r["Counter_Name"]] += 1
    res = {"workload": workload, "units": "counter sums per launch (mean over the launches of one bench step)",
           "kernels": {}}
    lines = []
    for k in sorted(sums):
        e = {c: sums[k][c] / max(launches[k][c], 1) for c in sorted(sums[k])}
        e["launches"] = min(launches[k].values())  # (a counter collected in two passes has twice the rows)
        g = lambda c: e.get(c, 0.0)
        if g("SQ_WAVES"):
            e["valu_per_wave"] = g("SQ_INSTS_VALU") / g("SQ_WAVES")
        if g("SQ_ACTIVE_INST_VALU"):
            e["lanes_active"] = g("SQ_THREAD_CYCLES_VALU") / g("SQ_ACTIVE_INST_VALU") if g("SQ_THREAD_CYCLES_VALU") else None
        if g("SQ_WAVE_CYCLES"):
            e["valu_active_frac"] = g("SQ_ACTIVE_INST_VALU") / g("SQ_WAVE_CYCLES")
            e["wait_inst_any_frac"] = g("SQ_WAIT_INST_ANY") / g("SQ_WAVE_CYCLES")
        if g("SQ_LDS_IDX_ACTIVE"):
            e["lds_conflict_frac"] = g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")
        res["kernels"][k] = e
        lines.append(f"{k:24s} launches {e['launches']:4d} waves/launch {g('SQ_WAVES'):9.0f} VALU/wave {e.get('valu_per_wave', 0):8.1f} "
                     f"lanes_active {e.get('lanes_active') or 0:6.2f} valu_active/wave_cycles {e.get('valu_active_frac', 0):.3f} "
                     f"wait_inst_any/wave_cycles {e.get('wait_inst_any_frac', 0):.3f} LDS insts/launch {g('SQ_INSTS_LDS'):12.0f} "
                     f"lds_conflict/lds_active {e.get('lds_conflict_frac', 0):.3f}")
    json.dump(res, open(out_json, "w"), indent=1)
    open(out_txt, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
