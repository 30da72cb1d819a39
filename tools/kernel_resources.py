#!/usr/bin/env python3
"""Prints VGPRs, spills, scratch and occupancy of every kernel of the library (the compiler's view,
hipcc -Rpass-analysis=kernel-resource-usage).  usage: python tools/kernel_resources.py [extra hipcc flags]"""
import os
import re
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math", "-fno-slp-vectorize", "-S",
       "--cuda-device-only", "-o", "/tmp/rtx_resources.s", os.path.join(root, "raytracer_amd/csrc/rtx_hip.hip"),
       "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\S+:\d+:\d+: )?\s*(.*?) \[-Rpass", line) or re.search(r":\d+:\d+:\s+(.*?) \[-Rpass", line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        if cur:
            rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if cur:
    rows.append(cur)
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.split("(")[0].strip()
    g = lambda *ks: next((r[k] for k in ks if k in r), "?")
    print(f"{name[-56:]:56s} VGPR {g('VGPRs'):>4s} spill {g('VGPR Spill', 'VGPRs Spill'):>3s} scratch {g('ScratchSize [bytes/lane]'):>5s} "
          f"occ {g('Occupancy [waves/SIMD]'):>2s} LDS {g('LDS Size [bytes/block]')}")
