"""Where the waves of the bounce kernel (k_trace_shade) spend their cycles, from a diagnostic build of the library
(make -C raytracer_amd/csrc OUT=../librtx_hip_ps.so EXTRA=-DRTX_PHASE_STATS).
usage: RTX_LIB=raytracer_amd/librtx_hip_ps.so python tools/phase_stats.py [WORKLOAD] [spp]"""
import ctypes as C
import sys
import time

import torch

import bench
from raytracer_amd import capi
from raytracer_amd.capi import Config, Renderer

PHASES = ["stage scene in LDS (to barrier)", "load path, analytic objects", "closest: push tasks", "barrier 1",
          "triangle tasks (closest)", "barrier 2", "shade", "analytic any, leaf boxes, push (light sample)",
          "barrier 3", "triangle tasks (any)", "barrier 4", "append, store", "  closest: walk_begin", "  closest: leaf boxes"]
OTHER = ["start to path request", "load path / primary ray", "first bounce: search", "first bounce: shade, store result",
         "first bounce: count", "last bounce: search", "last bounce: shade_arrive", "reserve slot, sample next direction",
         "", "", "", "store path", "", ""]
name = sys.argv[1] if len(sys.argv) > 1 else "sample3_mis_cornell"
wl = bench.workloads()[name]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
lib = capi.load_library()
out = (C.c_ulonglong * 16)()
with Renderer(Config(width=wl["width"], height=wl["height"], sky=wl["sky"])) as r:
    r.set_scene(wl["scene"]())
    acc = torch.zeros(wl["width"] * wl["height"] * 3, device="cuda")
    r.render(acc, spp)
    torch.cuda.synchronize()
    assert lib.rtx_debug_phase_stats(out) == 0
    t0 = time.perf_counter()
    r.render(acc, spp)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    assert lib.rtx_debug_phase_stats(out) == 0
v = list(out)
waves = v[15]
total = sum(v[:14])
print(f"{name}: frame {ms:.1f} ms; {waves} waves sampled, {total / waves:.0f} cycles per wave")
labels = PHASES if v[13] else OTHER   # (only the leaf-task kernel stamps slot 13)
for k, label in enumerate(labels):
    if label:
        print(f"  {v[k] / waves:8.0f}  {100 * v[k] / total:5.1f} %  {label}")
