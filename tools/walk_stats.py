"""What the waves of the wide-tree search kernel do, from a diagnostic build of the library
(make -C raytracer_amd/csrc OUT=../librtx_hip_ws.so EXTRA=-DRTX_WALK_STATS).
usage: RTX_LIB=raytracer_amd/librtx_hip_ws.so python tools/walk_stats.py [WORKLOAD] [spp]"""
import ctypes as C
import sys

import torch

import bench
from raytracer_amd import capi
from raytracer_amd.capi import Config, Renderer

name = sys.argv[1] if len(sys.argv) > 1 else "mesh_1m"
wl = bench.workloads()[name]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 16
lib = capi.load_library()
out = (C.c_ulonglong * 8)()
with Renderer(Config(width=wl["width"], height=wl["height"], sky=wl["sky"])) as r:
    r.set_scene(wl["scene"]())
    acc = torch.zeros(wl["width"] * wl["height"] * 3, device="cuda")
    r.render(acc, spp)
    torch.cuda.synchronize()
    st = r.stats()
    assert lib.rtx_debug_walk_stats(out) == 0
it, node, leaf, both, nonly, lonly, refill, _ = list(out)
rays = st.bounces
print(f"{name}: {rays} rays, {it} wave iterations, {refill} refill stops")
print(f"  per ray: {node / rays:.2f} node steps, {leaf / rays:.2f} triangle steps")
print(f"  per iteration: {node / it:.1f} lanes at a node, {leaf / it:.1f} at a triangle, {(node + leaf) / it:.1f} walking")
print(f"  iterations with both kinds {both / it:.3f}, nodes only {nonly / it:.3f}, triangles only {lonly / it:.3f}")
print(f"  rays per wave iteration {rays / it:.3f}; refill stop every {it / max(refill, 1):.1f} iterations")
