"""Frame time with rtx_render (each frame waited for) and rtx_render_async (two frames in flight).
usage: python tools/async_probe.py [workload] [frames]"""
import sys, time
sys.path.insert(0, '.')
import bench
from raytracer_amd.capi import Config, Renderer

w = sys.argv[1] if len(sys.argv) > 1 else "sample2_glass"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8
wl = bench.workloads()[w]
cfg = Config(width=wl["width"], height=wl["height"], sky=wl["sky"])
with Renderer(cfg) as r:
    r.set_scene(wl["scene"]())
    acc = r.accum_alloc()
    for name, fn in (("render", r.render), ("render_async", r.render_async), ("render", r.render), ("render_async", r.render_async)):
        fn(acc, wl["spp"]); r.wait()
        t0 = time.perf_counter()
        for k in range(n):
            fn(acc, wl["spp"], sample_base=k * wl["spp"])
        r.wait()
        dt = (time.perf_counter() - t0) / n
        print(f"{w} {name:13s} {dt * 1e3:8.3f} ms per frame")
