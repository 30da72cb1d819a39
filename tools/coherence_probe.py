import sys, time
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer
sc = scenes.terrain(707)
for mb in (1, 2, 3, 0):
    cfg = Config(width=1920, height=1080, sky=0, max_bounces=mb)
    with Renderer(cfg) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        r.render(acc, 4)
        r.reset_stats(); r.set_profiling(True)
        r.render(acc, 16)
        st = r.stats()
        print(f"max_bounces {mb}: bounces/path {st.bounces/st.paths:.2f} intersect {st.ms_intersect:.2f} ms -> {st.ms_intersect*1e6/st.bounces:.3f} ns/ray; shadow {st.ms_shadow:.2f} ms -> {st.ms_shadow*1e6/max(st.shadow_rays,1):.3f} ns/ray")
