#!/usr/bin/env python3
"""Two bounces per launch of the light-free bounce kernel of scenes without a tree (the default)
against one (RTX_FEW_BOUNCES=1): same image bits and counts, frame times of the sample2 glass workload.
usage (GPU box): python tools/few_bounces_probe.py"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.capi import Config, Renderer  # noqa: E402


def render(cfg, sc, spp, reps=1):
    with Renderer(cfg) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        r.render(acc, spp)
        t0 = time.perf_counter()
        for _ in range(reps - 1):
            r.render(acc, spp)
        dt = (time.perf_counter() - t0) / max(reps - 1, 1)
        st = r.stats()
        return r.resolve(acc, 1), (st.paths, st.bounces, st.nee_samples, st.shadow_rays), dt


ok = True
for sc in (scenes.sphere_sky(glass=True), scenes.sphere_sky(glass=False)):
    for kw in (dict(width=200, height=120, sky=1, pool_paths=1 << 13), dict(width=97, height=61, sky=1, pool_paths=700, max_bounces=5)):
        os.environ["RTX_FEW_BOUNCES"] = "1"
        ia, sa, _ = render(Config(**kw), sc, 4)
        for k in ("2",):
            os.environ["RTX_FEW_BOUNCES"] = k
            ib, sb, _ = render(Config(**kw), sc, 4)
            same = np.array_equal(ia.view(np.uint32), ib.view(np.uint32)) and sa == sb
            ok &= same
            print(sc.name, kw, k, "same bits and counts" if same else f"DIFFERENT {sa} {sb}", flush=True)
if not ok:
    sys.exit(1)
cfg = Config(width=1024, height=1024, sky=1)
for rep in range(2):
    for k in ("1", "2"):
        os.environ["RTX_FEW_BOUNCES"] = k
        img, st, dt = render(cfg, scenes.sphere_sky(glass=True), 256, reps=8)
        print(f"sample2 1024x1024x256, {k} bounces per launch: {dt * 1e3:.3f} ms per frame (frame by frame), bounces {st[1]}", flush=True)
