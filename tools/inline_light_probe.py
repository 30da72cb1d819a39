#!/usr/bin/env python3
"""Light samples traced inside the bounce kernel (the default for rooms of a few surfaces) against the queued
form (RTX_QUEUE_LIGHT_SAMPLES): same image bits and counts on rooms with lights, then frame times of the MIS Cornell box.
usage (GPU box): python tools/inline_light_probe.py"""
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


def both(cfg, sc, spp, reps=1):
    os.environ["RTX_QUEUE_LIGHT_SAMPLES"] = "1"
    a = render(cfg, sc, spp, reps)
    os.environ.pop("RTX_QUEUE_LIGHT_SAMPLES", None)
    b = render(cfg, sc, spp, reps)
    return a, b


ok = True
for sc in (scenes.cornell(), scenes.cornell(lights=4)):
    for kw in (dict(width=200, height=120, pool_paths=1 << 13), dict(width=97, height=61, pool_paths=700, max_bounces=4)):
        (ia, sa, _), (ib, sb, _) = both(Config(**kw), sc, 4)
        same = np.array_equal(ia.view(np.uint32), ib.view(np.uint32)) and sa == sb
        ok &= same
        print(sc.name, kw, "same bits and counts" if same else f"DIFFERENT {sa} {sb}", flush=True)
if not ok:
    sys.exit(1)
cfg = Config(width=1024, height=1024)
for rep in range(2):
    (ia, sa, ta), (ib, sb, tb) = both(cfg, scenes.cornell(), 256, reps=4)
    print(f"cornell 1024x1024x256: queued {ta * 1e3:.2f} ms, inline {tb * 1e3:.2f} ms, same bits "
          f"{np.array_equal(ia.view(np.uint32), ib.view(np.uint32))}", flush=True)
