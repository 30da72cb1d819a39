#!/usr/bin/env python3
"""Random rooms with lights (Cornell with and without its blocks and glass sphere, one or four lights), random frame
sizes, pools, cameras and bounce limits: the bounce kernel that traces its light samples itself against the queued
form (RTX_QUEUE_LIGHT_SAMPLES) and against the per-lane walk (RTX_NO_LEAF_TASKS): same image bits and counts.
usage (GPU box): python tools/room_fuzz.py [first_seed last_seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.capi import Config, Renderer  # noqa: E402


def render(cfg, sc, spp, env):
    for k in ("RTX_QUEUE_LIGHT_SAMPLES", "RTX_NO_LEAF_TASKS", "RTX_NO_PLAIN_KERNELS"):
        os.environ.pop(k, None)
    for k in env:
        os.environ[k] = "1"
    with Renderer(cfg) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        r.render(acc, spp)
        st = r.stats()
        return r.resolve(acc, 1), (st.paths, st.bounces, st.nee_samples, st.shadow_rays, st.nan_paths)


def differences(lo, hi):
    bad = 0
    for seed in range(lo, hi):
        rng = np.random.default_rng(seed)
        sc = scenes.cornell(glass_sphere=bool(rng.integers(0, 2)), boxes=bool(rng.integers(0, 2)),
                            lights=int(rng.choice([1, 4])), light_emission=float(rng.choice([10.0, 0.5, 40.0])))
        cam = int(rng.integers(0, 3))
        kw = dict(width=int(rng.integers(5, 140)), height=int(rng.integers(5, 90)), camera=cam, fov=60.0 if cam == 1 else 50.0,
                  pool_paths=int(rng.choice([256, 700, 4096, 1 << 15])), finish_paths=int(rng.choice([0, 1, 64])),
                  max_bounces=int(rng.choice([0, 0, 1, 3, 6])), sky=int(rng.integers(0, 2)))
        spp = int(rng.integers(1, 5))
        a, sa = render(Config(**kw), sc, spp, [])
        for env in (["RTX_QUEUE_LIGHT_SAMPLES"], ["RTX_NO_LEAF_TASKS"], ["RTX_NO_PLAIN_KERNELS"]):
            b, sb = render(Config(**kw), sc, spp, env)
            if not (np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa == sb):
                bad += 1
                print("DIFFERENT", seed, env, kw, sa, sb, flush=True)
    for k in ("RTX_QUEUE_LIGHT_SAMPLES", "RTX_NO_LEAF_TASKS", "RTX_NO_PLAIN_KERNELS"):
        os.environ.pop(k, None)
    return bad


if __name__ == "__main__":
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 60)
    bad = differences(lo, hi)
    print("seeds", lo, hi, "differences:", bad)
    sys.exit(1 if bad else 0)
