#!/usr/bin/env python3
"""Random light-free scenes of up to eight triangles and a few spheres (tested without a tree), random frame sizes,
pools, cameras, bounce limits, sky on and off: two bounces per launch against one (RTX_FEW_BOUNCES=1) and against
the general build of the kernels (RTX_NO_PLAIN_KERNELS): same image bits and counts.
usage (GPU box): python tools/few_fuzz.py [first_seed last_seed]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raytracer_amd import scenes  # noqa: E402
from raytracer_amd.capi import Config, Renderer  # noqa: E402

HOOKS = ("RTX_FEW_BOUNCES", "RTX_NO_PLAIN_KERNELS")


def render(cfg, sc, spp, env):
    for k in HOOKS:
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
        if rng.random() < 0.3:
            sc = scenes.sphere_sky(glass=bool(rng.integers(0, 2)))
        else:
            sc = scenes.soup(int(rng.integers(1, 9)), int(rng.integers(0, 4)), seed=seed + 7, size=float(rng.uniform(0.2, 1.5)))
        cam = int(rng.integers(0, 3))
        kw = dict(width=int(rng.integers(5, 160)), height=int(rng.integers(5, 100)), camera=cam, fov=60.0 if cam == 1 else 50.0,
                  pool_paths=int(rng.choice([256, 700, 4096, 1 << 15])), finish_paths=int(rng.choice([0, 1, 64])),
                  max_bounces=int(rng.choice([0, 0, 1, 2, 5])), sky=int(rng.integers(0, 2)),
                  primary_rays=int(rng.integers(0, 2)))
        spp = int(rng.integers(1, 6))
        a, sa = render(Config(**kw), sc, spp, [])
        for env in (["RTX_FEW_BOUNCES"], ["RTX_NO_PLAIN_KERNELS"]):
            b, sb = render(Config(**kw), sc, spp, env)
            if not (np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa == sb):
                bad += 1
                print("DIFFERENT", seed, env, kw, sa, sb, flush=True)
    for k in HOOKS:
        os.environ.pop(k, None)
    return bad


if __name__ == "__main__":
    lo, hi = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (0, 60)
    bad = differences(lo, hi)
    print("seeds", lo, hi, "differences:", bad)
    sys.exit(1 if bad else 0)
