"""Random fan-tessellated quad meshes on the wide tree (leaf pairs, dense leaf records) against the oracle's linear
scan: closest hit (kind, primitive, t bits, normal bits) and occlusion, for many seeds and mesh sizes.
usage: python tools/quad_fuzz.py [first_seed] [count]"""
import sys

import numpy as np

from oracle import oracle_py
from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer

first = int(sys.argv[1]) if len(sys.argv) > 1 else 0
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
bad = 0
for seed in range(first, first + count):
    rng = np.random.default_rng(seed)
    nquad = int(rng.integers(150, 4000))
    sc = scenes.quad_soup(nquad, seed=seed, extent=float(rng.uniform(0.3, 1.5)), size=float(rng.uniform(0.02, 0.6)))
    cfg = Config(width=64, height=64, pool_paths=1 << 14)
    osc = oracle_py.OracleScene(sc, cfg)
    rays = np.concatenate([scenes.random_rays(6000, seed=seed + 1000), scenes.random_rays(1500, seed=seed + 2000, unit=False)])
    # rays that start on surfaces
    k, p, t, n = osc.trace_closest(rays)
    hit = k != 0
    o = rays[hit, :3] + rays[hit, 3:] * t[hit, None]
    d = rng.normal(size=o.shape)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    rays = np.concatenate([rays, np.concatenate([o, d], 1).astype(np.float32)])
    dist = rng.uniform(0.0, 3.0, len(rays)).astype(np.float32)
    with Renderer(cfg) as r:
        info = r.set_scene(sc)
        gk, gp, gt, gn = r.trace_closest(rays)
        gocc = r.trace_occluded(rays, dist)
    wk, wp, wt, wn = osc.trace_closest(rays, use_bvh=False)
    wocc = osc.trace_occluded(rays, dist, use_bvh=False)
    ok = (np.array_equal(gk, wk) and np.array_equal(gp, wp) and np.array_equal(gt.view(np.uint32), wt.view(np.uint32))
          and np.array_equal(gn[wk != 0].view(np.uint32), wn[wk != 0].view(np.uint32)) and np.array_equal(gocc, wocc))
    if not ok:
        bad += 1
        print(f"seed {seed}: MISMATCH ({nquad} quads, width {info.bvh_width})", flush=True)
    elif seed % 20 == 0:
        print(f"seed {seed}: ok ({nquad} quads, {len(sc.triangles)} triangles, width {info.bvh_width}, {int((wk != 0).sum())} hits)", flush=True)
print(f"{count} seeds, {bad} mismatches")
sys.exit(1 if bad else 0)
