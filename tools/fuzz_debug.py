"""Prints where one fuzz case (tests/test_gpu_fuzz.py) differs from the oracle.  usage: python tools/fuzz_debug.py SEED..."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import test_gpu_fuzz as t
from oracle import oracle_py as orc
from raytracer_amd.capi import Config, Renderer

for seed in map(int, sys.argv[1:]):
    sc, kw = t.random_case(seed)
    spp = 1 + seed % 3
    print("seed", seed, kw, "tris", len(sc.triangles), "spheres", len(sc.spheres), "boxes", len(sc.boxes), "planes", len(sc.planes))
    for integ, n in ((1, spp), (0, 1)):
        for fused in (0, 1):
            cfg = Config(integrator=integ, **dict(kw, fused_trace=fused))
            want, ost = orc.OracleScene(sc, cfg).render(n)
            with Renderer(cfg) as r:
                info = r.set_scene(sc)
                acc = r.accum_alloc()
                r.render(acc, n)
                got = r.resolve(acc, 1)
                st = r.stats()
            same = np.array_equal(got.view(np.uint32), want.view(np.uint32))
            ok = np.all(np.abs(got - want) <= 1e-5 + 1e-3 * np.abs(want), axis=-1) | ~np.isfinite(want).all(-1)
            bad = np.argwhere(~ok)
            print(f" integrator {integ} fused {fused} width {info.bvh_width}: bit-equal {same}, within tol {ok.mean():.4f}, "
                  f"bounces {st.bounces} vs {ost.bounces}, paths {st.paths} vs {ost.paths}, first bad {bad[:3].tolist()}")
            for y, x in bad[:3]:
                print("   ", got[y, x], want[y, x])
