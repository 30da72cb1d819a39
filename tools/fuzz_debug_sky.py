"""One fuzz case path-traced with the sky switched on (every path then carries light), to locate the pixels whose
paths part from the oracle's.  usage: python tools/fuzz_debug_sky.py SEED..."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np
import test_gpu_fuzz as t
from oracle import oracle_py as orc
from raytracer_amd.capi import Config, Renderer

for seed in map(int, sys.argv[1:]):
    sc, kw = t.random_case(seed)
    kw = dict(kw, sky=1)
    cfg = Config(integrator=0, **kw)
    want, ost = orc.OracleScene(sc, cfg).render(1)
    with Renderer(cfg) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        r.render(acc, 1)
        got = r.resolve(acc, 1)
        st = r.stats()
    same = got.view(np.uint32) == want.view(np.uint32)
    ok = np.all(np.abs(got - want) <= 1e-5 + 1e-3 * np.abs(want), axis=-1)
    print("seed", seed, "pixels", ok.size, "bit-equal pixels", int(same.all(-1).sum()), "outside tolerance", int((~ok).sum()),
          "bounces", st.bounces, ost.bounces)
    for y, x in np.argwhere(~ok)[:8]:
        print("   ", y, x, got[y, x], want[y, x])
