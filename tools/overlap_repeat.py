"""Repeats one render with the split ordered sum forced on and compares every result with the
plain sum, bit for bit (a race between the drain kernel and the sum beside it would show here).
usage: python tools/overlap_repeat.py [repeats]"""
import os, sys
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
def render(sc, sky, threshold, **kw):
    os.environ["RTX_OVERLAP_MIN_SAMPLES"] = threshold
    with Renderer(Config(width=768, height=512, sky=sky, **kw)) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        out = []
        for _ in range(reps if threshold == "1" else 1):
            r.accum_clear(acc)
            r.render(acc, 48)
            out.append(r.resolve(acc, 1))
        return out
bad = 0
for sc, sky in [(scenes.sphere_sky(glass=True), 1), (scenes.cornell(), 0), (scenes.terrain(60), 0)]:
    for kw in (dict(), dict(finish_paths=1 << 16, pool_paths=1 << 20)):
        want = render(sc, sky, "18446744073709551615", **kw)[0]
        for i, got in enumerate(render(sc, sky, "1", **kw)):
            if not np.array_equal(got.view(np.uint32), want.view(np.uint32)):
                bad += 1
                print(sc.name, kw, "repeat", i, "differs in", int((got.view(np.uint32) != want.view(np.uint32)).sum()), "floats")
print("differences:", bad)
