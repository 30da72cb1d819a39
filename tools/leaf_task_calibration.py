"""Where do the leaf-task kernels start to pay?  Times small closed and open scenes with the per-lane walk
(RTX_NO_LEAF_TASKS) and with the tasks forced (RTX_LEAF_TASKS_ALWAYS) and prints the sum of the leaf boxes' surfaces
over the root box's beside the two times.  usage (GPU box): python tools/leaf_task_calibration.py"""
import os, sys, time
sys.path.insert(0, '.')
import numpy as np
from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer


def leaf_area_ratio(tris):
    """A stand-in for the host's figure (which uses the builder's leaves): pairs of consecutive triangles as leaves."""
    t = tris.reshape(-1, 3, 3)
    lo, hi = t.min((0, 1)), t.max((0, 1))
    def area(a, b):
        d = np.maximum(b - a, 0)
        return d[0] * d[1] + d[1] * d[2] + d[2] * d[0]
    s = 0.0
    for k in range(0, len(t), 2):
        p = t[k:k + 2].reshape(-1, 3)
        s += area(p.min(0), p.max(0))
    return s / area(lo, hi)


def timed(sc, sky, env):
    for k in ("RTX_NO_LEAF_TASKS", "RTX_LEAF_TASKS_ALWAYS"):
        os.environ.pop(k, None)
    os.environ[env] = "1"
    with Renderer(Config(width=1024, height=1024, sky=sky)) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        r.render(acc, 16)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            r.render(acc, 64)
            best = min(best, time.perf_counter() - t0)
    return best * 1e3


cases = [("cornell", scenes.cornell(), 0)]
for size in (0.15, 0.4, 0.8, 1.5, 3.0):
    cases.append((f"soup30 size {size}", scenes.soup(30, 2, seed=2, size=size), 1))
for name, sc, sky in cases:
    a, b = timed(sc, sky, "RTX_NO_LEAF_TASKS"), timed(sc, sky, "RTX_LEAF_TASKS_ALWAYS")
    print(f"{name:20s} leaf/root surface {leaf_area_ratio(sc.triangles):5.2f}  walk {a:7.2f} ms  tasks {b:7.2f} ms  ({b / a:.2f})")
