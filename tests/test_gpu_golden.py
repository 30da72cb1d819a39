"""GPU parity against the committed converged images (tests/golden/image_*.npz: the CPU oracle,
128x128, 1024 spp, seed 0x5EED).  The HIP renderer draws the same counter-based streams, so its
image differs from the committed one only where a last-ulp difference of sinf/cosf changed a path.

Tolerances (per scene):
  * per-pixel RMSE to the committed image <= 0.15 x the RMSE of an independent-seed render of the
    same spp to the committed image (that second figure is pure Monte-Carlo noise);
  * mean luminance within 0.5%; bounce and light-sample totals within 0.5%; no NaN/Inf."""
import os

import numpy as np
import pytest

from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

SCENES = {
    "cornell": lambda: scenes.cornell(),
    "cornell_4lights": lambda: scenes.cornell(lights=4),
    "sample1": lambda: scenes.sphere_sky(),
    "sample2": lambda: scenes.sphere_sky(glass=True),
    "ref_scene0": lambda: scenes.reference_scene(0),
    "soup3k": lambda: scenes.soup(3000, 3),
    "terrain40": lambda: scenes.terrain(40),
}


@pytest.mark.parametrize("name", sorted(SCENES))
def test_converged_image_within_tolerance(name, rtx_lib):
    g = np.load(os.path.join(GOLD, f"image_{name}.npz"))
    want, spp = g["mean"], int(g["spp"])
    cfg = Config(width=128, height=128, sky=int(g["sky"]))
    with Renderer(cfg) as r:
        r.set_scene(SCENES[name]())
        acc = r.accum_alloc()
        r.render(acc, spp, seed=0x5EED)
        got = r.resolve(acc, spp)
        st = r.stats()
        r.accum_clear(acc)
        r.render(acc, spp, seed=0xBEEF)
        other = r.resolve(acc, spp)
    assert np.isfinite(got).all()
    rmse = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)))
    same, noise = rmse(got, want), rmse(other, want)
    assert same <= 0.15 * noise, (name, same, noise)
    assert abs(float(got.mean()) / float(want.mean()) - 1) < 0.005
    assert abs(st.bounces - int(g["bounces"])) <= 0.005 * int(g["bounces"])
    assert abs(st.nee_samples - int(g["nee_samples"])) <= 0.005 * max(int(g["nee_samples"]), 1)
