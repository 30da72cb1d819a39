"""The bench configurations at their full frame size (1024x1024): the HIP renderer against
the oracle with the same counter-based streams, and size-independent properties.

  * sample2 glass, 256 spp (BASELINE configs[1], what bench.py times) and the MIS Cornell box,
    64 spp: per-pixel RMSE to the oracle's image <= 0.15 x the RMSE of an independent-seed
    render (pure Monte-Carlo noise); mean luminance within 0.5 %; bounce, light-sample and
    shadow-ray totals within 0.5 %; no NaN/Inf;
  * linearity over sample ranges: two calls of half the samples give the bits of one call."""
import numpy as np
import pytest

from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name,make,sky,spp", [
    ("sample2_glass", lambda: scenes.sphere_sky(glass=True), 1, 256),
    ("sample3_mis_cornell", scenes.cornell, 0, 64),
])
def test_full_frame_against_oracle(name, make, sky, spp, oracle, rtx_lib):
    cfg = Config(width=1024, height=1024, sky=sky)
    sc = make()
    want, ost = oracle.OracleScene(sc, cfg).render(spp, seed=0x5EED, nthreads=16)
    want = want / spp
    with Renderer(cfg) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        r.render(acc, spp, seed=0x5EED)
        whole_sum = r.resolve(acc, 1)
        st = r.stats()
        r.accum_clear(acc)
        r.render(acc, spp // 2, sample_base=0, seed=0x5EED)
        r.render(acc, spp - spp // 2, sample_base=spp // 2, seed=0x5EED)
        halves_sum = r.resolve(acc, 1)
        r.accum_clear(acc)
        r.render(acc, spp, seed=0xBEEF)
        other = r.resolve(acc, spp)
    got = whole_sum / spp
    assert np.isfinite(got).all()
    assert np.array_equal(whole_sum.view(np.uint32), halves_sum.view(np.uint32))
    rmse = lambda a, b: float(np.sqrt(np.mean((a - b) ** 2)))
    same, noise = rmse(got, want), rmse(other, want)
    assert same <= 0.15 * noise, (name, same, noise)
    assert abs(float(got.mean()) / float(want.mean()) - 1) < 0.005
    assert st.paths == ost.paths == 1024 * 1024 * spp
    assert abs(st.bounces - ost.bounces) <= 0.005 * ost.bounces
    assert abs(st.nee_samples - ost.nee_samples) <= 0.005 * max(ost.nee_samples, 1)


def test_million_triangle_mesh_render_against_oracle(oracle, rtx_lib):
    """The 1 M-triangle mesh of the bench (BASELINE configs[3]) at a reduced frame, 32 spp: same
    bars as above.  The traversal itself is checked bit for bit in test_gpu_trace.py."""
    cfg = Config(width=320, height=180, sky=0)
    sc = scenes.terrain(707)
    spp = 32
    want, ost = oracle.OracleScene(sc, cfg).render(spp, seed=0x5EED, nthreads=16)
    want = want / spp
    with Renderer(cfg) as r:
        info = r.set_scene(sc)
        assert info.bvh_width == 4 and info.num_triangles > 990000
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
    assert same <= 0.15 * noise, (same, noise)
    assert abs(float(got.mean()) / float(want.mean()) - 1) < 0.005
    assert abs(st.bounces - ost.bounces) <= 0.005 * ost.bounces
    assert abs(st.shadow_rays - st.nee_samples) <= st.nee_samples  # shadow rays are a subset of the light samples
    assert abs(st.nee_samples - ost.nee_samples) <= 0.005 * max(ost.nee_samples, 1)
