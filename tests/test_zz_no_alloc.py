"""GPU test of rtx_trace_closest_noalloc, the entry point that allocates nothing: its results
against rtx_trace_closest's, bit for bit (kind, primitive, t, normal), on two calls in a row,
the second with fewer rays (the staging array keeps the first call's contents), for every
traversal mode (LDS two-child tree, global four-child tree, analytic objects only), the
size limit, and the property itself: the library's count of hipMalloc calls
(rtx_stats.device_allocations) does not move across a call."""
import numpy as np
import pytest

from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer, RtxError

pytestmark = pytest.mark.gpu

MAX_RAYS = 1 << 18


def same_bits(got, want):
    for g, w in zip(got, want):
        assert g.shape == w.shape
        assert np.array_equal(g.view(np.uint32), w.view(np.uint32))


@pytest.mark.parametrize("make", [scenes.cornell, lambda: scenes.soup(10000, 3), lambda: scenes.reference_scene(0)],
                         ids=["cornell_lds_tree", "soup10k_wide_tree", "analytic_objects"])
def test_noalloc_matches_trace_closest_twice(make, rtx_lib):
    sc = make()
    with Renderer(Config(width=64, height=64, pool_paths=1 << 16)) as r:
        r.set_scene(sc)
        big = scenes.random_rays(MAX_RAYS, seed=11)       # the maximum size; several launches (pool < n)
        small = scenes.random_rays(5000, seed=12)          # then a smaller call on the same staging array
        for rays in (big, small):
            before = r.stats().device_allocations
            want = r.trace_closest(rays)
            mid = r.stats().device_allocations
            got = r.trace_closest(rays, noalloc=True)
            # the property itself: the allocating entry point made hipMalloc calls (the counter is live),
            # the other one made none
            assert mid > before and r.stats().device_allocations == mid
            same_bits(got, want)
            assert (want[0] != 0).any()
        # without normals
        mid = r.stats().device_allocations
        got = r.trace_closest(small, want_normal=False, noalloc=True)
        assert r.stats().device_allocations == mid
        want = r.trace_closest(small, want_normal=False)
        same_bits(got[:3], want[:3])
        assert got[3] is None


def test_noalloc_rejects_oversized_calls(rtx_lib):
    with Renderer(Config(width=64, height=64, pool_paths=1 << 12)) as r:
        r.set_scene(scenes.cornell())
        with pytest.raises(RtxError) as e:
            r.trace_closest(np.zeros((MAX_RAYS + 1, 6), np.float32), noalloc=True)
        assert e.value.status == -1 and "262144" in str(e.value)
        # and the context still works
        rays = scenes.random_rays(1000, seed=5)
        same_bits(r.trace_closest(rays, noalloc=True), r.trace_closest(rays))


def test_noalloc_two_contexts_share_the_device_array(rtx_lib):
    """Two contexts on one GPU use the same staging array one after the other."""
    rays = scenes.random_rays(3000, seed=9)
    with Renderer(Config(width=32, height=32, pool_paths=1 << 12)) as a, \
            Renderer(Config(width=32, height=32, pool_paths=1 << 12)) as b:
        a.set_scene(scenes.cornell())
        b.set_scene(scenes.sphere_sky())
        wa, wb = a.trace_closest(rays), b.trace_closest(rays)
        same_bits(a.trace_closest(rays, noalloc=True), wa)
        same_bits(b.trace_closest(rays, noalloc=True), wb)
        same_bits(a.trace_closest(rays, noalloc=True), wa)
