"""GPU parity, rendering: the wavefront path tracer (rtx_render through the C ABI)
against the CPU oracle on the same counter-based random streams.

Integer/index work and trig-free float work must be bit-exact.  Paths that call
sinf/cosf (diffuse bounces, Integrator.h:22-23) cannot be: glibc and the device
maths library differ in the last ulp, so a path's direction may differ by an ulp
and, rarely, a discrete decision flips.  The bars, stated per test:
  * normals integrator (no trig): image bit-exact;
  * 1 spp path tracing: >= 99% of pixels within 1e-3 relative, bounce counts within 0.5%;
  * converged images: mean luminance within 0.5%, RMSE below the Monte-Carlo noise floor;
  * oven box: mean 1.0 +- 0.5% (analytic answer, reference Raytracer.cpp:203-214).
"""
import numpy as np
import pytest

from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer

pytestmark = pytest.mark.gpu


def gpu_render(cfg, sc, spp, calls=None, **kw):
    with Renderer(cfg) as r:
        info = r.set_scene(sc)
        acc = r.accum_alloc()
        if calls is None:
            r.render(acc, spp, **kw)
        else:
            for c in calls:
                r.render(acc, **c)
        img = r.resolve(acc, 1)  # sums
        return img, r.stats(), info


def luminance(img):
    return float((img * np.array([0.2126, 0.7152, 0.0722], np.float32)).sum(-1).mean())


def close_fraction(a, b, rtol=1e-3, atol=1e-5):
    ok = np.all(np.abs(a - b) <= atol + rtol * np.abs(b), axis=-1)
    return float(ok.mean())


def test_normals_integrator_bit_exact(oracle, rtx_lib):
    for sc, sky in [(scenes.cornell(), 0), (scenes.sphere_sky(), 1), (scenes.soup(2000, 3), 1)]:
        cfg = Config(width=160, height=120, integrator=1, sky=sky, pool_paths=1 << 13)
        want, ost = oracle.OracleScene(sc, cfg).render(2)
        got, st, _ = gpu_render(cfg, sc, 2)
        # Two samples per pixel are added in either order: a+b is commutative, so still exact.
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), sc.name
        assert st.paths == 160 * 120 * 2 and st.bounces == ost.bounces


@pytest.mark.parametrize("finish_paths", [0, 1], ids=["finish_default", "wavefront_only"])
@pytest.mark.parametrize("name,sky,camera", [
    ("cornell", 0, 0), ("cornell_4l", 0, 0), ("sample1", 1, 0), ("sample2", 1, 0),
    ("cornell_thinlens", 0, 1), ("cornell_ortho", 0, 2), ("soup", 1, 0),
    ("ref_scene0", 1, 0), ("ref_scene1", 0, 0), ("ref_scene3", 0, 0),
])
def test_one_spp_per_path_parity(name, sky, camera, finish_paths, oracle, rtx_lib):
    sc = {"cornell": scenes.cornell, "cornell_4l": lambda: scenes.cornell(lights=4),
          "sample1": scenes.sphere_sky, "sample2": lambda: scenes.sphere_sky(glass=True),
          "cornell_thinlens": scenes.cornell, "cornell_ortho": scenes.cornell,
          "soup": lambda: scenes.soup(3000, 3),
          "ref_scene0": lambda: scenes.reference_scene(0), "ref_scene1": lambda: scenes.reference_scene(1),
          "ref_scene3": lambda: scenes.reference_scene(3)}[name]()
    cfg = Config(width=128, height=128, sky=sky, camera=camera, fov=60.0 if camera == 1 else 50.0,
                 pool_paths=1 << 12,  # pool < pixels: the queue is refilled several times
                 finish_paths=finish_paths)
    want, ost = oracle.OracleScene(sc, cfg).render(1)
    got, st, _ = gpu_render(cfg, sc, 1)
    assert st.paths == ost.paths == 128 * 128
    frac = close_fraction(got, want)
    assert frac >= 0.99, f"{name}: only {frac:.4f} of pixels agree"
    # Closed +-800 rooms with glass: a path that diverges by one decision can differ by hundreds
    # of bounces, so the totals of 16k paths get a wider band there.
    btol = 0.02 if name in ("ref_scene1", "ref_scene3") else 0.005
    assert abs(st.bounces - ost.bounces) <= btol * ost.bounces, (st.bounces, ost.bounces)
    assert abs(st.nee_samples - ost.nee_samples) <= 0.005 * max(ost.nee_samples, 1)
    assert np.isfinite(got).all() == np.isfinite(want).all()


def test_converged_image_parity(oracle, rtx_lib):
    """Same seeds: the two images agree far below the noise floor.  Different seeds: they
    agree within Monte-Carlo noise.  Tolerances: mean luminance 0.5%; same-seed RMSE
    <= 10% of the different-seed RMSE (which is pure noise)."""
    sc = scenes.cornell()
    cfg = Config(width=64, height=64)
    spp = 256
    osc = oracle.OracleScene(sc, cfg)
    want = osc.render(spp, seed=0x5EED)[0] / spp
    other = osc.render(spp, seed=0xBEEF)[0] / spp
    got = gpu_render(cfg, sc, spp, seed=0x5EED)[0] / spp
    assert np.isfinite(got).all()
    noise = float(np.sqrt(np.mean((want - other) ** 2)))
    same = float(np.sqrt(np.mean((got - want) ** 2)))
    cross = float(np.sqrt(np.mean((got - other) ** 2)))
    assert abs(luminance(got) / luminance(want) - 1) < 0.005
    assert same <= 0.1 * noise, (same, noise)
    assert cross <= 1.25 * noise, (cross, noise)


@pytest.mark.parametrize("make", [scenes.oven, lambda: scenes.reference_scene(2), scenes.oven_objects],
                         ids=["triangles", "planes", "glass_and_mirror_inside"])
def test_oven_box_is_one(make, rtx_lib):
    """The reference's oven test (Raytracer.cpp:203-214), from triangles, from its six planes, and
    with a glass and a mirror sphere of albedo 1 inside (the answer stays 1.0)."""
    cfg = Config(width=64, height=64)
    spp = 256
    got, st, _ = gpu_render(cfg, make(), spp)
    img = got / spp
    assert abs(img.mean() - 1.0) < 0.005, img.mean()
    assert np.isfinite(img).all()
    assert st.nee_samples == 0  # emission 0.5 >= threshold, or no mesh lights: NEE is skipped (Integrator.h:209)


def test_sample_ranges_row_bands_and_chunks_are_bit_identical(rtx_lib):
    """Samples are added per pixel in sample-index order, whatever the queue order was:
    consecutive sample ranges, row bands (tiles), result-buffer chunks, another pool size
    and a second run all give the same bits."""
    sc = scenes.cornell()
    cfg = Config(width=96, height=64, pool_paths=1 << 12)
    whole, st, _ = gpu_render(cfg, sc, 8)
    again, _, _ = gpu_render(cfg, sc, 8)
    split, st2, _ = gpu_render(cfg, sc, 0, calls=[dict(spp=3, sample_base=0), dict(spp=5, sample_base=3)])
    bands, st3, _ = gpu_render(cfg, sc, 0, calls=[dict(spp=8, row0=0, row1=20), dict(spp=8, row0=20, row1=64)])
    # 96*64*12 B = 72 KiB per spp: a 1 MiB result buffer forces chunks of 14... use 3 spp by hand
    chunked, st4, _ = gpu_render(Config(width=96, height=64, pool_paths=1 << 14, finish_paths=1,
                                        result_buffer_mib=1), sc, 0,
                                 calls=[dict(spp=8)])
    for other in (again, split, bands, chunked):
        assert np.array_equal(other.view(np.uint32), whole.view(np.uint32))
    assert st.paths == st2.paths == st3.paths == st4.paths == 96 * 64 * 8
    assert st.bounces == st2.bounces == st3.bounces == st4.bounces
    assert st.nee_samples == st4.nee_samples and st.shadow_rays == st4.shadow_rays


@pytest.mark.parametrize("camera", [0, 2], ids=["perspective", "orthogonal"])
def test_recomputed_and_stored_primary_rays_give_the_same_bits(camera, rtx_lib):
    """For LDS-resident scenes the kernels recompute the primary ray of a fresh path from its
    sample number instead of reading what k_generate wrote (primary_rays = 0); primary_rays = 1
    forces the stored form.  Same arithmetic, so the images and the counts are identical."""
    for sc, sky in [(scenes.cornell(), 0), (scenes.sphere_sky(glass=True), 1)]:
        kw = dict(width=200, height=120, sky=sky, camera=camera, pool_paths=1 << 13, max_bounces=0)
        a, sa, _ = gpu_render(Config(primary_rays=0, **kw), sc, 6)
        b, sb, _ = gpu_render(Config(primary_rays=1, **kw), sc, 6)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
        assert (sa.paths, sa.bounces, sa.nee_samples, sa.shadow_rays) == (sb.paths, sb.bounces, sb.nee_samples, sb.shadow_rays)
        c, sc_, _ = gpu_render(Config(primary_rays=0, **dict(kw, max_bounces=4)), sc, 3)
        d, sd, _ = gpu_render(Config(primary_rays=1, **dict(kw, max_bounces=4)), sc, 3)
        assert np.array_equal(c.view(np.uint32), d.view(np.uint32)) and sc_.bounces == sd.bounces


@pytest.mark.parametrize("camera", [0, 1, 2], ids=["perspective", "thin_lens", "orthogonal"])
def test_one_kernel_and_two_kernel_bounces_give_the_same_bits(camera, rtx_lib):
    """Scenes whose tree is staged in LDS search and shade in one kernel (fused_trace = 0);
    fused_trace = 1 keeps k_intersect and k_shade apart.  Same arithmetic per path and per-sample
    result slots, so images and counts are identical, with either source of primary rays."""
    for sc, sky in [(scenes.cornell(), 0), (scenes.sphere_sky(glass=True), 1), (scenes.soup(150, 3, seed=2), 1)]:
        for primary in (0, 1):
            kw = dict(width=160, height=100, sky=sky, camera=camera, pool_paths=1 << 13, primary_rays=primary)
            a, sa, ia = gpu_render(Config(fused_trace=0, **kw), sc, 5)
            b, sb, _ = gpu_render(Config(fused_trace=1, **kw), sc, 5)
            assert ia.bvh_width == 2
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
            assert (sa.paths, sa.bounces, sa.nee_samples, sa.shadow_rays) == (sb.paths, sb.bounces, sb.nee_samples, sb.shadow_rays)
            assert sa.launches < sb.launches
        c, sc_, _ = gpu_render(Config(fused_trace=0, max_bounces=3, integrator=0, **kw), sc, 2)
        d, sd, _ = gpu_render(Config(fused_trace=1, max_bounces=3, integrator=0, **kw), sc, 2)
        assert np.array_equal(c.view(np.uint32), d.view(np.uint32)) and sc_.bounces == sd.bounces


def test_two_bounces_per_launch_keep_the_bits(rtx_lib, monkeypatch):
    """Light-free scenes whose mesh is tested without a tree (the reference's sphere scenes) make two bounces per
    launch of the bounce kernel: a path that goes on after the first stays in its lane.  RTX_FEW_BOUNCES=1 keeps one
    bounce per launch: same images, same counts (the second bounces are counted apart from the queue fills), with
    every camera, a bounce limit of one, two and five, and a pool smaller than a wave's worth of segments."""
    for sc in (scenes.sphere_sky(glass=True), scenes.sphere_sky(glass=False)):
        for kw in (dict(width=200, height=120, sky=1, pool_paths=1 << 13),
                   dict(width=97, height=61, sky=1, pool_paths=700, max_bounces=5, camera=1, fov=60.0),
                   dict(width=64, height=48, sky=1, pool_paths=1 << 12, max_bounces=1),
                   dict(width=64, height=48, sky=1, pool_paths=1 << 12, max_bounces=2, camera=2),
                   dict(width=80, height=50, sky=0, pool_paths=1 << 12, primary_rays=1)):
            monkeypatch.setenv("RTX_FEW_BOUNCES", "1")
            a, sa, _ = gpu_render(Config(**kw), sc, 6)
            monkeypatch.delenv("RTX_FEW_BOUNCES")
            b, sb, _ = gpu_render(Config(**kw), sc, 6)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (sc.name, kw)
            assert (sa.paths, sa.bounces, sa.nee_samples, sa.shadow_rays, sa.nan_paths) == \
                   (sb.paths, sb.bounces, sb.nee_samples, sb.shadow_rays, sb.nan_paths), (sc.name, kw)
            # (b ran kernels built for what the scene and the call are not: no boxes, no planes, no bounce limit, and
            # for the first case the open sky with nothing that emits; the general build of the same kernel:)
            monkeypatch.setenv("RTX_NO_PLAIN_KERNELS", "1")
            g, sg, _ = gpu_render(Config(**kw), sc, 6)
            monkeypatch.delenv("RTX_NO_PLAIN_KERNELS")
            assert np.array_equal(b.view(np.uint32), g.view(np.uint32)), (sc.name, kw, "general build")
            assert (sg.paths, sg.bounces) == (sb.paths, sb.bounces)


@pytest.mark.parametrize("cap", [None, "1", "2"], ids=["queue_default", "queue_1_per_thread", "queue_2_per_thread"])
def test_triangle_tests_as_tasks_of_the_block_keep_the_bits(cap, rtx_lib, monkeypatch):
    """Trees of up to 32 leaves in scenes where rays meet leaves at every turn (Cornell) have every lane test all
    leaf boxes and the block test the (ray, leaf) pairs from a queue in LDS, for bounces and light samples alike.
    RTX_NO_LEAF_TASKS keeps the per-lane walk, RTX_LEAF_TASKS_ALWAYS uses the tasks for sparse scenes too (leaves of
    one triangle, overlapping triangles with equal t, spheres in front of the mesh), RTX_LEAF_TASK_CAP shrinks the
    queue until lanes have to test their leaves themselves: same images, same counts."""
    cases = [(scenes.cornell(), 0), (scenes.cornell(lights=4), 0), (scenes.soup(30, 3, seed=2, size=0.5), 1),
             (scenes.soup(57, 0, seed=9, size=0.9), 1), (scenes.soup(9, 2, seed=4, size=0.9), 1),
             # lit meshes of up to eight triangles are tested without a tree: they must not get the task kernels
             # (whose workspace sits behind a tree staged in LDS), whatever the hook asks for
             (scenes.soup(8, 1, seed=6, size=0.9), 0), (scenes.soup(6, 0, seed=3, size=0.9), 1),
             # around the 32 leaves the task kernels take (a quad makes a leaf of two, every tenth quad comes twice):
             # 30 to 33 quads, the last ones left to the walk
             (scenes.quad_soup(27, seed=5, size=0.8), 1), (scenes.quad_soup(28, seed=7, size=0.8), 1),
             (scenes.quad_soup(29, seed=8, size=0.8), 0), (scenes.quad_soup(30, seed=2, size=0.8), 1)]
    for sc, sky in cases:
        for kw in (dict(width=200, height=120, sky=sky, pool_paths=1 << 13),
                   dict(width=97, height=61, sky=sky, pool_paths=700, max_bounces=4, camera=2)):
            monkeypatch.setenv("RTX_NO_LEAF_TASKS", "1")
            monkeypatch.delenv("RTX_LEAF_TASKS_ALWAYS", raising=False)
            monkeypatch.delenv("RTX_LEAF_TASK_CAP", raising=False)
            a, sa, _ = gpu_render(Config(**kw), sc, 4)
            monkeypatch.delenv("RTX_NO_LEAF_TASKS")
            monkeypatch.setenv("RTX_LEAF_TASKS_ALWAYS", "1")
            if cap:
                monkeypatch.setenv("RTX_LEAF_TASK_CAP", cap)
            b, sb, _ = gpu_render(Config(**kw), sc, 4)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (sc.name, kw)
            assert (sa.paths, sa.bounces, sa.nee_samples, sa.shadow_rays) == (sb.paths, sb.bounces, sb.nee_samples, sb.shadow_rays)
            # (b traced its light samples inside the bounce kernel, as a second round of tasks; with them queued
            # for k_shadow_leaf_tasks instead:)
            monkeypatch.setenv("RTX_QUEUE_LIGHT_SAMPLES", "1")
            q, sq_, _ = gpu_render(Config(**kw), sc, 4)
            monkeypatch.delenv("RTX_QUEUE_LIGHT_SAMPLES")
            assert np.array_equal(a.view(np.uint32), q.view(np.uint32)), (sc.name, kw, "queued light samples")
            assert (sa.paths, sa.bounces, sa.nee_samples, sa.shadow_rays) == (sq_.paths, sq_.bounces, sq_.nee_samples, sq_.shadow_rays)
    monkeypatch.delenv("RTX_LEAF_TASKS_ALWAYS", raising=False)
    monkeypatch.delenv("RTX_LEAF_TASK_CAP", raising=False)


def test_leaf_pairs_of_the_wide_tree_keep_the_bits(rtx_lib, monkeypatch):
    """On the wide tree a leaf that holds the two halves of a fan-tessellated quad is one record of four vertices
    (edges and normals computed in the walk); RTX_NO_LEAF_PAIRS=1 keeps one record per triangle with the normal
    stored.  Same images, same counts, for the closest-hit walk, the light samples and the drain kernel."""
    cases = [(scenes.terrain(40), 0), (scenes.quad_soup(1500, seed=3), 1), (scenes.quad_soup(400, seed=9, size=0.6), 0)]
    for sc, sky in cases:
        for kw in (dict(width=160, height=100, sky=sky, pool_paths=1 << 13),
                   dict(width=97, height=61, sky=sky, pool_paths=900, max_bounces=5, camera=1, finish_paths=1 << 20)):
            monkeypatch.setenv("RTX_NO_LEAF_PAIRS", "1")
            a, sa, ia = gpu_render(Config(**kw), sc, 5)
            monkeypatch.delenv("RTX_NO_LEAF_PAIRS")
            b, sb, _ = gpu_render(Config(**kw), sc, 5)
            assert ia.bvh_width == 4
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (sc.name, kw)
            assert (sa.paths, sa.bounces, sa.nee_samples, sa.shadow_rays, sa.nan_paths) == \
                   (sb.paths, sb.bounces, sb.nee_samples, sb.shadow_rays, sb.nan_paths), (sc.name, kw)


def test_scenes_without_emitters_skip_the_colour_state_and_keep_their_bits(oracle, rtx_lib):
    """Where no material emits and there are no lights a live path's colour is exactly 0, and the queues
    do not carry it.  An emissive material that no surface uses switches the skipping off: same image."""
    for fused in (0, 1):
        for sc in (scenes.sphere_sky(glass=True), scenes.sphere_sky(glass=False)):
            kw = dict(width=150, height=90, sky=1, pool_paths=1 << 12, fused_trace=fused)
            a, sa, _ = gpu_render(Config(**kw), sc, 6)
            lit = scenes.SceneData(sc.triangles, sc.material_index,
                                   np.concatenate([sc.materials, np.array([scenes.material((1, 1, 1), 3.0)], scenes.MATERIAL_DTYPE)]),
                                   sc.spheres, name="with_unused_emitter")
            b, sb, ib = gpu_render(Config(**kw), lit, 6)
            assert ib.num_lights == 0
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)) and sa.bounces == sb.bounces
            want, ost = oracle.OracleScene(sc, Config(**kw)).render(6)
            assert sa.paths == ost.paths and close_fraction(a, want) >= 0.98


def test_uploaded_sums_resume_a_render(rtx_lib):
    """rtx_accum_upload: the sums of 5 samples, taken to the host and put into another context's
    buffer, then 4 more samples from index 5 = 9 samples in one call."""
    sc = scenes.cornell()
    cfg = Config(width=90, height=60, pool_paths=1 << 13)
    whole, _, _ = gpu_render(cfg, sc, 9)
    with Renderer(cfg) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        r.render(acc, 5)
        sums = r.resolve(acc, 1)
    with Renderer(cfg) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        r.accum_upload(acc, sums)
        assert np.array_equal(r.resolve(acc, 1).view(np.uint32), sums.view(np.uint32))
        r.render(acc, 4, sample_base=5)
        got = r.resolve(acc, 1)
    assert np.array_equal(got.view(np.uint32), whole.view(np.uint32))


@pytest.mark.parametrize("finish_paths", [0, 4096, 1 << 20])
def test_sum_beside_the_drain_kernel_keeps_the_bits(finish_paths, rtx_lib, monkeypatch):
    """Large chunks add up most of their samples on a second stream while the drain kernel runs the
    last paths (marker in the unfinished result slots, the rest added afterwards).  Forced on for
    small renders and switched off, the sums are the same bits, whether the drain kernel takes over
    as soon as the samples are exhausted or only for the last 4096 paths."""
    split_ran = False
    for sc, sky in [(scenes.sphere_sky(glass=True), 1), (scenes.cornell(), 0), (scenes.soup(3000, 2, seed=9), 1)]:
        out = []
        for threshold in ("1", "18446744073709551615"):
            monkeypatch.setenv("RTX_OVERLAP_MIN_SAMPLES", threshold)
            kw = dict(width=128, height=96, sky=sky, pool_paths=1 << 14, finish_paths=finish_paths)
            img, st, _ = gpu_render(Config(**kw), sc, 12, calls=[dict(spp=7), dict(spp=5, sample_base=7)])
            band, _, _ = gpu_render(Config(**kw), sc, 6, calls=[dict(spp=6, row0=8, row1=40)])
            out.append((img, band, st.bounces, st.drain_launches, st.overlapped_sums))
        monkeypatch.delenv("RTX_OVERLAP_MIN_SAMPLES")
        assert np.array_equal(out[0][0].view(np.uint32), out[1][0].view(np.uint32))
        assert np.array_equal(out[0][1].view(np.uint32), out[1][1].view(np.uint32))
        assert out[0][2] == out[1][2]
        # The drain kernel runs only if a snapshot of the queue reaches the host between "samples exhausted, at
        # most finish_paths paths left" and "no path left", which depends on timing for renders this small; the
        # sum runs beside it whenever it does (forced on), never when forced off.
        assert out[0][4] == out[0][3] and out[1][4] == 0
        split_ran = split_ran or out[0][4] > 0
    if finish_paths == 1 << 20:
        assert split_ran                       # with the drain kernel taking over at once it ran for some chunk


@pytest.mark.parametrize("inside", [True, False], ids=["triangle_inside_sphere", "triangle_far_away"])
def test_drain_kernel_shortcut_inside_spheres(inside, oracle, rtx_lib):
    """The drain kernel skips the mesh for a ray that runs from inside a sphere to that sphere again
    when the host found no triangle near the sphere.  A small bright triangle INSIDE the glass sphere
    must switch the shortcut off (its light would be lost otherwise); with the triangle far away the
    shortcut is on.  Every path runs in the drain kernel (finish_paths above the sample count); the
    image must follow the oracle either way."""
    sc0 = scenes.sphere_sky(glass=True)
    c = np.array([0.0, 1.0, -1.0]) if inside else np.array([0.0, 5.0, -1.0])
    tri = np.concatenate([c + [-0.2, -0.1, 0.0], c + [0.2, -0.1, 0.05], c + [0.0, 0.25, -0.05]]).astype(np.float32)
    mats = np.concatenate([sc0.materials, np.array([scenes.material((1, 1, 1), 40.0)], scenes.MATERIAL_DTYPE)])
    sc = scenes.SceneData(np.concatenate([sc0.triangles, tri[None]]), np.concatenate([sc0.material_index, [len(mats) - 1]]),
                          mats, sc0.spheres, name="lamp_in_glass")
    cfg = Config(width=96, height=96, sky=0, pool_paths=1 << 17, finish_paths=1 << 20, nee_threshold=1000.0)
    want, ost = oracle.OracleScene(sc, cfg).render(8)
    got, st, info = gpu_render(cfg, sc, 8)
    assert info.num_lights == 0 and st.paths == ost.paths
    assert close_fraction(got, want) >= 0.98
    assert abs(luminance(got) - luminance(want)) <= 0.02 * luminance(want) + 1e-4
    if inside:
        assert luminance(want) > 0.05      # the lamp is seen through the glass: losing it would show


def test_result_buffer_chunking(rtx_lib):
    """A result buffer smaller than the call's samples: the call is rendered in chunks."""
    sc = scenes.cornell()
    big = Config(width=256, height=256, pool_paths=1 << 15)          # 768 KiB per spp
    small = Config(width=256, height=256, pool_paths=1 << 15, result_buffer_mib=2)  # 2 spp per chunk
    a, sa, _ = gpu_render(big, sc, 5)
    b, sb, _ = gpu_render(small, sc, 5)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert sa.bounces == sb.bounces and sa.paths == sb.paths


def test_max_bounces_cap(oracle, rtx_lib):
    sc = scenes.cornell()
    cfg = Config(width=64, height=64, max_bounces=3)
    want, ost = oracle.OracleScene(sc, cfg).render(2)
    got, st, _ = gpu_render(cfg, sc, 2)
    assert close_fraction(got, want) >= 0.99
    assert abs(st.bounces - ost.bounces) <= 0.005 * ost.bounces
    assert st.bounces <= 3 * st.paths


def test_camera_offset(oracle, rtx_lib):
    sc = scenes.cornell()
    cfg = Config(width=64, height=64, integrator=1)
    off = (0.25, -0.1, -0.5)
    want, _ = oracle.OracleScene(sc, cfg).render(1, cam_offset=off)
    got, _, _ = gpu_render(cfg, sc, 1, cam_offset=off)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_errors(rtx_lib):
    from raytracer_amd.capi import RtxError
    with pytest.raises(RtxError):
        Renderer(Config(width=0, height=10))
    with Renderer(Config(width=8, height=8, pool_paths=256)) as r:
        acc = r.accum_alloc()
        with pytest.raises(RtxError, match="no scene"):
            r.render(acc, 1)
        bad = scenes.cornell()
        bad.material_index[3] = 99
        with pytest.raises(RtxError, match="material index"):
            r.set_scene(bad)
        r.set_scene(scenes.cornell())
        with pytest.raises(RtxError, match="row range"):
            r.render(acc, 1, row0=5, row1=3)
