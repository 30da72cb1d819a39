"""GPU tests: the device's generator against the reference's known answers, the NaN and
truncation counters, recovery after a failed allocation, and the 3840x2160 frame of
BASELINE configs[4] (row bands + sample ranges, ordinal chunking)."""
import json
import os

import numpy as np
import pytest

from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer, RtxError

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def test_device_pcg_known_answers(rtx_lib):
    """The device functions pcg_next / pcg_uniform / pcg_hash against the vectors generated from
    the reference's Sampler.h compiled unchanged (tests/golden/pcg_kat.json): bit-exact."""
    kat = json.load(open(os.path.join(HERE, "golden", "pcg_kat.json")))
    with Renderer(Config(width=8, height=8, pool_paths=1 << 10)) as r:
        for seed, v in kat["seeds"].items():
            want_rand = np.array(v["rand"], np.uint32)
            rnd, uni, _ = r.debug_pcg(int(seed), len(want_rand))
            assert np.array_equal(rnd, want_rand), seed
            if "uniform_bits" in v:
                want_u = np.array(v["uniform_bits"], np.uint32)
                assert np.array_equal(uni.view(np.uint32)[:len(want_u)], want_u), seed
            # Uniform() = min(0x1.fffffep-1, rand * 2^-32), Sampler.h:39-41
            expect = np.minimum(np.float32(float.fromhex('0x1.fffffep-1')), want_rand.astype(np.float32) * np.float32(2.0 ** -32))
            assert np.array_equal(uni.view(np.uint32), expect.view(np.uint32)), seed
        for x, h in kat["hash"].items():
            _, _, hsh = r.debug_pcg(int(x), 1)
            assert int(hsh[0]) == h, x
        # The survey's independent known answers (seed 0: 0, 129708002, 582399676, then U = 0.234235808)
        rnd, _, _ = r.debug_pcg(0, 3)
        assert rnd.tolist() == [0, 129708002, 582399676]
        rnd, _, _ = r.debug_pcg(12345, 3)
        assert rnd.tolist() == [985940161, 4099845390, 430018922]


def test_nan_and_truncation_counters_match_the_oracle(oracle, rtx_lib):
    """rtx_stats.nan_paths counts finished paths with a non-finite colour, as the oracle does;
    trig-free scenes agree exactly, path-traced ones produce none on either side here."""
    for sc, sky, integ in [(scenes.cornell(), 0, 0), (scenes.sphere_sky(glass=True), 1, 0), (scenes.cornell(), 0, 1)]:
        cfg = Config(width=96, height=96, sky=sky, integrator=integ, pool_paths=1 << 12)
        _, ost = oracle.OracleScene(sc, cfg).render(4)
        with Renderer(cfg) as r:
            r.set_scene(sc)
            acc = r.accum_alloc()
            r.render(acc, 4)
            img = r.resolve(acc, 4)
            st = r.stats()
        assert st.truncated_paths == 0
        assert st.nan_paths == ost.nan_paths == 0
        assert np.isfinite(img).all()
    # A scene that does produce non-finite colours: a NaN vertex makes every hit test of that
    # triangle false, and an infinite emission makes inf * 0 = NaN for black-attenuation paths;
    # simpler and deterministic: a material with infinite emission seen directly.
    sc = scenes.cornell()
    sc.materials["emission"][sc.materials["emission"] > 1.0] = np.inf
    cfg = Config(width=64, height=64, integrator=0, pool_paths=1 << 12, nee_threshold=0.05)
    _, ost = oracle.OracleScene(sc, cfg).render(1)
    with Renderer(cfg) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        r.render(acc, 1)
        st = r.stats()
    assert ost.nan_paths > 0
    assert abs(int(st.nan_paths) - int(ost.nan_paths)) <= max(2, 0.01 * ost.nan_paths), (st.nan_paths, ost.nan_paths)


def test_render_after_a_failed_result_buffer_allocation(rtx_lib):
    """A result buffer the device cannot hold fails with a status and leaves the context usable:
    once memory is free again the same context renders the bits of a fresh one.  (The buffer
    is 12 bytes per sample of a chunk, at most 2^32 samples = 51.5 GB; the test first fills
    the device with ballast so that this does not fit.)"""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so.7")  # the runtime the library is linked against (already loaded)
    sc = scenes.cornell()
    W, H, spp = 256, 256, 8
    with Renderer(Config(width=W, height=H, pool_paths=1 << 14)) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        r.render(acc, spp)
        want = r.resolve(acc, 1)
    with Renderer(Config(width=W, height=H, pool_paths=1 << 14, result_buffer_mib=64 * 1024)) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        free, total = ctypes.c_size_t(), ctypes.c_size_t()
        assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
        ballast = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(ballast), ctypes.c_size_t(max(free.value - (16 << 30), 1 << 20))) == 0
        try:
            with pytest.raises(RtxError) as e:
                r.render(acc, 70000)  # one chunk of 65535 spp * 256*256 * 12 B = 51.5 GB
            assert e.value.status == -4, e.value  # RTX_ERR_OUT_OF_MEMORY
        finally:
            assert hip.hipFree(ballast) == 0
        r.accum_clear(acc)
        # The failed call must not have left a "large enough" null buffer behind.
        r.render(acc, spp)
        got = r.resolve(acc, 1)
        assert r.stats().paths == W * H * spp
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))


def test_set_scene_failure_leaves_no_scene(rtx_lib):
    """rtx_set_scene rejects bad input without touching the current scene, and a context whose
    scene upload failed reports RTX_ERR_NO_SCENE instead of rendering from freed memory."""
    sc = scenes.cornell()
    with Renderer(Config(width=32, height=32, pool_paths=1 << 10)) as r:
        r.set_scene(sc)
        bad = scenes.SceneData(sc.triangles, sc.material_index + 1000, sc.materials, sc.spheres)
        with pytest.raises(RtxError) as e:
            r.set_scene(bad)
        assert e.value.status == -1
        acc = r.accum_alloc()
        r.render(acc, 1)  # the earlier scene is still there
        assert np.isfinite(r.resolve(acc, 1)).all()


W4K, H4K = 3840, 2160


def test_4k_frame_bands_ranges_and_ordinal_chunks_bit_identical(rtx_lib):
    """BASELINE configs[4]'s frame (3840x2160, Cornell with NEE/MIS): rendered whole, as four row
    bands, and as two sample ranges -> identical bits, finite, equal bounce totals."""
    sc = scenes.cornell()
    spp = 4
    cfg = Config(width=W4K, height=H4K)
    with Renderer(cfg) as r:
        r.set_scene(sc)
        whole = r.accum_alloc()
        r.render(whole, spp)
        a = r.resolve(whole, 1)
        st_whole = r.stats()
        r.reset_stats()
        bands = r.accum_alloc()
        for b in range(4):
            r.render(bands, spp, row0=b * H4K // 4, row1=(b + 1) * H4K // 4)
        b_img = r.resolve(bands, 1)
        st_bands = r.stats()
        r.reset_stats()
        ranges = r.accum_alloc()
        r.render(ranges, 1, sample_base=0)
        r.render(ranges, spp - 1, sample_base=1)
        c_img = r.resolve(ranges, 1)
        st_ranges = r.stats()
    assert np.isfinite(a).all()
    assert np.array_equal(a.view(np.uint32), b_img.view(np.uint32))
    assert np.array_equal(a.view(np.uint32), c_img.view(np.uint32))
    assert st_whole.paths == st_bands.paths == st_ranges.paths == W4K * H4K * spp
    assert st_whole.bounces == st_bands.bounces == st_ranges.bounces
    assert st_whole.shadow_rays == st_bands.shadow_rays == st_ranges.shadow_rays
    assert st_whole.nan_paths == 0 and st_whole.truncated_paths == 0


def test_4k_call_beyond_the_32_bit_ordinal_limit(rtx_lib):
    """Sample ordinals inside a chunk are 32-bit: a 4K call of more than 0xffffffff / (3840*2160)
    = 517 samples per pixel must be cut into chunks by that limit, not by the result-buffer
    budget.  With a budget large enough for 517 spp in one chunk (51 GB), 520 spp render as
    517 + 3 and give the bits of the same call cut by the default budget (chunks of 86 spp).
    The normals integrator keeps it short (one bounce per path)."""
    sc = scenes.cornell()
    spp = 520
    limit = 0xffffffff // (W4K * H4K)
    assert spp > limit == 517
    imgs = []
    for mib in (0, 56 * 1024):
        cfg = Config(width=W4K, height=H4K, integrator=1, result_buffer_mib=mib)
        with Renderer(cfg) as r:
            r.set_scene(sc)
            acc = r.accum_alloc()
            r.render(acc, spp)
            imgs.append(r.resolve(acc, 1))
            st = r.stats()
            assert st.paths == W4K * H4K * spp
    assert np.isfinite(imgs[0]).all()
    assert np.array_equal(imgs[0].view(np.uint32), imgs[1].view(np.uint32))
