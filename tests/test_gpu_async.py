"""rtx_render_async: frames left in flight on two sets of queues give the sums rtx_render gives, bit
for bit, whatever the interleaving: frames that add to one buffer, to two buffers in turn, multi-chunk
frames, a scene change and a synchronous render in between; statistics are those of the same frames."""
import numpy as np
import pytest

from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer

pytestmark = pytest.mark.gpu


def bits(a):
    return a.view(np.uint32)


@pytest.mark.parametrize("scene,sky", [("cornell", 0), ("glass", 1), ("soup", 1)])
def test_frames_in_flight_add_up_to_the_same_bits(scene, sky, rtx_lib):
    sc = {"cornell": scenes.cornell, "glass": lambda: scenes.sphere_sky(glass=True),
          "soup": lambda: scenes.soup(3000, 3, seed=2)}[scene]()
    cfg = Config(width=160, height=96, sky=sky, pool_paths=1 << 13, finish_paths=256)
    frames = [(3, 0), (2, 3), (4, 5), (1, 9), (2, 10)]            # (spp, sample_base)
    with Renderer(cfg) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        for spp, base in frames:
            r.render(acc, spp, sample_base=base)
        want = r.resolve(acc, 12)
        st_want = r.stats()
        r.reset_stats()
        # the same frames without waiting in between, all into one buffer
        acc2 = r.accum_alloc()
        for spp, base in frames:
            r.render_async(acc2, spp, sample_base=base)
        r.wait()
        got = r.resolve(acc2, 12)
        st = r.stats()
        assert np.array_equal(bits(got), bits(want))
        assert (st.paths, st.bounces, st.shadow_rays, st.nee_samples) == \
               (st_want.paths, st_want.bounces, st_want.shadow_rays, st_want.nee_samples)
        # into two buffers in turn, waiting for one buffer only, with a synchronous frame in between
        a, b = r.accum_alloc(), r.accum_alloc()
        r.render_async(a, 3, sample_base=0)
        r.render_async(b, 2, sample_base=3)
        r.wait(a)
        r.render_async(a, 4, sample_base=5)
        r.render(b, 1, sample_base=9)            # waits for everything in flight first
        r.render_async(b, 2, sample_base=10)
        ia, ib = r.resolve(a, 1), r.resolve(b, 1)  # resolve waits for its buffer's frames
        with Renderer(cfg) as ref:
            ref.set_scene(sc)
            ra, rb = ref.accum_alloc(), ref.accum_alloc()
            ref.render(ra, 3, sample_base=0)
            ref.render(ra, 4, sample_base=5)
            ref.render(rb, 2, sample_base=3)
            ref.render(rb, 1, sample_base=9)
            ref.render(rb, 2, sample_base=10)
            assert np.array_equal(bits(ia), bits(ref.resolve(ra, 1)))
            assert np.array_equal(bits(ib), bits(ref.resolve(rb, 1)))


def test_frames_in_flight_with_chunks_and_a_scene_change(rtx_lib):
    """A frame of several chunks (small result buffer) leaves only its last chunk in flight; a new
    scene waits for the frames and reaches both sets."""
    cfg = Config(width=128, height=64, sky=1, pool_paths=1 << 12, finish_paths=64, result_buffer_mib=1)
    s1, s2 = scenes.cornell(), scenes.soup(150, 3, seed=2)
    with Renderer(cfg) as r:
        out = {}
        for mode in ("sync", "async"):
            render = r.render if mode == "sync" else r.render_async
            images = []
            for sc in (s1, s2, s1):
                r.set_scene(sc)
                acc = r.accum_alloc()
                for k in range(3):
                    render(acc, 15, sample_base=15 * k)     # 15 spp * 8192 px * 12 B > 1 MiB: several chunks
                images.append(r.resolve(acc, 45))
            out[mode] = images
        for x, y in zip(out["sync"], out["async"]):
            assert np.array_equal(bits(x), bits(y))


@pytest.mark.parametrize("seed", range(6))
def test_random_interleavings_of_frames_in_flight(seed, rtx_lib):
    """Random programmes of asynchronous and synchronous frames into three buffers, with waits on one
    buffer, waits on all, ray queries (which share the traversal stacks' spill columns with the frames)
    and statistics reads thrown in: every buffer ends with the bits of the same frames rendered one
    after the other, and the path counts add up."""
    rng = np.random.default_rng(1000 + seed)
    sc = [scenes.cornell(), scenes.sphere_sky(glass=True), scenes.soup(3000, 2, seed=3)][seed % 3]
    cfg = Config(width=96, height=64, sky=int(seed % 3 != 0), pool_paths=int(rng.choice([1 << 11, 1 << 13])),
                 finish_paths=int(rng.choice([0, 64, 1 << 20])))
    rays = scenes.random_rays(500, seed=seed)
    with Renderer(cfg) as r, Renderer(cfg) as ref:
        r.set_scene(sc)
        ref.set_scene(sc)
        bufs = [r.accum_alloc() for _ in range(3)]
        refs = [ref.accum_alloc() for _ in range(3)]
        done = [0, 0, 0]               # samples each buffer holds
        for step in range(int(rng.integers(8, 16))):
            b = int(rng.integers(0, 3))
            spp = int(rng.integers(1, 4))
            op = rng.random()
            if op < 0.6:
                r.render_async(bufs[b], spp, sample_base=done[b])
            elif op < 0.8:
                r.render(bufs[b], spp, sample_base=done[b])
            else:
                spp = 0
                what = int(rng.integers(0, 4))
                if what == 0:
                    r.wait(bufs[b])
                elif what == 1:
                    r.wait()
                elif what == 2:
                    r.trace_closest(rays)
                else:
                    r.stats()
            if spp:
                ref.render(refs[b], spp, sample_base=done[b])
                done[b] += spp
        for b in range(3):
            assert np.array_equal(bits(r.resolve(bufs[b], 1)), bits(ref.resolve(refs[b], 1))), (seed, b)
        st, st_ref = r.stats(), ref.stats()
        assert (st.paths, st.bounces, st.shadow_rays) == (st_ref.paths, st_ref.bounces, st_ref.shadow_rays)
