"""GPU tests of rtx_config.rng_streams, the opt-in generator mode for renders of more than 2^32 samples: the
reference's Pcg (Sampler.h:11-42) with an odd increment of its own per sample.  The default mode stays the
reference's generator bit for bit (tests/test_gpu_counters_and_4k.py::test_device_pcg_known_answers); here:
the starting (state, increment) pairs of a 3840x2160 frame at 520 samples per pixel are all distinct where the
default mode's starting states cannot be, paths agree with the oracle's restatement of the mode sample by sample,
and mean and variance of a Cornell render agree with the default mode's."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PCG_INCREMENT = 2891336453


def stream_pairs(*args):
    """tools/stream_pairs.py in a process of its own: it sorts the pairs with torch on the GPU, and torch has to be
    imported before the library for the two to share one HIP runtime."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "stream_pairs.py"), *map(str, args)],
                         capture_output=True, text=True, timeout=500, env=dict(os.environ, PYTHONPATH=ROOT))
    assert out.returncode == 0, out.stderr[-2000:]
    return json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])


def test_default_mode_keeps_the_reference_increment_and_the_hashed_start(oracle, rtx_lib):
    got = stream_pairs(64, 48, 10, 0, 5)
    assert got["distinct_increments_in_first_million"] == 1 and got["first_increment"] == PCG_INCREMENT
    for k, v in got["at"].items():
        pixel, sample = int(k) % (64 * 48), int(k) // (64 * 48)
        assert v & 0xffffffff == oracle.path_seed(pixel, sample, 0x5EED) and v >> 32 == PCG_INCREMENT


def test_4k_frame_of_520_samples_starts_from_distinct_pairs(rtx_lib):
    """3840 x 2160 x 520 = 4.3e9 samples, more than the 2^32 states of the reference's one cycle: with streams every
    sample starts from its own (state, increment); the default mode's starting states repeat already on a
    1024 x 1024 x 64 slice (6.7e7 samples: about n^2 / 2^33 = 5e5 repeats)."""
    got = stream_pairs(3840, 2160, 520, 1)
    assert got["samples"] == 3840 * 2160 * 520 and got["all_increments_odd"]
    assert got["distinct_increments_in_first_million"] > 1000000   # (2^20 samples, all on cycles of their own)
    assert got["repeats"] <= 2                                      # (n^2 / 2^65 = 0.5 expected)
    assert stream_pairs(1024, 1024, 64, 0)["repeats"] > 100000


def test_stream_mode_follows_the_oracle_sample_by_sample(oracle, rtx_lib):
    """One sample per pixel, so that every pixel is one path: the HIP path in stream mode against the oracle's
    restatement of the same mode (tolerance as tests/test_gpu_render.py: sinf/cosf differ in the last ulp), same
    path and bounce counts within the same slack; and the image is not the default mode's."""
    for sc, sky in [(scenes.cornell(), 0), (scenes.sphere_sky(glass=True), 1), (scenes.terrain(12), 0)]:
        kw = dict(width=160, height=120, sky=sky, pool_paths=1 << 13)
        imgs = {}
        for mode in (0, 1):
            cfg = Config(rng_streams=mode, **kw)
            with Renderer(cfg) as r:
                r.set_scene(sc)
                acc = r.accum_alloc()
                r.render(acc, 1, sample_base=3, seed=77)
                imgs[mode] = r.resolve(acc, 1)
                st = r.stats()
            want, ost = oracle.OracleScene(sc, cfg).render(1, sample_base=3, seed=77)
            close = np.all(np.abs(imgs[mode] - want) <= 1e-5 + 1e-3 * np.abs(want), axis=-1).mean()
            assert close >= 0.98, (sc.name, mode, close)
            assert st.paths == ost.paths and abs(st.bounces - ost.bounces) <= 0.005 * ost.bounces
        assert not np.array_equal(imgs[0], imgs[1])


def test_mean_and_variance_agree_with_the_default_mode(rtx_lib):
    """Cornell at 128 x 128: eight independent renders of 32 samples per pixel in each mode.  The means of all 256
    samples agree within the noise two halves of one mode show, mean luminance within 1 %, and the per-pixel
    variance between the eight renders is the same within 15 %."""
    sc = scenes.cornell()
    stats = {}
    for mode in (0, 1):
        with Renderer(Config(width=128, height=128, rng_streams=mode, pool_paths=1 << 14)) as r:
            r.set_scene(sc)
            runs = []
            for k in range(8):
                acc = r.accum_alloc()
                r.render(acc, 32, sample_base=32 * k, seed=0xC0FFEE)
                runs.append(r.resolve(acc, 32).astype(np.float64))
        runs = np.array(runs)
        stats[mode] = (runs.mean(0), runs.var(0, ddof=1).mean(), runs)
    m0, v0, r0 = stats[0]
    m1, v1, _ = stats[1]
    within = np.sqrt(np.mean((r0[:4].mean(0) - r0[4:].mean(0)) ** 2))     # two halves of 128 spp: sqrt(2) x the noise of 256
    between = np.sqrt(np.mean((m0 - m1) ** 2))
    assert between <= 1.3 * within, (between, within)
    assert abs(m0.mean() / m1.mean() - 1.0) < 0.01
    assert abs(v0 / v1 - 1.0) < 0.15, (v0, v1)
