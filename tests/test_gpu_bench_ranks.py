"""GPU tests of the multi-rank plumbing that a one-GPU box can exercise: bench.py started as a plain
command with --gpus 2 spawns its own two ranks (both on the one GPU, buffers summed through gloo),
and the library's RCCL exchange step works with a single rank through the Python binding."""
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


def run_bench(*args, env=None):
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                         timeout=280, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout
    return json.loads(lines[0])


def test_bench_starts_its_own_ranks(rtx_lib):
    """`python bench.py --gpus 2` without torch.distributed.run: two child ranks, one JSON line from
    rank 0, twice the paths of the one-rank run (weak scaling: disjoint sample ranges per rank)."""
    common = ["--workload", "sample1_diffuse", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-profile"]
    one = run_bench(*common)
    env = dict(os.environ, RTX_BENCH_SAME_GPU="1")
    env.pop("WORLD_SIZE", None)
    two = run_bench(*common, "--gpus", "2", "--backend", "gloo", env=env)
    assert one["n_gpus"] == 1 and two["n_gpus"] == 2 and two["scaling"] == "weak"
    paths_one = one["mpaths_per_s"] * one["ms_per_step"]
    paths_two = two["mpaths_per_s"] * two["ms_per_step"]
    assert abs(paths_two / paths_one - 2.0) < 0.01
    assert two["value"] > 0 and two["cpu_baseline"] is None


def test_default_bench_line_carries_time_to_image_and_a_strong_leg(rtx_lib):
    """The default workload's line: with one rank a numeric time_to_image_ms (frames rendered and waited for
    one at a time); with two ranks (both on the one GPU, gloo) the weak value plus a `strong` leg in which the
    256 spp are split over the ranks and summed per frame."""
    common = ["--steps", "2", "--warmup", "1", "--no-cpu", "--no-profile"]
    env = dict(os.environ, RTX_BENCH_SAME_GPU="1")
    env.pop("WORLD_SIZE", None)
    two = run_bench(*common, "--gpus", "2", "--backend", "gloo", env=env)
    assert two["n_gpus"] == 2 and two["scaling"] == "weak"
    assert two["strong"]["spp_per_gpu"] == 128 and two["strong"]["time_to_image_ms"] > 0
    assert two["time_to_image_ms"] == two["strong"]["time_to_image_ms"]
    assert two["time_to_image"]["frames"]["n"] == 5


def test_exchange_step_with_one_rank_through_the_binding(rtx_lib):
    """rtx_comm_create_id / init / reduce / destroy with a communicator of one rank: the reduce is
    stream-ordered behind the render and leaves the root's sums as they are."""
    sc = scenes.cornell()
    with Renderer(Config(width=64, height=48, pool_paths=1 << 12)) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        r.render(acc, 3)
        want = r.resolve(acc, 1)
        comm_id = Renderer.comm_create_id()
        assert len(comm_id) == 128 and any(comm_id)
        r.comm_init(comm_id, 0, 1)
        r.accum_clear(acc)
        r.render(acc, 3)
        r.comm_reduce(acc, root=0)
        r.stream_synchronize()
        got = r.resolve(acc, 1)
        r.comm_destroy()
        r.comm_destroy()  # a second call is a no-op
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
