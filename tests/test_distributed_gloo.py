"""The N > 1 path on CPU: two gloo ranks render disjoint sample ranges (and row bands),
sum their buffers onto rank 0 with raytracer_amd.dist.reduce_sum_to_root, and the result
equals one process rendering everything.  The CPU oracle stands in for the GPU renderer
here (same sample-range and row-band arguments as rtx_render)."""
import os
import socket

import numpy as np
import pytest

from raytracer_amd import dist as rdist


def test_split_range_partitions_exactly():
    for n in [0, 1, 7, 256, 1000]:
        for parts in [1, 2, 3, 8]:
            spans = [rdist.split_range(n, parts, i) for i in range(parts)]
            assert spans[0][0] == 0 and sum(c for _, c in spans) == n
            for (s0, c0), (s1, _) in zip(spans, spans[1:]):
                assert s0 + c0 == s1
            assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
    assert rdist.sample_range(256, 3, 8, sample_base=100) == (196, 32)
    assert rdist.weak_sample_range(256, 3) == (768, 256)
    # 4 tile groups x 2 sample replicas over 8 ranks (the 4K configuration)
    got = [rdist.tile_and_samples(2160, 4096, r, 8, 4) for r in range(8)]
    assert got[0] == (0, 540, 0, 2048) and got[1] == (0, 540, 2048, 2048) and got[7] == (1620, 2160, 2048, 2048)
    with pytest.raises(ValueError):
        rdist.tile_and_samples(100, 8, 0, 6, 4)


def _worker(rank, world, port, mode, out_path):
    import torch
    import torch.distributed as dist
    from raytracer_amd import scenes
    from raytracer_amd.capi import Config
    from oracle import oracle_py
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    cfg = Config(width=40, height=24)
    osc = oracle_py.OracleScene(scenes.cornell(), cfg)
    acc = np.zeros((cfg.height, cfg.width, 3), np.float32)
    if mode == "samples":
        base, count = rdist.sample_range(6, rank, world)
        osc.render(count, sample_base=base, accum=acc, nthreads=2)
    else:
        r0, r1, base, count = rdist.tile_and_samples(cfg.height, 6, rank, world, tile_groups=2)
        osc.render(count, sample_base=base, row0=r0, row1=r1, accum=acc, nthreads=2)
    t = torch.from_numpy(acc)
    rdist.reduce_sum_to_root(t, dst=0)
    dist.barrier()
    if rank == 0:
        np.save(out_path, t.numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["samples", "tiles"])
def test_two_ranks_add_up_to_one_process(mode, tmp_path, oracle):
    import torch.multiprocessing as mp
    from raytracer_amd import scenes
    from raytracer_amd.capi import Config
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "sum.npy")
    mp.spawn(_worker, args=(2, port, mode, out), nprocs=2, join=True)
    got = np.load(out)
    cfg = Config(width=40, height=24)
    want, _ = oracle.OracleScene(scenes.cornell(), cfg).render(6)
    if mode == "tiles":
        # disjoint bands, one rank each: exact
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    else:
        # (s0+s1+s2) + (s3+s4+s5) vs sequential: float association differs
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)


def test_spawn_local_ranks(tmp_path):
    """bench.py --gpus N starts its own ranks with spawn_local_ranks when it was not launched by
    torch.distributed.run: every rank gets its rendezvous environment, only rank 0's stdout
    comes through, and a failing rank makes the whole launch fail."""
    import json
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    child = os.path.join(here, "native", "rank_echo.py")
    code = ("import sys; sys.path.insert(0, %r); from raytracer_amd import dist; "
            "sys.exit(dist.spawn_local_ranks(%r, sys.argv[1:], 2))" % (os.path.dirname(here), child))
    ok = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120)
    assert ok.returncode == 0, ok.stderr
    assert "noise from another rank" not in ok.stdout
    lines = [l for l in ok.stdout.splitlines() if l.startswith("{")]  # (gloo itself prints a connection note)
    assert len(lines) == 1
    assert json.loads(lines[0]) == {"world": 2, "sum": 3.0, "local_rank": "0"}
    bad = subprocess.run([sys.executable, "-c", code, "--fail-rank", "1"], capture_output=True, text=True, timeout=120)
    assert bad.returncode == 7
