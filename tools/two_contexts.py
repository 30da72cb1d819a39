"""Do two renders that run side by side fill the wave slots a single one leaves empty at the ends of its launches?
Two contexts, each driven by a host thread of its own, each rendering FRAMES frames of a bench workload at half the
queue pool; against one context rendering 2 x FRAMES frames at the full pool.
usage (GPU box): PYTHONPATH=. python tools/two_contexts.py [WORKLOAD] [FRAMES]"""
import sys
import threading
import time

import torch

import bench
from raytracer_amd.capi import Config, Renderer

name = sys.argv[1] if len(sys.argv) > 1 else "sample2_glass"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 10
wl = bench.workloads()[name]
W, H, spp = wl["width"], wl["height"], wl["spp"]


def run(r, acc, n):
    for _ in range(n):
        r.render_async(acc, spp)
    r.wait()


def make(pool):
    r = Renderer(Config(width=W, height=H, sky=wl["sky"], pool_paths=pool))
    r.set_scene(wl["scene"]())
    acc = torch.zeros(W * H * 3, device="cuda")
    run(r, acc, 2)
    torch.cuda.synchronize()
    return r, acc


full = 1 << 24
for pool in (full, full // 2):
    r, acc = make(pool)
    t0 = time.perf_counter()
    run(r, acc, 2 * frames)
    torch.cuda.synchronize()
    one = (time.perf_counter() - t0) / (2 * frames)
    print(f"{name}: one context, pool {pool >> 20} Mi paths: {one * 1e3:.3f} ms per frame")
    r.close()
for pool in (full, full // 2):
    a, b = make(pool), make(pool)
    t0 = time.perf_counter()
    th = [threading.Thread(target=run, args=(x[0], x[1], frames)) for x in (a, b)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    torch.cuda.synchronize()
    two = (time.perf_counter() - t0) / (2 * frames)
    print(f"{name}: two contexts side by side, pool {pool >> 20} Mi paths each: {two * 1e3:.3f} ms per frame")
    a[0].close()
    b[0].close()
