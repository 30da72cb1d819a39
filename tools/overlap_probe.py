"""Would two half-size pools on two streams overlap the VALU-bound traversal of one with the
HBM-bound shading of the other?  Two contexts, two host threads, half the samples each,
against one context rendering all of them."""
import sys, time, threading
sys.path.insert(0, '.')
import torch
from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer

def run(name, sc, sky, spp=256):
    cfg1 = Config(width=1024, height=1024, sky=sky)
    with Renderer(cfg1) as r:
        r.set_scene(sc)
        acc = torch.zeros(1024 * 1024 * 3, device="cuda")
        r.render(acc, spp)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            r.render(acc, spp)
        torch.cuda.synchronize()
        one = (time.perf_counter() - t0) / 3
    for pool in (1 << 23, 1 << 24):
        cfg2 = Config(width=1024, height=1024, sky=sky, pool_paths=pool)
        rs = [Renderer(cfg2), Renderer(cfg2)]
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        accs = [torch.zeros(1024 * 1024 * 3, device="cuda") for _ in range(2)]
        for r in rs:
            r.set_scene(sc)
        def work(i, reps):
            for _ in range(reps):
                rs[i].render(accs[i], spp // 2, sample_base=i * (spp // 2), stream=streams[i].cuda_stream)
        for i in range(2):
            work(i, 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        th = [threading.Thread(target=work, args=(i, 3)) for i in range(2)]
        for t in th: t.start()
        for t in th: t.join()
        torch.cuda.synchronize()
        two = (time.perf_counter() - t0) / 3
        print(f"{name}: one context {one*1e3:.2f} ms; two contexts (pool {pool>>20} Mi each) on two streams {two*1e3:.2f} ms")
        for r in rs: r.close()

run("sample2", scenes.sphere_sky(glass=True), 1)
run("cornell", scenes.cornell(), 0)
