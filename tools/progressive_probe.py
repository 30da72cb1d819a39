"""Per-call cost of the progressive loop (Raytracer.cpp:307-331 renders SAMPLES_PER_PIXEL = 1
per pass): many small rtx_render calls into one sum buffer."""
import sys, time
sys.path.insert(0, '.')
from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer
import torch

for name, sc, sky in [("cornell", scenes.cornell(), 0), ("sample2", scenes.sphere_sky(glass=True), 1)]:
    cfg = Config(width=1024, height=1024, sky=sky)
    with Renderer(cfg) as r:
        r.set_scene(sc)
        acc = r.accum_alloc()
        for spp in (1, 4, 16):
            r.render(acc, spp)
            torch.cuda.synchronize()
            n = 50
            t0 = time.perf_counter()
            for i in range(n):
                r.render(acc, spp, sample_base=(i + 1) * spp)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / n
            print(f"{name}: {spp} spp per call: {dt*1e3:.3f} ms per call, {1024*1024*spp/dt/1e6:.1f} Mpaths/s")
        free, total = torch.cuda.mem_get_info()
        print(f"   device memory in use {(total-free)/2**30:.2f} GiB")
