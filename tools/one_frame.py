"""One warm-up frame and one measured frame of a bench workload through rtx_render (for kernel traces).
usage: python tools/one_frame.py WORKLOAD [spp]"""
import sys

import torch

import bench
from raytracer_amd.capi import Config, Renderer

wl = bench.workloads()[sys.argv[1]]
spp = int(sys.argv[2]) if len(sys.argv) > 2 else wl["spp"]
with Renderer(Config(width=wl["width"], height=wl["height"], sky=wl["sky"])) as r:
    r.set_scene(wl["scene"]())
    acc = torch.zeros(wl["width"] * wl["height"] * 3, device="cuda")
    for _ in range(2):
        r.render(acc, spp)
    torch.cuda.synchronize()
    st = r.stats()
    print(f"launches {st.launches} rounds {st.iterations} paths {st.paths} bounces {st.bounces} stored {st.stored_paths}")
