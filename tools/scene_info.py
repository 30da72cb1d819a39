import sys; sys.path.insert(0,'/root/repo')
from raytracer_amd import scenes
from raytracer_amd.capi import Config, Renderer
import time
for name, sc in [("terrain707", scenes.terrain(707)), ("soup100k", scenes.soup(100000,0)), ("cornell", scenes.cornell())]:
    with Renderer(Config(width=64,height=64,pool_paths=4096)) as r:
        t=time.time(); i=r.set_scene(sc); dt=time.time()-t
        print(name, "nodes",i.num_nodes,"depth",i.max_depth,"stack",i.stack_entries,"build_ms",round(i.build_ms), "set_scene_s", round(dt,2))
