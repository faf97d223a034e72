import sys, os; sys.path.insert(0, '.')
import numpy as np
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
s = scenes.material_zoo()
w, h, spp = 112, 80, 2
ref = {}
for rank in range(3):
    pt = PathTracer(s, width=w, height=h, samplesPerPass=spp, rank=rank, world_size=3, schedule=2)
    p = pt.params(seed=1234); pt.render_pass(p); ref[rank] = pt.readback(); pt.close()
for rank in range(3):
    pt = PathTracer(s, width=w, height=h, samplesPerPass=spp, rank=rank, world_size=3, schedule=1)
    p = pt.params(seed=1234); pt.render_pass(p); a = pt.readback(); pt.close()
    bad = (a.view(np.uint32) != ref[rank].view(np.uint32)).any(-1)
    ys, xs = np.nonzero(bad)
    print("rank", rank, "bad pixels", bad.sum(), "blocks", sorted(set(zip((xs//16).tolist(), (ys//16).tolist()))))
