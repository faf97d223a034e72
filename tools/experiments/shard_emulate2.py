import sys, time; sys.path.insert(0, '.')
import numpy as np
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
s = scenes.make_scene("sponza", tex_size=1024)
W,H=1920,1080
def seed_of(k): return (0x12345678 + k*0x9E3779B9) & 0xFFFFFFFF
K=16
for world in (2, 4, 8):
    for sched in (1,):
        pt = PathTracer(s, width=W, height=H, samplesPerPass=8, rank=0, world_size=world, schedule=sched)
        def run(k0, n):
            for k in range(k0, k0+n):
                p = scenes.frame_params(s, W, H, spp=8, current_sample=8*k, seed=seed_of(k))
                pt.render_pass(p); pt.flip()
            pt.synchronize()
        run(0, 2)
        pt.reset_stats(); t0=time.perf_counter(); run(2, K); wall=(time.perf_counter()-t0)*1e3/K
        rays = pt.stats().rays / K
        print(f"world {world} sched {sched}: rank-0 share {rays/1e6:.1f} Mrays/pass, {wall:.2f} ms/pass pipelined -> {rays/wall/1e3:.0f} Mrays/s per GPU, x{world} = {rays/wall/1e3*world:.0f} aggregate (no reduce)", flush=True)
        pt.close()
