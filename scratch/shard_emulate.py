import sys, time; sys.path.insert(0, '.')
import numpy as np
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
s = scenes.make_scene("sponza", tex_size=1024)
W,H=1920,1080
def seed_of(k): return (0x12345678 + k*0x9E3779B9) & 0xFFFFFFFF
for world in (1, 2, 4, 8):
    for sched in (0, 1):
        pt = PathTracer(s, width=W, height=H, samplesPerPass=8, rank=0, world_size=world, schedule=sched)
        pt.set_profiling(True)
        ms = []
        for k in range(4):
            p = scenes.frame_params(s, W, H, spp=8, current_sample=0, seed=seed_of(k))
            pt.reset_timings(); pt.reset_stats(); t0=time.perf_counter(); pt.render_pass(p); pt.synchronize(); wall=(time.perf_counter()-t0)*1e3
            ms.append((pt.timings().kernelMsLast, wall))
        rays = pt.stats().rays
        best = min(m[0] for m in ms[1:]); bw = min(m[1] for m in ms[1:])
        print(f"world {world} sched {sched}: rank-0 share {rays/1e6:.1f} Mrays, device {best:.2f} ms, wall {bw:.2f} ms -> {rays/best/1e3:.0f} Mrays/s per GPU, x{world} = {rays/bw/1e3*world:.0f} aggregate (wall)", flush=True)
        pt.close()
