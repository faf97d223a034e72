import sys, time; sys.path.insert(0, '.')
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
W,H=1920,1080
for name, s in (("instanced(TLAS)", scenes.instanced_scene(count=200, detail=48)), ("baked", scenes.bake_instances(scenes.instanced_scene(count=200, detail=48)))):
    for sched in (0, 1):
        pt = PathTracer(s, width=W, height=H, samplesPerPass=8, schedule=sched)
        for k in range(2): pt.OnRenderImage(100+k)
        pt.synchronize(); pt.reset_stats()
        t0=time.perf_counter()
        for k in range(4): pt.OnRenderImage(200+k)
        pt.synchronize(); dt=(time.perf_counter()-t0)/4
        print(f"{name} tris {s.tri_count} schedule {sched}: {dt*1e3:.2f} ms/pass, {pt.stats().rays/4/dt/1e6:.0f} Mrays/s", flush=True)
        pt.close()
