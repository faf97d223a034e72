import os, sys, time; sys.path.insert(0, '.')
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
W,H=1920,1080
which = sys.argv[1:] or ["tlas", "baked"]
for name in which:
    s = scenes.instanced_scene(count=200, detail=48)
    if name == "baked": s = scenes.bake_instances(s)
    for sched in (1,):
        pt = PathTracer(s, width=W, height=H, samplesPerPass=8, schedule=sched)
        if os.environ.get("PT_SETS"): pt.set_passes_in_flight(int(os.environ["PT_SETS"]))
        for k in range(3): pt.OnRenderImage(100+k)
        pt.synchronize(); pt.reset_stats()
        t0=time.perf_counter()
        for k in range(8): pt.OnRenderImage(200+k)
        pt.synchronize(); dt=(time.perf_counter()-t0)/8
        print(f"{name} tris {s.tri_count} schedule {sched}: {dt*1e3:.2f} ms/pass, {pt.stats().rays/8/dt/1e6:.0f} Mrays/s", flush=True)
        pt.close()
