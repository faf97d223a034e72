import sys, time, os; sys.path.insert(0, '.')
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
W,H=1920,1080
for name in ("tlas", "baked"):
    s = scenes.instanced_scene(count=200, detail=48)
    if name == "baked": s = scenes.bake_instances(s)
    pt = PathTracer(s, width=W, height=H, samplesPerPass=8, schedule=1)
    for k in range(14): pt.OnRenderImage(100+k)
    pt.synchronize(); pt.reset_stats()
    t0=time.perf_counter()
    for k in range(12): pt.OnRenderImage(200+k)
    pt.synchronize(); dt=(time.perf_counter()-t0)/12
    rate = pt.stats().rays/12/dt/1e6
    pt.set_stats_level(1); pt.reset_stats(); pt.OnRenderImage(300); pt.synchronize()
    st = pt.stats().as_dict(); rays = st['closestHitRays'] + st['shadowRays']
    print(f"{name} tris {s.tri_count}: {dt*1e3:.2f} ms/pass, {rate:.0f} Mrays/s; per ray: nodes {st['nodeVisits']/rays:.2f}, tris {st['triTests']/rays:.2f}, tlas nodes {st['tlasNodeVisits']/rays:.2f}, instances {st['instanceVisits']/rays:.2f}; rays {rays/1e6:.1f} M, maxStack {st['maxStackDepth']}", flush=True)
    pt.close()
