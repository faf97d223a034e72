import sys, os; sys.path.insert(0, '.')
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
W,H=1920,1080
s = scenes.instanced_scene(count=200, detail=48)
pt = PathTracer(s, width=W, height=H, samplesPerPass=8, schedule=1)
pt.OnRenderImage(100); pt.synchronize(); pt.reset_stats()
pt.OnRenderImage(200); pt.synchronize()
st = pt.stats().as_dict(); px = W*H
print(os.environ.get("PT_PLUGIN","default").split("/")[-1], "c10 (pixelsWritten - px) =", st['pixelsWritten'] - px, " c11 (pixelsRead - px) =", st['pixelsRead'] - px, " rays", st['closestHitRays']+st['shadowRays'])
