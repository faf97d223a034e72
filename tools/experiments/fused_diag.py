"""Phase times of the fused persistent kernel (diagnostics build -DPT_FUSED_DIAG, PT_PLUGIN=tools/experiments/variants/diag.so)."""
import sys, os
sys.path.insert(0, os.getcwd())
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
s = scenes.make_scene("sponza", tex_size=1024, detail=1.0)
pt = PathTracer(s, width=1920, height=1080, samplesPerPass=8, schedule=4)
pt.set_passes_in_flight(1)
for k in range(2):
    pt.render_pass(pt.params(seed=1 + k)); pt.synchronize()
pt.reset_stats(); pt.set_stats_level(0)
import time
t0 = time.time(); pt.render_pass(pt.params(seed=5)); pt.synchronize(); dt = time.time() - t0
st = pt.stats().as_dict()
G = int(os.environ.get("FUSED_GROUPS", "4"))
waves = min(4096, 1920 * 1088 // (64 * G))
tot = st['attrFetches'] + st['materialFetches'] + st['lightFetches']
rays = st['closestHitRays'] + st['shadowRays']
rounds = st['texDescriptorFetches'] / waves
print(f"pass {dt*1e3:.2f} ms; wave time: refill {100*st['attrFetches']/tot:.1f} %, trace {100*st['materialFetches']/tot:.1f} %, shade {100*st['lightFetches']/tot:.1f} %; rounds/wave {rounds:.1f}; "
      f"rays/round {rays/waves/rounds:.0f}; trace wave-iterations/round {st['texelFetches']/waves/rounds:.1f} (tri block in {100*st['instanceVisits']/max(1,st['texelFetches']):.0f} %), scan windows/round {st['tlasNodeVisits']/waves/rounds:.1f}; "
      f"us per trace iteration {dt*1e6*st['materialFetches']/tot/(st['texelFetches']/waves):.2f}; us per shade group {dt*1e6*st['lightFetches']/tot/(rounds*G):.1f}")
