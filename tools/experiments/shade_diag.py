"""Lane utilisation of the shade kernel (needs a -DPT_SHADE_DIAG variant through PT_PLUGIN): wave-level executions of the
shade body (tlasNodeVisits) and lanes alive in them (instanceVisits)."""
import sys, os
sys.path.insert(0, os.getcwd())
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
s = scenes.make_scene("sponza", tex_size=64, detail=1.0)
pt = PathTracer(s, width=1920, height=1080, samplesPerPass=8, schedule=1)
pt.set_stats_level(1)
pt.render_pass(pt.params(seed=0x12345678)); pt.synchronize()
st = pt.stats().as_dict()
waves, lanes = st["tlasNodeVisits"], st["instanceVisits"]
print("shade wave executions", waves, "alive lanes", lanes, "utilisation", lanes / (64.0 * waves), "closest rays", st["closestHitRays"], "paths", st["paths"])
