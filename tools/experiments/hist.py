"""Active-lane histogram of the refill trace kernel (needs a -DPT_TRACE_HIST variant through PT_PLUGIN)."""
import sys, os
sys.path.insert(0, os.getcwd())
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
s = scenes.make_scene("sponza", tex_size=64, detail=1.0)
pt = PathTracer(s, width=1920, height=1080, samplesPerPass=8, schedule=1)
pt.set_stats_level(1)
# shade kernel also counts into these fields: run with stats on, subtract a run where trace hist is absent? simpler: report raw
p = pt.params(seed=0x12345678)
pt.render_pass(p); pt.synchronize()
st = pt.stats().as_dict()
print({k: st[k] for k in ("attrFetches", "materialFetches", "lightFetches", "texelFetches", "texDescriptorFetches", "tlasNodeVisits", "nodeVisits", "triTests")})
