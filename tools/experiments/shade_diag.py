import sys; sys.path.insert(0, '.')
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
s = scenes.make_scene("sponza", tex_size=64)
pt = PathTracer(s, width=1920, height=1080, samplesPerPass=8, schedule=1)
pt.set_stats_level(1)
p = scenes.frame_params(s, 1920, 1080, spp=8, seed=0x12345678)
pt.render_pass(p); pt.synchronize()
st = pt.stats().as_dict()
print(st)
we, al = st["tlasNodeVisits"], st["instanceVisits"]
print(f"shade wave executions {we/1e6:.2f} M, active lanes {al/1e6:.2f} M -> {al/we/64*100:.1f} % lane utilisation; closest rays {st['closestHitRays']/1e6:.2f} M")
