"""Where a wave of the shade kernel spends its time (a -DPT_SHADE_TIME variant; pt_device.h path_shade_hit): clock ticks per section of the
shading step, summed over the waves of one pass of the benchmark scene, passes serialised.  usage: PT_PLUGIN=variants/shade_time.so python tools/experiments/shade_time.py"""
import sys; sys.path.insert(0, '.')
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
s = scenes.make_scene("sponza")
pt = PathTracer(s, width=1920, height=1080, samplesPerPass=8, schedule=1)
pt.set_passes_in_flight(1)
for k in range(2): pt.OnRenderImage(100 + k)
pt.synchronize(); pt.set_stats_level(1); pt.reset_stats()
pt.OnRenderImage(200); pt.synchronize()
d = pt.stats().as_dict()
names = [("attrFetches", "hit attributes + light intersection"), ("pixelsRead", "miss (sky)"), ("materialFetches", "material + textures"),
         ("texelFetches", "environment NEE"), ("lightFetches", "light NEE"), ("texDescriptorFetches", "BSDF sampling, roulette, new ray")]
tot = sum(d[k] for k, _ in names)
print("wave executions of the shading step:", d["instanceVisits"], " ticks>>4 in the listed sections:", tot)
for k, what in names: print(f"  {what:45s} {100.0 * d[k] / tot:5.1f} %   {d[k] / max(1, d['instanceVisits']) * 16:8.0f} ticks per wave execution")
pt.close()
