import sys; sys.path.insert(0, '.')
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
s = scenes.make_scene("sponza", tex_size=64)
pt = PathTracer(s, width=1920, height=1080, samplesPerPass=8, schedule=1)
pt.set_wavefront_iterations(200)   # everything in the wavefront kernels; shade-side counters are polluted by diag only in trace
pt.set_stats_level(1)
p = scenes.frame_params(s, 1920, 1080, spp=8, seed=0x12345678)
pt.render_pass(p); pt.synchronize()
st = pt.stats().as_dict()
print(st)
