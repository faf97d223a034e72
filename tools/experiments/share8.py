import sys, time; sys.path.insert(0, '.')
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
s = scenes.make_scene("sponza", tex_size=1024)
W,H=1920,1080
def seed_of(k): return (0x12345678 + k*0x9E3779B9) & 0xFFFFFFFF
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
K = 12
pt = PathTracer(s, width=W, height=H, samplesPerPass=8, rank=0, world_size=world, schedule=1)
def run(k0, n):
    t = []
    for k in range(k0, k0+n):
        p = scenes.frame_params(s, W, H, spp=8, current_sample=8*k, seed=seed_of(k))
        t0 = time.perf_counter(); pt.render_pass(p); pt.flip(); t.append(time.perf_counter() - t0)
    pt.synchronize()
    return t
run(0, 3)
t0=time.perf_counter(); t = run(3, K); wall=(time.perf_counter()-t0)*1e3/K
print(f"world {world}: {wall:.2f} ms/pass; host launch time per pass: mean {sum(t)/len(t)*1e3:.2f} ms, max {max(t)*1e3:.2f} ms", flush=True)
pt.close()
