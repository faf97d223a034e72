import sys, time, os; sys.path.insert(0, '.')
from unity_webgpu_pathtracer_amd import scenes
from unity_webgpu_pathtracer_amd.pathtracer import PathTracer
s = scenes.make_scene("sponza", tex_size=64)
W,H=1920,1080
def seed_of(k): return (0x12345678 + k*0x9E3779B9) & 0xFFFFFFFF
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 0
K = 24
pt = PathTracer(s, width=W, height=H, samplesPerPass=8, rank=0, world_size=world, schedule=1)
if iters: pt.set_wavefront_iterations(iters)
def run(k0, n):
    t = []
    for k in range(k0, k0+n):
        p = scenes.frame_params(s, W, H, spp=8, current_sample=8*k, seed=seed_of(k))
        t0 = time.perf_counter(); pt.render_pass(p); pt.flip(); t.append(time.perf_counter() - t0)
    pt.synchronize()
    return t
run(0, 16)                    # more passes than state sets: every set allocated and touched before the timed region
pt.reset_stats()
t0=time.perf_counter(); t = run(16, K); wall=(time.perf_counter()-t0)*1e3/K
rays = pt.stats().rays / K
print(f"{os.environ.get('PT_PLUGIN','default').split('/')[-1]} q={os.environ.get('GPU_MAX_HW_QUEUES')} world {world} iters {iters}: {wall:.2f} ms/pass -> {rays/wall/1e3*world:.0f} Mrays/s aggregate; host launch per pass: mean {sum(t)/len(t)*1e3:.2f} ms", flush=True)
pt.close()
